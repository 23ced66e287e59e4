import os, sys, numpy as np, torch
ROOT="/root/repo"
sys.path[:0]=[ROOT, ROOT+"/tests"]
import helpers as h
from tiktoken_amd._tiktoken import CoreBPE
from tiktoken_ext import amd_shaped
for cfg in (sys.argv[1:] or ["C2","C5"]):
    enc_name, pat, specs, blob, off, allowed = h.baseline_config(cfg)
    en = {"C2":"cl100k_shaped","C5":"o200k_custom8","C3":"o200k_shaped","N1":"o200k_shaped"}[cfg]
    spec = amd_shaped.ENCODING_CONSTRUCTORS[en]()
    core = CoreBPE(spec["mergeable_ranks"], spec["special_tokens"], spec["pat_str"])
    n=len(blob); host=np.zeros(n+64,np.uint8); host[:n]=blob
    d_text=torch.from_numpy(host).cuda(); d_off=torch.from_numpy(off.view(np.int64)).cuda(); nd=len(off)-1
    core.encode_batch_device(d_text.data_ptr(), n, d_off.data_ptr(), off, nd, allowed)
    core.stat("time_reset")
    core.encode_batch_device(d_text.data_ptr(), n, d_off.data_ptr(), off, nd, allowed)
    torch.cuda.synchronize()
    tiles=max(core.stat("time_15"),1); d=core.stat("time_14"); ts=max(core.stat("time_s15"),1)
    print(cfg, "tiles", tiles, "deferred: no certain start in left context", d & 0xFFFFFFFF, "piece leaves window", d>>32, "starts-kernel tiles", ts,
          "cycles per deferred tile", {i: round(core.stat(f"time_s{i}")/ts) for i in (0,1,2,10,12,13,3)},
          "cycles per tile", {i: round(core.stat(f"time_{i}")/tiles) for i in (0,1,2,12,13,3,4,5,10,9)})
