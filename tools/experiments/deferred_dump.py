import os, sys, numpy as np, torch
ROOT="/root/repo"
sys.path[:0]=[ROOT, ROOT+"/tests"]
import helpers as h
from tiktoken_amd._tiktoken import CoreBPE
from tiktoken_ext import amd_shaped
for cfg in sys.argv[1:]:
    enc_name, pat, specs, blob, off, allowed = h.baseline_config(cfg)
    en = {"C2":"cl100k_shaped","C5":"o200k_custom8","C3":"o200k_shaped"}[cfg]
    spec = amd_shaped.ENCODING_CONSTRUCTORS[en]()
    core = CoreBPE(spec["mergeable_ranks"], spec["special_tokens"], spec["pat_str"])
    n=len(blob); host=np.zeros(n+64,np.uint8); host[:n]=blob
    d_text=torch.from_numpy(host).cuda(); d_off=torch.from_numpy(off.view(np.int64)).cuda(); nd=len(off)-1
    core.encode_batch_device(d_text.data_ptr(), n, d_off.data_ptr(), off, nd, allowed)
    torch.cuda.synchronize()
    cnt = core.stat("deferred_count")
    print(cfg, "deferred tiles:", cnt)
    for i in range(min(cnt, 12)):
        t = core.stat(f"deferred_tile_{i}")
        a = t * 3840
        ctx = bytes(blob[max(0, a - 128):a + 3840 + 128])
        print("  tile", t, "left context + 40:", ctx[:168].decode("utf-8", "replace").replace("\n", "\\n"))
        print("      window's end:", ctx[-200:].decode("utf-8", "replace").replace("\n", "\\n"))
