#!/usr/bin/env python3
"""Host-buffer path (tk_encode_batch, staged + overlapped copies) against the device-resident path on the same corpus."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as h
from bench import DevArray, gen_corpus
from tiktoken_amd._tiktoken import CoreBPE
from tiktoken_ext import amd_shaped

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spec = amd_shaped.ENCODING_CONSTRUCTORS["o200k_shaped"]()
core = CoreBPE(spec["mergeable_ranks"], spec["special_tokens"], spec["pat_str"])
n = mib << 20
blob, off = gen_corpus(0x5EED0003, 1, n, 32)
nd = len(off) - 1
d_text = torch.from_numpy(blob).cuda(); d_off = torch.from_numpy(off.view(np.int64)).cuda()
dt, nt, do = core.encode_batch_device(d_text.data_ptr(), n, d_off.data_ptr(), off, nd)
ref_off = torch.as_tensor(DevArray(do, nd + 1, "<i8"), device="cuda").cpu().numpy().astype(np.uint64)
ref_tok = torch.as_tensor(DevArray(dt, max(nt, 1), "<i4"), device="cuda")[:nt].cpu().numpy().view(np.uint32)
for it in range(5):
    t0 = time.perf_counter(); tok, toff = core.encode_batch_packed(blob[:n], off); dt_ = time.perf_counter() - t0
    same = np.array_equal(toff, ref_off) and np.array_equal(tok, ref_tok)
    print(f"run {it}: {dt_*1e3:.1f} ms  tokens {len(tok)} vs {nt}  identical {same}", flush=True)
    if not same and len(tok) == nt and np.array_equal(toff, ref_off):
        bad = np.flatnonzero(tok != ref_tok)
        print(f"  same offsets; {len(bad)} token ids differ; first at {bad[:5]} last at {bad[-5:]}")
        runs = np.flatnonzero(np.diff(bad) > 1)
        print(f"  {len(runs) + 1} runs; run starts {bad[np.r_[0, runs + 1]][:12]}")
        d = int(np.searchsorted(ref_off, bad[0], side="right")) - 1
        print(f"  first differing token in doc {d} at bytes [{off[d]}, {off[d+1]}): got {tok[bad[0]:bad[0]+6]} want {ref_tok[bad[0]:bad[0]+6]}")
    elif not same:
        a, b = np.diff(toff.astype(np.int64)), np.diff(ref_off.astype(np.int64))
        bad = np.flatnonzero(a != b)
        print("  docs with another token count:", len(bad), bad[:10])
        for d in bad[:10]:
            print(f"   doc {d}: bytes [{off[d]}, {off[d+1]})  mod 64MiB {off[d] % (64<<20)} .. {off[d+1] % (64<<20)}  tokens {a[d]} vs {b[d]}")
