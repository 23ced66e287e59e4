import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, helpers as h
from tiktoken_amd import CoreBPE
for dbg in ("0", "2048", "4096", "6144"):
    os.environ["TIKTOKEN_AMD_DEBUG"] = dbg
    for name in ("o200k_shaped", "gpt2_shaped"):
        g = h.load_golden(name)
        core = CoreBPE(h.golden_vocab(name), g["special_tokens"], g["pat_str"])
        for unit, n in (("x", 1_000_000), ("x", 250_000), (" ", 1_000_000), ("0", 1_000_000)):
            s = (unit * n).encode()
            core._encode_np(s, None)
            core.set_profiling(True); core.reset_kernel_ms()
            t0 = time.perf_counter(); toks = core._encode_np(s, None); dt = time.perf_counter() - t0
            core.set_profiling(False)
            km = {k: round(core.kernel_ms(k)[0], 2) for k in ("tk_k_front", "tk_k_front_slow", "tk_k_merge_rounds", "tk_k_merge_long", "tk_k_back")}
            print(f"dbg {dbg:5} {name:13} {unit!r} x {n:8}: {dt*1e3:8.1f} ms tokens {len(toks):8} {km}", flush=True)
