#!/usr/bin/env python3
"""Timing + parity of the reference's pathological inputs (tests/test_encoding.py:52-57,113-124 and CHANGELOG v0.13.0): long runs of one
character or of a short unit.  Prints per input: wall time of Encoding.encode_ordinary (second call), the library's per-kernel times of
that call, token count, parity with the C oracle."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, helpers as h
import tiktoken_amd

names = sys.argv[1:] or ["o200k_shaped"]
KERN = ["tk_k_front", "tk_k_front_slow", "tk_k_merge_rounds", "tk_k_merge_rounds_wide", "tk_k_merge_long", "tk_k_place"]
for name in names:
    enc = tiktoken_amd.get_encoding(name)
    C = h.c_oracle_for(name)
    core = enc._core_bpe
    units = ["x", "0", "^", " ", "\n", "a1", "中", "1", "é", " \n"] + ([] if os.environ.get("STRESS_SKIP_CHAINS") else ["Ab", "x'll"])
    for unit in units:
        for n in (100_000, 1_000_000):
            s = unit * (n // len(unit))
            enc.encode_ordinary(s)
            core.set_profiling(True); core.reset_kernel_ms()
            t0 = time.perf_counter(); toks = enc.encode_ordinary(s); dt = time.perf_counter() - t0
            core.set_profiling(False)
            km = {k: round(core.kernel_ms(k)[0], 2) for k in KERN}
            ok = bool(np.array_equal(np.asarray(toks, np.uint32), C.encode_ordinary(s.encode())))
            print(f"{name} {unit!r:7} x {n:>9}: {dt*1e3:9.1f} ms  tokens {len(toks):>8}  parity {ok}  kernels {km}", flush=True)
