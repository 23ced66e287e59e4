#!/usr/bin/env python3
"""Mid-size calls through the general pipeline: wall time per call (host buffers in and out) and the kernels behind it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h
import tiktoken_amd
from bench import KERNELS

_, _, _, blob, off, _ = h.baseline_config("C1")
g = tiktoken_amd.get_encoding("gpt2_shaped")
core = g._core_bpe
for nbytes in ([int(os.environ["MID_ONLY"])] if os.environ.get("MID_ONLY") else (4096, 65536, 1 << 20)):
    data = blob[:nbytes].tobytes()
    for _ in range(5): core._encode_np(data, None)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter_ns(); core._encode_np(data, None); ts.append(time.perf_counter_ns() - t0)
    ts.sort()
    core.set_profiling(True); core.reset_kernel_ms()
    core._encode_np(data, None)
    core.set_profiling(False)
    ks = {k: core.kernel_ms(k) for k in KERNELS + ["tk_k_small", "tk_k_single_front"]}
    ks = {k: (round(v[0] * 1e3, 1), v[1]) for k, v in ks.items() if v[1]}
    print(f"{nbytes} bytes: median {ts[len(ts)//2]/1e3:.1f} us  p10 {ts[len(ts)//10]/1e3:.1f} us; kernels (us, launches): {ks}; sum {sum(v[0] for v in ks.values()):.1f} us in {sum(v[1] for v in ks.values())} launches")
