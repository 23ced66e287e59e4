#!/usr/bin/env python3
"""The o200k pat_str through the generic engine (TIKTOKEN_AMD_DEBUG=1048576) beside the hand-written scanners on a web-text corpus:
same tokens, kernel times of both.  Usage: generic_vs_scanners.py [MiB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as h
from tiktoken_amd import CoreBPE

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
name = "o200k_shaped"
g = h.load_golden(name)
t = time.perf_counter(); blob, off = h.gen_corpus(0x5EED0003, 1, mib << 20, 32); print("corpus: %d MiB, %d documents, %.1f s" % (mib, len(off) - 1, time.perf_counter() - t), flush=True)
NAMES = ["tk_k_mark_docs", "tk_k_rx_speculate", "tk_k_rx_resolve", "tk_k_rx_merge", "tk_k_front", "tk_k_front_slow", "tk_k_bincount", "tk_k_binfill",
         *[f"tk_k_merge_llane_{i}" for i in (16, 24, 32, 48, 64)], *[f"tk_k_merge_group_{i}" for i in (8, 16, 32, 64)], "tk_k_merge_rounds",
         "tk_k_merge_rounds_wide", "tk_k_merge_long", "tk_k_place", "tk_k_docoff", "tk_k_scan_small"]
res = {}
for label, dbg in (("scanners", None), ("generic", "1048576")):
    if dbg: os.environ["TIKTOKEN_AMD_DEBUG"] = dbg
    core = CoreBPE(h.golden_vocab(name), g["special_tokens"], g["pat_str"])
    os.environ.pop("TIKTOKEN_AMD_DEBUG", None)
    core.encode_batch_packed(blob, off)
    core.set_profiling(True)
    core.reset_kernel_ms()
    t = time.perf_counter(); toks, toff = core.encode_batch_packed(blob, off); dt = time.perf_counter() - t
    ks = {k: core.kernel_ms(k)[0] for k in NAMES if core.kernel_ms(k)[1]}
    split = sum(v for k, v in ks.items() if k in ("tk_k_rx_speculate", "tk_k_rx_resolve", "tk_k_rx_merge", "tk_k_front", "tk_k_front_slow", "tk_k_mark_docs"))
    print("%-8s host-to-host %.1f ms; split + probe kernels %.2f ms = %.1f GB/s of text; all kernels %.2f ms; %s" % (
        label, dt * 1e3, split, len(blob) / split / 1e6, sum(ks.values()), " ".join("%s=%.2f" % (k.replace("tk_k_", ""), v) for k, v in ks.items() if v >= 0.05)), flush=True)
    res[label] = (toks, toff)
same = np.array_equal(res["scanners"][0], res["generic"][0]) and np.array_equal(res["scanners"][1], res["generic"][1])
print("every token equal:", same, "(%d tokens)" % len(res["scanners"][0]))
sys.exit(0 if same else 1)
