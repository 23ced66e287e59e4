#!/usr/bin/env python3
"""Host-path decode (tk_decode_batch) on 256 MiB of the bench corpus: wall time of repeated calls, ids page-locked and pageable."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import gen_corpus
from tiktoken_amd._tiktoken import CoreBPE
from tiktoken_ext import amd_shaped
spec = amd_shaped.o200k_shaped()
core = CoreBPE(spec["mergeable_ranks"], spec["special_tokens"], spec["pat_str"])
blob, off = gen_corpus(0x5EED0003, 1, 256 << 20, 32)
n = int(off[-1])
toks, toff = core.encode_batch_packed(blob[:n], off)
for name, src in (("page-locked ids", toks), ("pageable ids", np.array(toks))):
    for rep in range(4):
        t0 = time.perf_counter()
        data, boff = core.decode_batch_packed(src, toff, as_array=True)
        dt = time.perf_counter() - t0
        ok = bool(np.array_equal(data, blob[:n]))
        print(f"{name}: run {rep}: {dt * 1e3:.2f} ms = {n / dt / 1e9:.2f} GB/s of text, identical {ok}", flush=True)
        del data
