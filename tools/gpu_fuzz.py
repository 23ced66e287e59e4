#!/usr/bin/env python3
"""Randomised batches of awkward documents (tests/helpers.py fuzz_batch: runs of one class, chains of uncertain boundaries, contractions in
every case, digits, white space with and without newlines, CJK, combining marks, specials, empty and tiny documents) through the whole
pipeline, every token against the C oracle.  Deterministic per (encoding, seed).  Usage: gpu_fuzz.py [rounds] [MiB per batch] [first seed]   |   gpu_fuzz.py generic [patterns] [seed]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as h
from tiktoken_amd import CoreBPE

if len(sys.argv) > 1 and sys.argv[1] == "generic":
    # gpu_fuzz.py generic [patterns] [seed]: random pat_str over the whole supported syntax, compiled and run on the device, split + gaps +
    # tokens against Python `regex` and the oracle (the loop of tests/test_gpu_regex.py::test_generated_patterns_on_the_device, more of it)
    from test_gpu_regex import generated_patterns_on_the_device

    n_pat = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    t0 = time.perf_counter()
    ran, gave_up = generated_patterns_on_the_device(seed, n_pat)
    print(f"generic engine: {ran} generated patterns equal to Python regex (split, gaps, tokens of every fourth), {gave_up} given up loudly, "
          f"{time.perf_counter() - t0:.1f} s", flush=True)
    sys.exit(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 24
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad = 0
for name in h.ENCODING_NAMES:
    g = h.load_golden(name)
    core = CoreBPE(h.golden_vocab(name), g["special_tokens"], g["pat_str"])
    C = h.c_oracle_for(name)
    for rnd in range(first, first + rounds):
        seed = zlib.crc32(name.encode()) ^ (rnd * 0x9E3779B1 & 0xFFFFFFFF)
        docs = h.fuzz_batch(seed, mib << 20)
        blob, off = h.pack(docs)
        for allowed in (None, "all"):
            t0 = time.perf_counter()
            try:
                toks, toff = core.encode_batch_packed(blob, off, allowed)
            except RuntimeError as e:
                print(f"{name} seed {rnd} allowed={allowed}: ERROR {e}", flush=True)
                bad += 1
                continue
            dt = time.perf_counter() - t0
            rt, ro = C.encode_batch(blob, off, allowed, os.cpu_count() or 8)
            ok = bool(np.array_equal(toff, ro) and np.array_equal(toks, rt))
            print(f"{name} seed {rnd} allowed={allowed}: {len(docs)} docs {len(blob) >> 20} MiB {dt * 1e3:.0f} ms {'ok' if ok else 'MISMATCH'}", flush=True)
            if not ok:
                bad += 1
                a, b = np.diff(toff.astype(np.int64)), np.diff(ro.astype(np.int64))
                d = np.flatnonzero(a != b)
                if len(d): print("   first document with another token count:", int(d[0]), docs[int(d[0])][:120])
print("mismatches:", bad)
sys.exit(1 if bad else 0)
