#!/usr/bin/env python3
"""Small and mid-size calls on text WITH long pieces that are not tokens (the synthetic web corpus: URLs, identifiers, hex strings):
wall time per call through the one-launch paths (tk_k_small / encode_mid) against the general pipeline (debug bits 2048 | 0x4000000),
parity of both with the oracle, and how many calls each path took."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as h
from tiktoken_amd import CoreBPE

name = "o200k_shaped"
g = h.load_golden(name)
C = h.c_oracle_for(name)
fast = CoreBPE(h.golden_vocab(name), g["special_tokens"], g["pat_str"])
os.environ["TIKTOKEN_AMD_DEBUG"] = str(2048 | 0x4000000)
slow = CoreBPE(h.golden_vocab(name), g["special_tokens"], g["pat_str"])
del os.environ["TIKTOKEN_AMD_DEBUG"]
blob, off = h.gen_corpus(0x51D0C0, 1, 4 << 20)
text = blob.tobytes()
rng = random.Random(5)


def cut(a, n):
    return text[a:a + n].decode("utf-8", errors="ignore").encode()


def med(core, data, reps=60):
    for _ in range(5): core._encode_np(data, None)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter_ns(); core._encode_np(data, None); ts.append(time.perf_counter_ns() - t0)
    ts.sort()
    return ts[len(ts) // 2] / 1e3


print("bytes      one-launch path   general pipeline   (median us per call, host buffers in and out; o200k_shaped, web corpus)")
for n in (300, 1000, 2048, 4096, 16384, 65536, 131072):
    rows = []
    for a in (0, 1 << 20, 2 << 20):
        d = cut(a, n)
        assert np.array_equal(fast._encode_np(d, None), C.encode_ordinary(d))
        rows.append((med(fast, d), med(slow, d)))
    print(f"{n:7d}   " + "   ".join(f"{f:7.1f} / {s:7.1f}" for f, s in rows))
s0 = {k: fast.stat(k) for k in ("small_calls", "mid_calls", "small_launches")}
bad = docs = 0
fast.set_profiling(False)
for _ in range(1500):
    n = rng.choice((40, 200, 700, 2048, 3000, 5000, 9000, 20000, 50000))
    d = cut(rng.randrange(0, len(text) - n), n)
    docs += 1
    bad += not np.array_equal(fast._encode_np(d, None), C.encode_ordinary(d))
for seed in range(3):
    for d in h.fuzz_batch(100 + seed, 1 << 20):
        if 0 < len(d) <= 131072:
            docs += 1
            bad += not np.array_equal(fast._encode_np(d, None), C.encode_ordinary(d))
s1 = {k: fast.stat(k) for k in s0}
print(f"parity: {docs} documents, {bad} mismatches; of them small calls {s1['small_calls'] - s0['small_calls']}, mid calls {s1['mid_calls'] - s0['mid_calls']}, launches {s1['small_launches'] - s0['small_launches']}")
sys.exit(1 if bad else 0)
