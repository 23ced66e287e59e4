#!/usr/bin/env python3
"""A fuzz campaign on the CPU: whole-piece probe + merge through the device tables (tk_device.h compiled for the host: the short / mid / long
probes, the packed pair table, the per-lane merge) against the oracle's byte_pair_merge, on random pieces -- concatenations of vocabulary
tokens (so that long merges happen), random bytes, repeats.  usage: python tools/fuzz_pieces.py ENCODING SECONDS [SEED]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h

name, seconds = sys.argv[1], float(sys.argv[2])
rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 4242)
V = h.load_vocab(name)
sim = h.HostSim(h.PAT_STR[h.PATTERN_OF[name]], V, h.SPECIALS[name])
C = h.c_oracle_for(name)
toks = [t for t in V if len(t) <= 12]
t_end, n, nbytes = time.time() + seconds, 0, 0
while time.time() < t_end:
    r = rng.random()
    if r < 0.5:
        p = b"".join(rng.choice(toks) for _ in range(rng.randint(1, 8)))
    elif r < 0.7:
        p = bytes(rng.randrange(256) for _ in range(rng.randint(1, 40)))
    elif r < 0.85:
        p = rng.choice(toks) * rng.randint(1, 30)
    else:
        p = bytes(rng.choice(b"abcdefxyz0123456789-_/.%") for _ in range(rng.randint(1, 120)))
    p = p[:127]  # (the per-lane merge of the simulation takes pieces of up to 127 bytes; longer ones are the merge kernels' job)
    if not p:
        continue
    got, want = sim.encode_piece(p), C.encode_piece(p)
    if got != want:
        print(f"MISMATCH {name}: {p!r}: {got} != {want}")
        sys.exit(1)
    n += 1
    nbytes += len(p)
print(f"{name}: {n} pieces, {nbytes / 1e6:.1f} MB, no mismatch")
