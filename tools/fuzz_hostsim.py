#!/usr/bin/env python3
"""A fuzz campaign on the CPU: the device's pre-tokeniser logic (tk_chunk.h / tk_device.h, compiled for the host by tests/hostsim) against
the oracle's split, with seeds the test-suite does not use.  usage: python tools/fuzz_hostsim.py ENCODING SECONDS [SEED0]
Prints the first mismatching document (and exits 1) or a summary."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as h

name, seconds = sys.argv[1], float(sys.argv[2])
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else int(time.time())
sim = h.HostSim(h.PAT_STR[h.PATTERN_OF[name]], h.load_vocab(name), h.SPECIALS[name])
C = h.c_oracle_for(name)
t_end = time.time() + seconds
n_docs = n_bytes = 0
seed = seed0
units = h.ADV + h.FUZZ_UNITS + [" ", "­", "ͅ", "ẞ", "ﬁ", "\U0001d400", "กั", "́́", "'S", "'LL", "'rE", "\r", "\r\r\n", " \r", "\x0b", "\x0c", "\x1f", " ", "﻿"]
while time.time() < t_end:
    seed += 1
    rng = random.Random(seed)
    docs = []
    for _ in range(rng.randint(1, 6)):
        r = rng.random()
        if r < 0.5:
            docs.append("".join(rng.choice(units) for _ in range(rng.randint(0, 80))).encode())
        elif r < 0.8:
            docs.append(h.fuzz_doc(rng)[:30000].encode())
        else:  # long runs with awkward joints, shifted against the tiles
            parts = ["y" * rng.randint(0, 70)]
            for _ in range(rng.randint(1, 12)):
                parts.append(rng.choice(["", " ", "x", "'", "X", "\n"]) + rng.choice(units) * rng.choice([1, 2, 3, 30, 59, 64, 65, 130, 700, 4100]) + rng.choice(["", " ", "b", "'ll", "\n", "9", "'S"]))
            docs.append("".join(parts).encode())
    blob, off = h.pack(docs)
    ref = []
    for d, dd in enumerate(docs):
        ref += [int(off[d]) + e for e in C.split(dd)]
    for what, got in (("bytewalk", sim.piece_ends(blob, off)[0]), ("bitparallel", sim.piece_ends(blob, off, bits=True)[0]),
                      ("tiles 64/16", sim.piece_ends_tiled(blob, off, tile=64, left=16)[0]), ("tiles 4096/64", sim.piece_ends_tiled(blob, off)[0])):
        if got.tolist() != ref:
            print(f"MISMATCH {name} seed {seed} {what}: {docs!r}"[:4000])
            sys.exit(1)
    n_docs += len(docs)
    n_bytes += len(blob) - 64
print(f"{name}: seeds {seed0 + 1}..{seed}, {n_docs} documents, {n_bytes / 1e6:.1f} MB, no mismatch")
