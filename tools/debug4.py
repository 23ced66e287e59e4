import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h
from tiktoken_amd import CoreBPE
name = "o200k_shaped"
g = h.load_golden(name)
core = CoreBPE(h.golden_vocab(name), g["special_tokens"], g["pat_str"])
C = h.c_oracle_for(name)
thai = bytes.fromhex("e0b8aee0b88ae0b881")
doc = b"x" * 4025 + thai
print("single piece ok:", core.encode_single_piece(doc) == C.encode_piece(doc))
for k in list(range(4016, 4040)) + [3000, 3830, 3838, 3840, 3842, 5000, 7700]:
    for tail in (thai, b"yz", "é".encode() * 3):
        d = b"x" * k + tail
        got, want = core._encode_np(d, None), C.encode_ordinary(d)
        if not np.array_equal(got, want):
            print("BAD k", k, "tail", tail.hex(), "len", len(d), "gpu n", len(got), "oracle n", len(want), "gpu tail", got[-3:].tolist(), "oracle tail", want[-3:].tolist())
print("done")
