#!/usr/bin/env python3
"""GPU debugging aid: shrink a corpus mismatch to a small single document that keeps its alignment to the 3840-byte tiles."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h
from tiktoken_amd import CoreBPE

TILE = 3840
name, mix, nbytes = "o200k_shaped", 1, 8 << 20
g = h.load_golden(name)
core = CoreBPE(h.golden_vocab(name), g["special_tokens"], g["pat_str"])
C = h.c_oracle_for(name)
blob, off = h.gen_corpus(0x5EED0000 + mix + 7, mix, nbytes)
bb = blob.tobytes()

def bad(data: bytes) -> bool:
    return not np.array_equal(core._encode_np(data, None), C.encode_ordinary(data))

def cut_to_char(data, i):
    while i < len(data) and (data[i] & 0xC0) == 0x80:
        i += 1
    return i

pos = int(sys.argv[1]) if len(sys.argv) > 1 else 522428
t = pos // TILE
a = cut_to_char(bb, (t - 1) * TILE)
b = cut_to_char(bb, (t + 2) * TILE)
pad = a % TILE
doc = b"x" * pad + bb[a:b]
print("window doc bad:", bad(doc), "len", len(doc), "pad", pad)
# shrink the tail
lo, hi = pos - a + pad, len(doc)
while hi - lo > 1:
    mid = cut_to_char(doc, (lo + hi) // 2)
    if mid >= hi:
        break
    if bad(doc[:mid]):
        hi = mid
    else:
        lo = mid
doc = doc[:hi]
print("after tail shrink:", len(doc), bad(doc))
# shrink the head: replace leading bytes by 'x' (same length keeps the alignment)
lo, hi = 0, pos - a + pad
keep = 0
while hi - lo > 1:
    mid = cut_to_char(doc, (lo + hi) // 2)
    if mid >= hi:
        break
    cand = b"x" * mid + doc[mid:]
    if bad(cand):
        lo = mid
        keep = mid
    else:
        hi = mid
doc = b"x" * keep + doc[keep:]
print("after head shrink: x *", keep, "+", len(doc) - keep, "bytes", bad(doc))
tail = doc[keep:]
print("tail repr:", repr(tail[:400]))
print("tail hex:", tail[:120].hex())
got, want = core._encode_np(doc, None), C.encode_ordinary(doc)
print("gpu   :", got[-12:].tolist(), [core.decode_bytes([int(x)]) for x in got[-6:]])
print("oracle:", want[-12:].tolist(), [core.decode_bytes([int(x)]) for x in want[-6:]])
st = core.pretokenize_packed(np.frombuffer(doc, np.uint8), np.array([0, len(doc)], np.uint64))
ends = C.split(doc)
print("gpu starts tail:", st[-8:].tolist(), "oracle ends tail:", ends[-8:], "equal:", st[1:].tolist() == ends)
print("positions in tile:", [(int(x) % TILE) for x in st[-8:]])
