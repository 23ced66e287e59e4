#!/usr/bin/env python3
"""GPU debugging aid: encode a synthetic corpus, find the first document whose tokens differ from the C oracle, and show where
(piece boundaries of the batch vs the oracle split, position inside the 3840-byte tile, the bytes around it)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h
from tiktoken_amd import CoreBPE

def main(tag=""):
    print("=====", tag, os.environ.get("TIKTOKEN_AMD_DEBUG"))
    name = sys.argv[1] if len(sys.argv) > 1 else "o200k_shaped"
    mix = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nbytes = int(sys.argv[3]) if len(sys.argv) > 3 else 8 << 20
    seed = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0x5EED0000 + mix + 7
    g = h.load_golden(name)
    core = CoreBPE(h.golden_vocab(name), g["special_tokens"], g["pat_str"])
    C = h.c_oracle_for(name)
    blob, off = h.gen_corpus(seed, mix, nbytes)
    toks, toff = core.encode_batch_packed(blob, off)
    rt, ro = C.encode_batch(blob, off, None, 8)
    print("tokens gpu/oracle", len(toks), len(rt))
    starts = core.pretokenize_packed(blob, off)
    bb = blob.tobytes()
    ref = []
    for d in range(len(off) - 1):
        a, b = int(off[d]), int(off[d + 1])
        ref += [a] + [a + e for e in C.split(bb[a:b])[:-1]] if b > a else []
    ref.append(len(bb))
    gs, rs = set(starts.tolist()), set(ref)
    print("piece starts gpu/oracle", len(gs), len(rs), "missing", len(rs - gs), "extra", len(gs - rs))
    for label, lst in (("missing", sorted(rs - gs)), ("extra", sorted(gs - rs))):
        for p in lst[:12]:
            d = int(np.searchsorted(off, p, side="right")) - 1
            print(label, p, "tile", p // 3840, "in-tile", p % 3840, "window pos", p % 3840 + 64, "doc", d, "doc-rel", p - int(off[d]), repr(bb[max(p - 24, 0):p]), "|", repr(bb[p:p + 24]))
    nbad = 0
    for d in range(len(off) - 1):
        a, b = toks[int(toff[d]):int(toff[d + 1])], rt[int(ro[d]):int(ro[d + 1])]
        if len(a) != len(b) or not np.array_equal(a, b):
            nbad += 1
            if nbad <= 6:
                alone = core._encode_np(bb[int(off[d]):int(off[d + 1])], None)
                print("   same document alone equals oracle:", bool(np.array_equal(alone, b)), "equals batch:", bool(np.array_equal(alone, a)))
                i = 0
                while i < min(len(a), len(b)) and a[i] == b[i]:
                    i += 1
                pre = core.decode_bytes(b[:i].tolist())
                print("doc", d, "bytes", int(off[d]), int(off[d + 1]), "first diff token", i, "text pos", int(off[d]) + len(pre), "tile", (int(off[d]) + len(pre)) // 3840,
                      "in-tile", (int(off[d]) + len(pre)) % 3840, "gpu", a[i:i + 6].tolist(), [core.decode_bytes([int(t)]) for t in a[i:i + 6]], "oracle", b[i:i + 6].tolist(),
                      [core.decode_bytes([int(t)]) for t in b[i:i + 6]])
    print("documents that differ:", nbad)
    return nbad

main("default")
os.environ["TIKTOKEN_AMD_DEBUG"] = "256"
main("no de-duplication")
