import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h
from tiktoken_amd._tiktoken import CoreBPE
from tiktoken_ext import amd_shaped
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16
spec = amd_shaped.o200k_custom8()
core = CoreBPE(spec["mergeable_ranks"], spec["special_tokens"], spec["pat_str"])
C = h.c_oracle.COracle(2, spec["mergeable_ranks"], spec["special_tokens"])
blob, off = h.gen_corpus(0x5EED0005, 1, mib << 20, threads=16)
bb = blob.tobytes()
rng = np.random.default_rng(5)
decoys = [b"<|custom_9|>", b"<|endoftext", b"<|custom_3|", b"<|", b"|>"]
parts = []
for d in range(len(off) - 1):
    t = bb[int(off[d]):int(off[d + 1])]
    out, pos = [], 0
    while pos < len(t):
        step = int(rng.integers(512, 3584))
        cut = min(len(t), pos + step)
        while cut < len(t) and (t[cut] & 0xC0) == 0x80: cut += 1
        out.append(t[pos:cut])
        if cut < len(t):
            out.append(b"<|custom_%d|>" % rng.integers(0, 8) if rng.random() < 0.8 else decoys[int(rng.integers(0, len(decoys)))])
        pos = cut
    parts.append(b"".join(out))
blob5, off5 = h.pack(parts)
toks, toff = core.encode_batch_packed(blob5, off5, "all")
rt, ro = C.encode_batch(blob5, off5, "all", 16)
print("equal:", np.array_equal(toff, ro) and np.array_equal(toks, rt), len(toks), len(rt))
nbad = 0
for d in range(len(parts)):
    a = toks[int(toff[d]):int(toff[d + 1])].tolist(); b = rt[int(ro[d]):int(ro[d + 1])].tolist()
    if a != b:
        nbad += 1
        if nbad <= 3:
            k = next(i for i in range(min(len(a), len(b)) + 1) if i >= min(len(a), len(b)) or a[i] != b[i])
            pre = core.decode_bytes(b[:k])
            print("doc", d, "len", len(parts[d]), "first diff token", k, "byte", len(pre), "gpu", a[k:k+6], "ref", b[k:k+6])
            print("   context:", parts[d][max(0, len(pre) - 40):len(pre) + 60])
            # does the doc alone encode right?
            alone = core._encode_np(parts[d], {"<|custom_%d|>" % i for i in range(8)} | {"<|endoftext|>", "<|endofprompt|>"}).tolist()
            print("   alone ok:", alone == b)
print("bad docs", nbad, "of", len(parts))
