#!/usr/bin/env python3
"""Train the offline "-shaped" vocabularies used by tests and bench.py.

No stock vocabulary file (*.tiktoken, vocab.bpe) exists in this environment and there is
no network, so the real `pat_str` / special-token ids of the stock
encodings (reference tiktoken_ext/openai_public.py) are paired with ranks trained here:

  * text: tkc_generate() synthetic corpus (tiktoken_amd/csrc/corpus_gen.cpp), fixed seed;
  * split: Python `regex.findall(pat_str)` -- the same split the encode path must
    reproduce, so merges never cross piece boundaries (as in real BPE training,
    reference tiktoken/_educational.py:119-141);
  * merges: HuggingFace `tokenizers` BpeTrainer over the byte-level alphabet;
  * conversion to `dict[bytes,int]`: 256 single bytes in the data-gym order, then one
    rank per merge in merge order -- what reference tiktoken/load.py:89-144 does for GPT-2.

Output: tiktoken_amd/vocab/<name>.tiktoken.gz in the reference's own wire format
(`base64(token) SP rank LF`, tiktoken/load.py:147-171), gzip-compressed, mtime zeroed so the
file is reproducible.

Usage: python tools/train_vocab.py {gpt2_shaped|cl100k_shaped|o200k_shaped} [train_MiB]
"""
import base64
import ctypes
import gzip
import io
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import regex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import py_oracle  # noqa: E402  (pattern strings only)

SPECS = {
    # name: (pattern, number of mergeable ranks, corpus mix, seed)
    "gpt2_shaped": (py_oracle.R50K_PAT, 50256, 1, 0x5EEDA001),
    "cl100k_shaped": (py_oracle.CL100K_PAT, 100256, 0, 0x5EEDA002),
    "o200k_shaped": (py_oracle.O200K_PAT, 199998, 1, 0x5EEDA003),
}
DELIM = "\x00"


def corpus(seed, mix, nbytes):
    lib = ctypes.CDLL(os.path.join(ROOT, "tiktoken_amd/csrc/libtkcorpus.so"))
    lib.tkc_generate.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int]
    out = np.empty(nbytes, np.uint8)
    maxd = nbytes // 64 + 2
    off = np.empty(maxd + 1, np.uint64)
    nd = ctypes.c_uint64()
    rc = lib.tkc_generate(seed, mix, nbytes, out.ctypes.data, off.ctypes.data, maxd, ctypes.byref(nd), 8)
    assert rc == 0
    return out.tobytes(), off[: nd.value + 1]


_pat = None


def _init(pat_str):
    global _pat
    _pat = regex.compile(pat_str)


def _split(doc: str) -> str:
    return DELIM.join(_pat.findall(doc))


def data_gym_byte_order():
    order = [b for b in range(256) if chr(b).isprintable() and chr(b) != " "]
    order += [b for b in range(256) if b not in order]
    return order


def bytes_to_unicode():
    # the GPT-2 byte <-> printable-unicode table used by tokenizers' ByteLevel
    bs = data_gym_byte_order()[:188]
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {chr(c): b for b, c in zip(bs, cs)}


def main():
    name = sys.argv[1]
    train_mib = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    pat_str, n_ranks, mix, seed = SPECS[name]
    t0 = time.time()
    blob, off = corpus(seed, mix, train_mib << 20)
    if name == "o200k_shaped":  # blend in the multilingual mix as well
        blob2, off2 = corpus(seed + 1, 0, (train_mib // 3) << 20)
        docs = [blob[int(a):int(b)].decode() for a, b in zip(off[:-1], off[1:])]
        docs += [blob2[int(a):int(b)].decode() for a, b in zip(off2[:-1], off2[1:])]
    else:
        docs = [blob[int(a):int(b)].decode() for a, b in zip(off[:-1], off[1:])]
    print("corpus ready", len(docs), "docs", time.time() - t0, flush=True)
    with mp.Pool(8, initializer=_init, initargs=(pat_str,)) as pool:
        split_docs = pool.map(_split, docs, chunksize=256)
    print("split done", time.time() - t0, flush=True)

    from tokenizers import Regex, Tokenizer, models, pre_tokenizers, trainers

    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.Sequence(
        [pre_tokenizers.Split(Regex(DELIM), "removed"), pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)]
    )
    trainer = trainers.BpeTrainer(
        vocab_size=n_ranks + 2000,  # slack for duplicate byte strings, trimmed below
        initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
        special_tokens=[],
        show_progress=False,
    )
    tok.train_from_iterator(split_docs, trainer=trainer)
    print("trained", time.time() - t0, flush=True)
    import json

    model = json.loads(tok.to_str())["model"]
    u2b = bytes_to_unicode()

    def tobytes(s):
        return bytes(u2b[ch] for ch in s)

    ranks = {}
    for b in data_gym_byte_order():
        ranks[bytes([b])] = len(ranks)
    for m in model["merges"]:
        a, b = m if isinstance(m, list) else m.split(" ")
        t = tobytes(a) + tobytes(b)
        if t not in ranks:
            ranks[t] = len(ranks)
        if len(ranks) == n_ranks:
            break
    assert len(ranks) == n_ranks, (len(ranks), n_ranks, "train on more data")
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", mtime=0, compresslevel=9) as gz:
        for t, r in sorted(ranks.items(), key=lambda kv: kv[1]):
            gz.write(base64.b64encode(t) + b" " + str(r).encode() + b"\n")
    os.makedirs(os.path.join(ROOT, "tiktoken_amd/vocab"), exist_ok=True)
    path = os.path.join(ROOT, "tiktoken_amd/vocab", name + ".tiktoken.gz")
    with open(path, "wb") as f:
        f.write(buf.getvalue())
    print("wrote", path, len(buf.getvalue()), "bytes; max token len", max(map(len, ranks)), time.time() - t0)


if __name__ == "__main__":
    main()
