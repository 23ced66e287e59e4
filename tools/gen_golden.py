#!/usr/bin/env python3
"""Generate tests/golden/*.json.gz by running the REFERENCE's own Python code.

Runs only in the build container (needs /root/reference); the fixtures it writes travel to
the GPU box.  What is executed from the reference:

  * `tiktoken._educational.bpe_encode` (reference tiktoken/_educational.py:83-116), the
    reference-authored pure-Python greedy merge, on every piece;
  * `regex.findall(pat_str, text)` exactly as reference tiktoken/core.py:395-404 and
    tiktoken/_educational.py:23-37 do;
  * `tiktoken.core.Encoding.__init__` argument checks via a stub `_tiktoken` module (the Rust
    extension itself cannot be built here: no cargo/rustc).

The whole-piece shortcut of src/lib.rs:367-368 is applied on top (a piece that is itself a
vocabulary key yields that single rank); the script asserts that the shortcut and the pure
merge agree for every such piece it meets, and that every vocabulary token re-encodes to
itself on a sample, so the shortcut is not hiding a divergence.

Special-token cases follow src/lib.rs:375-442 statement by statement: the next special is found with the
reference's own `tiktoken.core._special_token_regex` (core.py:431-438), a disallowed hit resumes the search one
char later, each slice is encoded as an independent haystack by the reference code above, and the special's
id is inserted.  No fixture registers a special that is a prefix of another (there the reference's pick
depends on hash-map order, lib.rs:625-631); the rule is recorded in the fixture as `special_tie_rule`.

Usage: python tools/gen_golden.py
"""
import base64
import gzip
import json
import os
import random
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

# --- import the reference's Python layer with the native extension stubbed out
stub = types.ModuleType("tiktoken._tiktoken")


class _StubCoreBPE:
    def __init__(self, *a, **k):
        pass


stub.CoreBPE = _StubCoreBPE
sys.modules["tiktoken._tiktoken"] = stub
sys.path.insert(0, REF)
import regex  # noqa: E402
import tiktoken  # noqa: E402  (the reference package)
import tiktoken._educational as edu  # noqa: E402
from tiktoken.core import Encoding as RefEncoding  # noqa: E402

assert tiktoken.__file__.startswith(REF), tiktoken.__file__

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import py_oracle as po  # noqa: E402  (pattern strings only)

sys.path.insert(0, ROOT)


def load_vocab(name):
    d = {}
    with gzip.open(os.path.join(ROOT, "tiktoken_amd/vocab", name + ".tiktoken.gz")) as f:
        for line in f.read().splitlines():
            t, r = line.split()
            d[base64.b64decode(t)] = int(r)
    return d


def corpus_sample(seed, mix, nbytes):
    import ctypes

    import numpy as np

    lib = ctypes.CDLL(os.path.join(ROOT, "tiktoken_amd/csrc/libtkcorpus.so"))
    lib.tkc_generate.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int]
    out = np.empty(nbytes, np.uint8)
    maxd = nbytes // 64 + 2
    off = np.empty(maxd + 1, np.uint64)
    nd = ctypes.c_uint64()
    assert lib.tkc_generate(seed, mix, nbytes, out.ctypes.data, off.ctypes.data, maxd, ctypes.byref(nd), 2) == 0
    b = out.tobytes()
    return [b[int(off[i]):int(off[i + 1])].decode() for i in range(nd.value)]


LOREM = ("Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor incididunt ut labore et "
         "dolore magna aliqua. Ut enim ad minim veniam, quis nostrud exercitation ullamco laboris nisi ut aliquip ex "
         "ea commodo consequat. Duis aute irure dolor in reprehenderit in voluptate velit esse cillum dolore eu "
         "fugiat nulla pariatur. Excepteur sint occaecat cupidatat non proident, sunt in culpa qui officia deserunt "
         "mollit anim id est laborum. ")

ADV = list("aAsStTlLvVeErRdDmMxZ") + ["ſ", "中", "́", "ʰ", "ǅ", "1", "2", "²", "٣", " ", " ", "\t", "\r", "\n",
                                         "　", "\x85", "'", "/", "!", ".", "\x1c", "é", "Ω", "я", "ก", "ั", "😀", "’",
                                         "hello", " world", "ing", "tion", "<|", "|>"]


def reference_texts():
    """Inputs the reference's own tests use (tests/test_encoding.py:14-124,149-155; test_offsets.py:49-79)."""
    t = ["", "hello world", "hello <|endoftext|>", "rer", "'rer", "today\n ", "today\n \n", "today\n  \n",
         " \x850", "👍", "�", "hello", "hello ", "hello  ", " hello", " hello ", " hello  ", "goodbye world",
         "请考试我的软件！12345", "helloqqqq", "hello fanta", "<|endoftext|>",
         "<|endoftext|> hello <|fim_prefix|> there <|fim_middle|>", "我非常渴望与人工智能一起工作", "நடிகர் சூர்யா",
         " Ġ除", "hello world<|endoftext|> green cow"]
    t += ["0" * k for k in range(1, 18)]
    for c in ["^", "0", "a", "'s", " ", "\n"]:
        big = c * 400  # the reference uses 10_000 (test_encoding.py:113-124); the GPU tests do that size
        t += [big, " " + big, " " + big + "\n"]
    t += ["x" * 300]
    return t


def ref_encode_ordinary(pat, ranks, text, stats):
    out = []
    for piece in pat.findall(text):
        pb = piece.encode("utf-8")
        merged = edu.bpe_encode(ranks, pb, visualise=None)
        if pb in ranks:
            stats["shortcut"] += 1
            assert merged == [ranks[pb]], (pb, merged)  # shortcut == pure merge for this piece
            out.append(ranks[pb])
        else:
            out.extend(merged)
        stats["pieces"] += 1
    return out


def ref_encode(pat, ranks, specials, text, allowed, stats):
    """src/lib.rs:375-442 statement by statement, with the REFERENCE's own special-token regex
    (tiktoken.core._special_token_regex, core.py:431-438: the alternation of every REGISTERED special, escaped --
    the same pattern CoreBPE::new_internal compiles at lib.rs:625-631)."""
    from tiktoken.core import _special_token_regex

    special_regex = _special_token_regex(frozenset(specials))
    allowed = set(allowed)
    out, start = [], 0
    while True:
        start_find = start
        while True:  # lib.rs:389-401: next ALLOWED special; a disallowed hit resumes one char after its start
            m = special_regex.search(text, start_find) if specials else None
            if m is None or m.group(0) in allowed:
                break
            start_find = m.start() + 1
        end = m.start() if m else len(text)
        out += ref_encode_ordinary(pat, ranks, text[start:end], stats)
        if m is None:
            return out
        out.append(specials[m.group(0)])
        start = m.end()


def check_no_prefix_ties(specials):
    """When one registered special is a prefix of another the reference's pick at that position depends on the
    iteration order of a hash map (lib.rs:625-631) -- such sets are kept out of the fixtures."""
    names = list(specials)
    for a in names:
        for b in names:
            assert a == b or not b.startswith(a), (a, b)


ENCODINGS = {
    # name: (pat_str, vocab file, special tokens, corpus mix)  -- ids follow openai_public.py:29,80-86,100
    "gpt2_shaped": (po.R50K_PAT, "gpt2_shaped", {"<|endoftext|>": 50256}, 1),
    "cl100k_shaped": (po.CL100K_PAT, "cl100k_shaped",
                      {"<|endoftext|>": 100257, "<|fim_prefix|>": 100258, "<|fim_middle|>": 100259,
                       "<|fim_suffix|>": 100260, "<|endofprompt|>": 100276}, 0),
    "o200k_shaped": (po.O200K_PAT, "o200k_shaped", {"<|endoftext|>": 199999, "<|endofprompt|>": 200018}, 1),
}


def main():
    rng = random.Random(0x601DE2)
    for name, (pat_str, vocab, specials, mix) in ENCODINGS.items():
        ranks = load_vocab(vocab)
        # the reference's own constructor checks (core.py:47-57) accept this vocabulary
        RefEncoding(name, pat_str=pat_str, mergeable_ranks=ranks, special_tokens=specials)
        check_no_prefix_ties(specials)
        pat = regex.compile(pat_str)
        stats = {"pieces": 0, "shortcut": 0}
        cases = []

        def add(cname, text, allowed=None):
            if allowed is None:
                toks = ref_encode_ordinary(pat, ranks, text, stats)
            else:
                toks = ref_encode(pat, ranks, specials, text, allowed, stats)
            cases.append({"name": cname, "text": base64.b64encode(text.encode("utf-8")).decode(),
                          "allowed": None if allowed is None else sorted(allowed), "tokens": toks})

        for i, t in enumerate(reference_texts()):
            add(f"ref{i}", t)
        add("lorem4k", (LOREM * 10)[:4096])
        for i in range(400):
            add(f"adv{i}", "".join(rng.choice(ADV) for _ in range(rng.randint(1, 48))))
        docs = corpus_sample(0x5EED0000 + mix, mix, 96 << 10)
        for i, d in enumerate(docs):
            add(f"corpus{i}", d)
        other = corpus_sample(0x5EED0010, 0 if mix else 1, 32 << 10)
        for i, d in enumerate(other):
            add(f"corpusx{i}", d)
        # special-token cases (lib.rs:375-442)
        sp = list(specials)
        sp_texts = ["<|endoftext|>", "hello <|endoftext|>", "<|endoftext|> hello <|fim_prefix|> there <|fim_middle|>",
                    "a <|endoftext|>  <|endoftext|>\n<|endofprompt|> b", "<|endoftext", "<|endoftext|><|endoftext|>",
                    "x  <|endoftext|>", "x \n<|endoftext|>y", "<<|endoftext|>>", "tail  "]
        for i, t in enumerate(sp_texts):
            add(f"special_all{i}", t, set(sp))
            add(f"special_none{i}", t, set())
            add(f"special_first{i}", t, {sp[0]})
            if len(sp) > 1:
                add(f"special_last{i}", t, {sp[-1]})
        for i in range(60):
            parts = [rng.choice(ADV + sp + sp) for _ in range(rng.randint(1, 24))]
            add(f"special_adv{i}", "".join(parts), set(rng.sample(sp, rng.randint(0, len(sp)))))
        # every vocabulary token must re-encode to itself through the pure merge (sample)
        toks = list(ranks.items())
        for tb, r in rng.sample(toks, 3000) + toks[:512]:
            assert edu.bpe_encode(ranks, tb, visualise=None) == [r], tb
        path = os.path.join(ROOT, "tests/golden", name + ".json.gz")
        payload = json.dumps({"encoding": name, "pat_str": pat_str, "vocab": vocab, "special_tokens": specials,
                              "generator": "tools/gen_golden.py (reference tiktoken/_educational.py bpe_encode + regex.findall; "
                                           "specials located by the reference's tiktoken.core._special_token_regex in the loop of src/lib.rs:386-402)",
                              "special_tie_rule": "no registered special of this fixture is a prefix of another (checked at generation); "
                                                  "there the reference's choice depends on hash-map order (lib.rs:625-631) and this "
                                                  "implementation takes the longest allowed special",
                              "cases": cases}, ensure_ascii=True).encode()
        with open(path, "wb") as f:
            with gzip.GzipFile(fileobj=f, mode="wb", mtime=0, compresslevel=9) as gz:
                gz.write(payload)
        ntok = sum(len(c["tokens"]) for c in cases)
        print(name, "cases", len(cases), "tokens", ntok, "pieces", stats, "file bytes", os.path.getsize(path))

    # A vocabulary trained by the reference's own `bpe_train` (tiktoken/_educational.py:119-185) on
    # its own source, as train_simple_encoding() does (:208-223): pins the oracle on ranks that no
    # code of ours produced.  The GPT-2 spelling of the pattern is used there; openai_public.py:9-14
    # declares it equivalent to r50k_pat_str (pattern id 0 of the scanners).
    src = open(os.path.join(REF, "tiktoken/_educational.py")).read()
    ranks = edu.bpe_train(data=src, vocab_size=600, pat_str=po.GPT2_ORIG_PAT, visualise=None)
    simple = edu.SimpleBytePairEncoding(pat_str=po.GPT2_ORIG_PAT, mergeable_ranks=ranks)
    toks = simple.encode("hello world", visualise=None)
    assert simple.decode_tokens_bytes(toks) == [b"hello", b" world"]  # _educational.py:218-221
    cases = []
    texts = ["hello world", src[:6000], "".join(reversed(src[:2000]))] + reference_texts()
    texts += ["".join(rng.choice(ADV) for _ in range(rng.randint(1, 48))) for _ in range(200)]
    for i, t in enumerate(texts):
        cases.append({"name": f"edu{i}", "text": base64.b64encode(t.encode("utf-8")).decode(), "allowed": None,
                      "tokens": simple.encode(t, visualise=None)})
    payload = json.dumps({"encoding": "edu600", "pat_str": po.R50K_PAT, "vocab": None, "special_tokens": {},
                          "mergeable_ranks": [[base64.b64encode(k).decode(), v] for k, v in ranks.items()],
                          "generator": "tools/gen_golden.py (reference bpe_train + SimpleBytePairEncoding.encode)",
                          "cases": cases}).encode()
    path = os.path.join(ROOT, "tests/golden", "edu600.json.gz")
    with open(path, "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0, compresslevel=9) as gz:
            gz.write(payload)
    print("edu600 cases", len(cases), "file bytes", os.path.getsize(path))
    print("done")


if __name__ == "__main__":
    main()
