#!/usr/bin/env python3
"""Generate tests/golden/generic_patterns.json.gz: pat_str outside the scanner families, encoded by the REFERENCE's own Python code.

For every pattern the reference's `tiktoken._educational.SimpleBytePairEncoding(pat_str=..., mergeable_ranks=...)` (reference
tiktoken/_educational.py:12-37: `regex.compile(pat_str).findall(text)`, then `bpe_encode` per piece) encodes the texts with the 600-token
vocabulary that the reference's own `bpe_train` produced (tests/golden/edu600.json.gz).  Only patterns whose spelling means the same in
Python `regex` and in fancy-regex are used (no unscoped `$`, no set operations -- those are compared with `regex` in
tests/test_regex_engine.py), and only texts the pattern covers completely (the reference drops unmatched text, this library refuses it).

Runs only in the build container (needs /root/reference).  Usage: python tools/gen_golden_generic.py
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
stub = types.ModuleType("tiktoken._tiktoken")
stub.CoreBPE = type("CoreBPE", (), {"__init__": lambda self, *a, **k: None})
sys.modules["tiktoken._tiktoken"] = stub
sys.path.insert(0, REF)
import regex  # noqa: E402
import tiktoken  # noqa: E402
import tiktoken._educational as edu  # noqa: E402

assert tiktoken.__file__.startswith(REF), tiktoken.__file__
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as h  # noqa: E402
from test_regex_engine import PATTERNS, random_text  # noqa: E402


def covered(pat, text):
    at = 0
    for m in pat.finditer(text):
        if m.start() != at or m.end() == m.start():
            return False
        at = m.end()
    return at == len(text)


def main():
    ranks = h.golden_vocab("edu600")
    rng = random.Random(0x6E0E41C)
    texts = ["hello world", "parseHTTPRequest42 isDone\n\n  fooBar_baz 1234567", "Hello, World! It's 2024-01-02; naïve café — 你好世界 こんにちは привет\n\n\tx = y**2  # 12345678\r\n",
             "", " ", "\n", "a", "'s", "don't DON'T x'll", "   \n  \n", "αβγ ΑΒΓ абв", "0" * 17, "x" * 300, " " * 200 + "y"]
    texts += [random_text(rng, rng.choice([1, 3, 10, 40, 150])) for _ in range(160)]
    texts += [h.fuzz_doc(rng)[:1200] for _ in range(16)]
    out = []
    for idx, (pat_str, py) in enumerate(PATTERNS):
        if py is not None or idx < 5:  # (a different spelling for Python, or a pattern of the scanner families: fixtures of their own)
            continue
        simple = edu.SimpleBytePairEncoding(pat_str=pat_str, mergeable_ranks=ranks)
        cases = []
        for t in texts:
            if covered(simple._pat, t):
                cases.append({"text": base64.b64encode(t.encode()).decode(), "tokens": simple.encode(t, visualise=None)})
        out.append({"pattern_index": idx, "pat_str": pat_str, "cases": cases})
        print(idx, pat_str[:50], len(cases), "cases", sum(len(c["tokens"]) for c in cases), "tokens")
    payload = json.dumps({"vocab": "edu600 (tests/golden/edu600.json.gz: trained by the reference's bpe_train)",
                          "generator": "tools/gen_golden_generic.py (reference tiktoken/_educational.py SimpleBytePairEncoding.encode)",
                          "patterns": out}).encode()
    path = os.path.join(ROOT, "tests", "golden", "generic_patterns.json.gz")
    with open(path, "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0, compresslevel=9) as gz:
            gz.write(payload)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
