#!/usr/bin/env python3
"""Writes tests/golden/reference_call_sites.json: every place where the reference's Python layer calls into its native core
(`self._core_bpe.<method>(...)`, `_tiktoken.CoreBPE(...)` in /root/reference/tiktoken/core.py) as [line, method, positional arity, keywords].

Data derived from the reference's AST, not its text: the GPU box has no /root/reference, and tests/test_reference_package.py uses this list
there to drive the shim in the forms the reference uses.  Runs only in the build container.   Usage: python tools/gen_call_sites.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from test_reference_package import CALL_SITES, REF, core_bpe_call_sites  # noqa: E402

sites = core_bpe_call_sites()
with open(CALL_SITES, "w") as f:
    f.write('{"source": "tiktoken/core.py of openai/tiktoken 0.14.0 (%s)", "fields": ["line", "method", "n_positional", "keywords"],\n "call_sites": [\n  ' % REF)
    f.write(",\n  ".join(json.dumps(s) for s in sites) + "\n ]}\n")
print(len(sites), "call sites ->", CALL_SITES)
