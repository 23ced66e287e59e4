#!/usr/bin/env python3
"""A fuzz campaign on the CPU for the scanner families: tests/test_patterns.py::test_generated_patterns_equal_regex (patterns generated from random
parameters -- contraction list, case sensitivity, digit group, suffix set, white-space rules, equivalent spellings -- against Python `regex`, through the
hand-written scanners of the host simulation or, where the family parser refuses, the generic engine) with OTHER seeds.
usage: python tools/fuzz_patterns.py OFFSET [OFFSET ...]"""
import os, sys, random, types, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_patterns as t

rc = 0
for off in [int(a) for a in sys.argv[1:]]:
    shim = types.SimpleNamespace(**{k: getattr(random, k) for k in dir(random) if not k.startswith("__")})
    shim.Random = lambda seed=None, _off=off: random.Random(None if seed is None else seed + _off)
    t.random = shim
    try:
        t.test_generated_patterns_equal_regex()
        print(f"offset {off}: ok", flush=True)
    except AssertionError as e:
        msg = str(e)
        print(f"offset {off}: ASSERTION {msg[:3000]}", flush=True)
        if not (msg.startswith("(") and len(msg) < 40):  # (the closing assertion counts how many patterns the family parser took)
            rc = 1
    except Exception:
        traceback.print_exc()
        rc = 1
sys.exit(rc)
