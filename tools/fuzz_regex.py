#!/usr/bin/env python3
"""A fuzz campaign on the CPU for the generic pat_str engine: the two generated-pattern tests of tests/test_regex_engine.py (random patterns over
the whole supported syntax against Python `regex`, program and table form) with OTHER seeds.  usage: python tools/fuzz_regex.py OFFSET [OFFSET ...]"""
import os, sys, random, types, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_regex_engine as t

rc = 0
for off in [int(a) for a in sys.argv[1:]]:
    shim = types.SimpleNamespace(**{k: getattr(random, k) for k in dir(random) if not k.startswith("__")})
    shim.Random = lambda seed=None, _off=off: random.Random(None if seed is None else seed + _off)
    t.random = shim
    fns = [t.test_generated_patterns_equal_python_regex, t.test_generated_patterns_in_table_form_equal_python_regex]
    if os.environ.get("FUZZ_FIXED_PATTERNS"):  # the hand-written pattern list on other texts (and with special tokens at other places) instead
        import functools
        which = [i for i in range(len(t.PATTERNS))]
        fns = [functools.partial(f, i) for i in which for f in (t.test_split_equals_python_regex, t.test_special_tokens_at_random_places)]
        for f in fns:
            f.__name__ = f"{f.func.__name__}[{f.args[0]}]"
    for fn in fns:
        try:
            fn()
            print(f"offset {off}: {fn.__name__} ok", flush=True)
        except LookupError:
            print(f"offset {off}: {fn.__name__} skipped (the pattern leaves gaps on these texts)", flush=True)
        except AssertionError as e:
            # (the tests' closing assertions are about the yield of the seed's patterns -- how many compile, how many have a table; a mismatch shows the pattern)
            msg = str(e)
            print(f"offset {off}: {fn.__name__} ASSERTION {msg[:1500]}", flush=True)
            if "(" in msg and "," in msg and len(msg) < 60:
                continue
            rc = 1
        except Exception:
            traceback.print_exc()
            rc = 1
sys.exit(rc)
