#!/usr/bin/env python3
"""Where the time of a generic-pat_str call goes: wall time per step and the library's per-kernel times (tk_set_profiling)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
T0 = time.perf_counter()
def say(*a):
    print("[%6.2f]" % (time.perf_counter() - T0), *a, flush=True)
import helpers as h
from tiktoken_amd import CoreBPE
say("imports done")
NAMES = ["tk_k_mark_docs", "tk_k_rx_speculate", "tk_k_rx_link", "tk_k_rx_resolve", "tk_k_rx_merge", "tk_k_front", "tk_k_front_slow", "tk_k_bincount", "tk_k_binfill",
         *[f"tk_k_merge_llane_{i}" for i in (16, 24, 32, 48, 64)], *[f"tk_k_merge_group_{i}" for i in (8, 16, 32, 64)], "tk_k_merge_rounds",
         "tk_k_merge_rounds_wide", "tk_k_merge_long", "tk_k_place", "tk_k_docoff", "tk_k_count", "tk_k_emit", "tk_k_scan_small"]
pat = sys.argv[1] if len(sys.argv) > 1 else r"\w+|[^\w\s]+|\s+"
vocab = h.golden_vocab("o200k_shaped")
core = CoreBPE(vocab, {}, pat)
say("core created for", pat)
core.set_profiling(True)
rng = random.Random(3)
words = ["hello ", "World", " 12345", "\n", " ", "中文", "é", "...", "CamelCase", " don't", "\r\n\r\n", "3.14 "]
def batch(kind):
    if kind == "words":
        docs = ["".join(rng.choice(words) for _ in range(rng.randrange(1, 3000))) for _ in range(300)]
    elif kind == "runs":
        docs = ["".join(rng.choice(words + ["x" * 5000, " " * 900]) for _ in range(3000)) for _ in range(2)]
    else:
        docs = [h.fuzz_doc(rng)[:60000] for _ in range(200)]
    return h.pack([d.encode() for d in docs])
for kind in sys.argv[2].split(",") if len(sys.argv) > 2 else ["words", "runs", "fuzz"]:
    blob, off = batch(kind)
    say(kind, "batch:", len(blob), "bytes,", len(off) - 1, "docs")
    for what in ("pretokenize", "encode", "encode"):
        core.reset_kernel_ms()
        t = time.perf_counter()
        try:
            r = core.pretokenize_packed(blob, off) if what == "pretokenize" else core.encode_batch_packed(blob, off)[0]
        except Exception as e:
            say(kind, what, "raised", type(e).__name__, str(e)[:200])
            continue
        dt = time.perf_counter() - t
        ks = {k: core.kernel_ms(k) for k in NAMES}
        say(kind, what, "%.1f ms wall, %d items;" % (dt * 1e3, len(r)), " ".join("%s=%.2f" % (k.replace("tk_k_", ""), v[0]) for k, v in ks.items() if v[1]))
