import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, helpers as h
import tiktoken_amd
enc = tiktoken_amd.get_encoding("o200k_shaped")
C = h.c_oracle_for("o200k_shaped")
for ch in ["x", "0", "^", " ", "\n", "a1", "Ab", "中"]:
    for n in (100_000, 1_000_000):
        s = ch * (n // len(ch))
        t0 = time.perf_counter(); toks = enc.encode_ordinary(s); dt = time.perf_counter() - t0
        ok = np.array_equal(np.asarray(toks, np.uint32), C.encode_ordinary(s.encode())) if n <= 100_000 or ch in "x0" else None
        print(f"{ch!r:6} x {n:>9}: {dt*1e3:9.1f} ms  tokens {len(toks):>8}  parity {ok}", flush=True)
