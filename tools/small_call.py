#!/usr/bin/env python3
"""Latency of small calls (the reference's most common use): microseconds per call, median of many."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tiktoken_amd

def med(f, n=2000, warm=200):
    for _ in range(warm): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter_ns(); f(); ts.append(time.perf_counter_ns() - t0)
    ts.sort()
    return ts[len(ts) // 2] / 1e3, ts[len(ts) // 10] / 1e3, ts[(len(ts) * 9) // 10] / 1e3

enc = tiktoken_amd.get_encoding(sys.argv[1] if len(sys.argv) > 1 else "o200k_shaped")
core = enc._core_bpe
for text in ["hello world", "The quick brown fox jumps over the lazy dog. " * 4, "lorem ipsum dolor sit amet " * 70]:
    b = text.encode()
    buf = np.frombuffer(b, np.uint8)
    off = (ctypes.c_uint64 * 2)(0, len(b))
    L = core._L
    def raw():
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        L.tk_encode_ordinary(core._h, buf.ctypes.data, len(b), ctypes.byref(out), ctypes.byref(n))
        L.tk_free(out)
    r = {"Encoding.encode": med(lambda: enc.encode(text)), "Encoding.encode_ordinary": med(lambda: enc.encode_ordinary(text)),
         "CoreBPE.encode_ordinary": med(lambda: core.encode_ordinary(text)), "C ABI tk_encode_ordinary (ctypes)": med(raw)}
    print(f"{len(b)} bytes, {len(enc.encode(text))} tokens:")
    for k, (m, lo, hi) in r.items():
        print(f"   {k:36} median {m:8.1f} us   p10 {lo:8.1f}   p90 {hi:8.1f}")
# C1: 1 MiB lorem, one document (general pipeline)
import helpers as h
_, _, _, blob, off, _ = h.baseline_config("C1")
g = tiktoken_amd.get_encoding("gpt2_shaped")
s = blob.tobytes().decode()
print("C1 (1 MiB, gpt2_shaped) Encoding.encode_ordinary:", med(lambda: g.encode_ordinary(s), n=30, warm=3))
print("C1 CoreBPE._encode_np:", med(lambda: g._core_bpe._encode_np(blob.tobytes(), None), n=30, warm=3))

# several threads on one Encoding (core.py:175): small calls take no lock (a slot each); calls per second through the C ABI and through
# Encoding.encode_ordinary (the latter holds the GIL for its Python part)
import threading
text = ("The quick brown fox jumps over the lazy dog; 3.14159 and so on, ünïcödé too. " * 3)[:180]
b = text.encode(); buf = np.frombuffer(b, np.uint8); L = core._L
def raw_call():
    out, n = ctypes.c_void_p(), ctypes.c_uint64()
    L.tk_encode_ordinary(core._h, buf.ctypes.data, len(b), ctypes.byref(out), ctypes.byref(n))
    L.tk_free(out)
def rate(f, nth, total=24000):
    def work():
        for _ in range(total // nth): f()
    for _ in range(200): f()
    th = [threading.Thread(target=work) for _ in range(nth)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return total / (time.perf_counter() - t0)
print(f"threads on one Encoding, {len(b)}-byte calls (calls per second):")
base_raw = base_py = None
for nth in (1, 2, 4, 8, 16, 32):
    l0, c0 = core.stat("small_launches"), core.stat("small_calls")
    r_raw = rate(raw_call, nth)
    l1, c1 = core.stat("small_launches"), core.stat("small_calls")
    r_py = rate(lambda: enc.encode_ordinary(text), nth)
    base_raw = base_raw or r_raw; base_py = base_py or r_py
    print(f"   {nth:2d} threads: C ABI {r_raw:9.0f}/s ({r_raw / base_raw:4.2f}x, {(c1 - c0) / max(l1 - l0, 1):4.2f} calls per launch)   "
          f"Encoding.encode_ordinary {r_py:9.0f}/s ({r_py / base_py:4.2f}x)")

# the same from NATIVE threads (tools/ubench/small_threads.c: no interpreter in the loop): what the C ABI itself scales to
import subprocess, tempfile
so = os.path.join(tempfile.gettempdir(), "tk_small_threads.so")
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", os.path.join(ROOT, "tools", "ubench", "small_threads.c"), "-o", so])
H = ctypes.CDLL(so)
H.tk_small_threads.restype = ctypes.c_double
H.tk_small_threads.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                               ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
enc_p = ctypes.cast(L.tk_encode_ordinary, ctypes.c_void_p)
free_p = ctypes.cast(L.tk_free, ctypes.c_void_p)
want = len(enc.encode_ordinary(text))
print(f"native threads on one core, {len(b)}-byte calls (calls per second through the C ABI):")
base = None
for nth in (1, 2, 4, 8, 16, 32, 64):
    l0, c0 = core.stat("small_launches"), core.stat("small_calls")
    tok, bad = ctypes.c_uint64(), ctypes.c_int()
    per = max(20000 // nth, 500)
    r = H.tk_small_threads(enc_p, free_p, core._h, buf.ctypes.data, len(b), nth, per, ctypes.byref(tok), ctypes.byref(bad))
    l1, c1 = core.stat("small_launches"), core.stat("small_calls")
    base = base or r
    ok = bad.value == 0 and tok.value == want * nth * per
    print(f"   {nth:2d} threads: {r:9.0f}/s ({r / base:5.2f}x, {(c1 - c0) / max(l1 - l0, 1):4.2f} calls per launch, results {'ok' if ok else 'WRONG'})")
