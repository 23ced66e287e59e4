"""ctypes binding of oracle/libtk_oracle.so (the C parity oracle).  TEST INFRASTRUCTURE ONLY.

Builds the library with `make -C oracle` on first use.  See oracle/tk_oracle.h for what is
restated and from which reference lines.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libtk_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libtk_oracle.so")
        src = os.path.join(_HERE, "tk_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        L = ctypes.CDLL(path)
        vp, u64, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
        L.tko_vocab_new.restype = vp
        L.tko_vocab_new.argtypes = [vp, vp, vp, u64, vp, vp, vp, u64, i32]
        L.tko_vocab_free.argtypes = [vp]
        L.tko_split.restype = ctypes.c_int64
        L.tko_split.argtypes = [i32, vp, u64, vp, u64]
        L.tko_encode_piece.restype = ctypes.c_int64
        L.tko_encode_piece.argtypes = [vp, vp, u64, vp, u64]
        L.tko_encode_ordinary.restype = ctypes.c_int64
        L.tko_encode_ordinary.argtypes = [vp, vp, u64, vp, u64]
        L.tko_encode.restype = ctypes.c_int64
        L.tko_encode.argtypes = [vp, vp, u64, vp, u64, vp, u64]
        L.tko_last_encode_seconds.restype = ctypes.c_double
        L.tko_encode_batch.restype = i32
        L.tko_encode_batch.argtypes = [vp, vp, vp, u64, i32, vp, u64, i32, vp, vp]
        _LIB = L
    return _LIB


def _pack(items):
    blob = b"".join(k for k, _ in items)
    off = np.zeros(len(items) + 1, np.uint64)
    if items:
        off[1:] = np.cumsum([len(k) for k, _ in items], dtype=np.uint64)
    ids = np.array([v for _, v in items], np.uint32)
    return np.frombuffer(blob, np.uint8) if blob else np.zeros(1, np.uint8), off, ids


class COracle:
    """One (pattern, mergeable_ranks, special_tokens) triple, i.e. what CoreBPE::new_internal
    (src/lib.rs:618-663) receives."""

    def __init__(self, pattern: int, mergeable_ranks: dict[bytes, int], special_tokens: dict[str, int] | None = None):
        self.pattern = pattern
        self.specials = dict(special_tokens or {})
        rb, ro, ri = _pack(list(mergeable_ranks.items()))
        sb, so, si = _pack([(k.encode("utf-8"), v) for k, v in self.specials.items()])
        self._keep = (rb, ro, ri, sb, so, si)
        self._h = lib().tko_vocab_new(rb.ctypes.data, ro.ctypes.data, ri.ctypes.data, len(ri),
                                      sb.ctypes.data, so.ctypes.data, si.ctypes.data, len(si), pattern)

    def __del__(self):
        try:
            lib().tko_vocab_free(self._h)
        except Exception:
            pass

    def split(self, data: bytes) -> list[int]:
        buf = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
        ends = np.empty(max(len(data), 1), np.uint64)
        n = lib().tko_split(self.pattern, buf.ctypes.data, len(data), ends.ctypes.data, len(ends))
        assert n >= 0
        return ends[:n].tolist()

    def encode_piece(self, piece: bytes) -> list[int]:
        buf = np.frombuffer(piece, np.uint8) if piece else np.zeros(1, np.uint8)
        out = np.empty(max(len(piece), 1), np.uint32)
        n = lib().tko_encode_piece(self._h, buf.ctypes.data, len(piece), out.ctypes.data, len(out))
        return out[:n].tolist()

    def encode_ordinary(self, data: bytes) -> np.ndarray:
        buf = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
        out = np.empty(max(len(data), 1), np.uint32)
        n = lib().tko_encode_ordinary(self._h, buf.ctypes.data, len(data), out.ctypes.data, len(out))
        assert n >= 0
        return out[:n].copy()

    def _allowed_ids(self, allowed_special) -> np.ndarray:
        if allowed_special == "all":
            allowed_special = set(self.specials)
        ids = [self.specials[s] for s in allowed_special if s in self.specials]
        return np.array(ids if ids else [0], np.uint32), len(ids)

    def encode(self, data: bytes, allowed_special) -> np.ndarray:
        buf = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
        out = np.empty(max(len(data), 1), np.uint32)
        ids, n_ids = self._allowed_ids(allowed_special)
        n = lib().tko_encode(self._h, buf.ctypes.data, len(data), ids.ctypes.data, n_ids, out.ctypes.data, len(out))
        assert n >= 0
        return out[:n].copy()

    def encode_batch(self, blob: np.ndarray, doc_off: np.ndarray, allowed_special=None, n_threads: int = 1, out=None):
        """blob: uint8 array of packed documents; doc_off: uint64[n_docs+1].  Returns (tokens, tok_off).
        `out` = (tokens uint32[>= total bytes], tok_off uint64[n_docs+1]) reuses caller buffers (so a timed
        call does not pay first-touch page faults)."""
        blob = np.ascontiguousarray(blob, np.uint8)
        doc_off = np.ascontiguousarray(doc_off, np.uint64)
        n_docs = len(doc_off) - 1
        total = int(doc_off[-1])
        if out is not None:
            tokens, tok_off = out
            assert len(tokens) >= max(total, 1) and len(tok_off) == n_docs + 1
        else:
            tokens = np.empty(max(total, 1), np.uint32)
            tok_off = np.empty(n_docs + 1, np.uint64)
        if allowed_special is None:
            mode, ids, n_ids = 0, np.zeros(1, np.uint32), 0
        else:
            mode = 1
            ids, n_ids = self._allowed_ids(allowed_special)
        b = blob if len(blob) else np.zeros(1, np.uint8)
        rc = lib().tko_encode_batch(self._h, b.ctypes.data, doc_off.ctypes.data, n_docs, mode, ids.ctypes.data, n_ids,
                                    n_threads, tokens.ctypes.data, tok_off.ctypes.data)
        assert rc == 0
        return tokens[: int(tok_off[-1])], tok_off


def last_encode_seconds() -> float:
    """Wall time of the parallel per-document phase of the last COracle.encode_batch call (its packing pass excluded)."""
    return float(lib().tko_last_encode_seconds())
