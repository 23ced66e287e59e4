"""Vocabulary files.  The `.tiktoken` wire format -- one `base64(token) SP rank` per line, reference
tiktoken/load.py:147-171 -- is parsed by the native library (`tk_parse_tiktoken_bpe`, include/tiktoken_amd.h) instead of a
per-line Python loop; the result keeps the packed arrays so that `CoreBPE(...)` hands them to `tk_create` without re-packing.
The GPT-2 `vocab.bpe` + `encoder.json` pair (load.py:89-144) is converted here as well.

Files are read from a local path, or -- for the URLs the stock constructors use -- from `$TIKTOKEN_CACHE_DIR` /
`$DATA_GYM_CACHE_DIR` under the reference's cache key `sha1(url)` (so an existing tiktoken cache is picked up), and fetched
over HTTP(S) only when absent.  A pinned sha256 is always verified.
"""
from __future__ import annotations

import base64
import ctypes
import gzip
import hashlib
import json
import os

import numpy as np

from . import _lib


class RankTable(dict):
    """`dict[bytes, int]` as parsed from a `.tiktoken` file.  It carries the packed (blob, offsets, ranks) arrays it was parsed from
    -- what `tk_create` takes, so building an Encoding never walks 200 000 Python objects -- and fills the dict itself from them on
    first use (a fifth of a second for the o200k file; an `Encoding` needs only the count and the largest rank).  Any mutation drops
    the arrays, so a CoreBPE built from the table always sees the dict's current contents.  C code that reads a dict's storage
    directly (`PyDict_Next`, PyO3's HashMap extraction) sees an EMPTY dict until the table is filled, so the lazy form is an internal
    fast path: `parse_tiktoken_bpe` / `load_tiktoken_bpe` return a filled table unless asked with `lazy=True` (the constructors under
    tiktoken_ext do; `Encoding._mergeable_ranks` fills it before handing it out)."""

    packed = None  # (blob uint8[], off uint64[n+1], ranks uint32[n]) while they describe the dict exactly
    _pending = None  # the same arrays until the dict has been filled from them
    _distinct = False  # tk_create has taken the arrays: no token is listed twice (the count is known without the dict)

    @classmethod
    def from_packed(cls, blob: bytes, off: np.ndarray, ids: np.ndarray) -> "RankTable":
        t = cls()
        t._pending = (blob, off, ids)
        t.packed = (np.frombuffer(blob, np.uint8) if blob else np.zeros(1, np.uint8), off, ids)
        return t

    def materialize(self) -> "RankTable":
        pend, self._pending = self._pending, None
        if pend is not None:
            blob, off, ids = pend
            bounds = off.tolist()
            dict.update(self, zip((blob[a:b] for a, b in zip(bounds[:-1], bounds[1:])), ids.tolist()))
            if dict.__len__(self) != len(ids):  # (duplicate keys collapse in the dict: then the packed form no longer matches it)
                self.packed = None
        return self

    # (what an Encoding asks for without needing the dict)
    def max_rank(self) -> int:
        if self.packed is not None:
            return int(self.packed[2].max()) if len(self.packed[2]) else 0
        return max(self.values())

    def __len__(self):
        if self._pending is not None and self._distinct:
            return len(self._pending[2])
        return dict.__len__(self.materialize())

    def _reading(name):  # noqa: N805
        def method(self, *a, **k):
            # (dict's own comparison and union read the OTHER operand's storage at C level: fill that one too)
            a = tuple(x.materialize() if isinstance(x, RankTable) else x for x in a)
            return getattr(dict, name)(self.materialize(), *a, **k)

        method.__name__ = name
        return method

    def _mutating(name):  # noqa: N805
        def method(self, *a, **k):
            self.materialize()
            self.packed = None
            return getattr(dict, name)(self, *a, **k)

        method.__name__ = name
        return method

    for _n in ("__getitem__", "__contains__", "__iter__", "__reversed__", "__eq__", "__ne__", "__repr__", "__or__", "__ror__", "keys", "values", "items", "get",
               "copy", "__sizeof__"):
        locals()[_n] = _reading(_n)
    for _n in ("__setitem__", "__delitem__", "pop", "popitem", "clear", "update", "setdefault", "__ior__"):
        locals()[_n] = _mutating(_n)
    del _n, _mutating, _reading
    __hash__ = None

    def __reduce__(self):
        return (dict, (dict(self.materialize()),))


def parse_tiktoken_bpe(contents: bytes, source: str = "<bytes>", *, lazy: bool = False) -> RankTable:
    """`lazy=True`: the dict is filled on first use at Python level (RankTable) -- for tables that go straight into an Encoding."""
    L = _lib.lib()
    buf = np.frombuffer(contents, np.uint8) if contents else np.zeros(1, np.uint8)
    pb, po, pi, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
    rc = L.tk_parse_tiktoken_bpe(buf.ctypes.data, len(contents), ctypes.byref(pb), ctypes.byref(po), ctypes.byref(pi), ctypes.byref(n))
    if rc != _lib.TK_OK:
        raise ValueError(f"{_lib.last_error()} ({source})")
    try:
        cnt = n.value
        off = np.ctypeslib.as_array(ctypes.cast(po, ctypes.POINTER(ctypes.c_uint64)), shape=(cnt + 1,)).copy()
        ids = np.ctypeslib.as_array(ctypes.cast(pi, ctypes.POINTER(ctypes.c_uint32)), shape=(max(cnt, 1),))[:cnt].copy()
        blob = ctypes.string_at(pb, int(off[-1]))
    finally:
        for p in (pb, po, pi):
            L.tk_free(p)
    table = RankTable.from_packed(blob, off, ids)
    return table if lazy else table.materialize()


def dump_tiktoken_bpe(bpe_ranks: dict[bytes, int], tiktoken_bpe_file: str) -> None:
    with open(tiktoken_bpe_file, "wb") as f:
        for token, rank in sorted(bpe_ranks.items(), key=lambda kv: kv[1]):
            f.write(base64.b64encode(token) + b" %d\n" % rank)


def _cache_root() -> str | None:
    for var in ("TIKTOKEN_CACHE_DIR", "DATA_GYM_CACHE_DIR"):
        if var in os.environ:
            return os.environ[var] or None  # empty string: caching switched off
    import tempfile

    return os.path.join(tempfile.gettempdir(), "data-gym-cache")


def fetch(location: str, expected_hash: str | None = None) -> bytes:
    """Bytes of a local file or of an http(s) URL (cache first); raises ValueError when the sha256 does not match."""
    def verified(data: bytes, where: str) -> bytes:
        if expected_hash and hashlib.sha256(data).hexdigest() != expected_hash:
            raise ValueError(f"Hash mismatch for data from {where} (expected sha256 {expected_hash})")
        return data

    if "://" not in location:
        with open(location, "rb") as f:
            return verified(f.read(), location)
    root = _cache_root()
    slot = os.path.join(root, hashlib.sha1(location.encode()).hexdigest()) if root else None
    if slot and os.path.exists(slot):
        with open(slot, "rb") as f:
            data = f.read()
        if not expected_hash or hashlib.sha256(data).hexdigest() == expected_hash:
            return data
    if not location.startswith(("http://", "https://")):
        raise ValueError(f"unsupported location scheme: {location}")
    import urllib.request

    with urllib.request.urlopen(location) as resp:  # noqa: S310 (pinned by sha256 below)
        data = verified(resp.read(), location)
    if slot:
        os.makedirs(root, exist_ok=True)
        tmp = f"{slot}.{os.getpid()}.part"
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, slot)
    return data


def load_tiktoken_bpe(tiktoken_bpe_file: str, expected_hash: str | None = None, *, lazy: bool = False) -> RankTable:
    contents = fetch(tiktoken_bpe_file, expected_hash)
    if tiktoken_bpe_file.endswith(".gz"):
        contents = gzip.decompress(contents)
    return parse_tiktoken_bpe(contents, tiktoken_bpe_file, lazy=lazy)


def data_gym_byte_order() -> list[int]:
    """Printable non-space bytes first, then the rest: the GPT-2 byte <-> rank convention."""
    order = [b for b in range(256) if chr(b).isprintable() and chr(b) != " "]
    return order + [b for b in range(256) if b not in order]


def data_gym_to_mergeable_bpe_ranks(vocab_bpe_file: str, encoder_json_file: str, vocab_bpe_hash: str | None = None,
                                    encoder_json_hash: str | None = None, clobber_one_byte_tokens: bool = False) -> dict[bytes, int]:
    rank_to_byte = data_gym_byte_order()
    n_printable = sum(1 for b in range(256) if chr(b).isprintable() and chr(b) != " ")
    char_to_byte = {chr(b): b for b in rank_to_byte[:n_printable]}
    for i, b in enumerate(rank_to_byte[n_printable:]):
        char_to_byte[chr(256 + i)] = b

    def to_bytes(s: str) -> bytes:
        return bytes(char_to_byte[ch] for ch in s)

    merges_text = fetch(vocab_bpe_file, vocab_bpe_hash).decode()
    merges = [tuple(line.split()) for line in merges_text.split("\n")[1:-1]]
    ranks = {bytes([b]): i for i, b in enumerate(rank_to_byte)}
    n = len(ranks)
    for first, second in merges:  # (a counter, not len(ranks): a merge listed twice takes a number both times -- load.py:116-119)
        ranks[to_bytes(first) + to_bytes(second)] = n
        n += 1
    encoder = {to_bytes(k): v for k, v in json.loads(fetch(encoder_json_file, encoder_json_hash)).items()}
    encoder.pop(b"<|endoftext|>", None)
    encoder.pop(b"<|startoftext|>", None)
    if clobber_one_byte_tokens:
        for k, v in encoder.items():
            if len(k) == 1:
                ranks[k] = v
    assert ranks == encoder  # merge order must equal token index order
    return ranks
