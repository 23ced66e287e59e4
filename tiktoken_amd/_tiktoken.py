"""`CoreBPE` -- the drop-in for the reference's Rust extension class `tiktoken._tiktoken.CoreBPE`
(reference src/py.rs:13-184), implemented as a thin ctypes shim over the HIP library.

Same constructor and the same eleven methods, same exception types.  Two additive entry points,
`encode_batch_packed` and `pretokenize_packed`, expose the batch shape the GPU actually runs
(one launch sequence per batch instead of one FFI call per document).
"""
from __future__ import annotations

import atexit
import ctypes
import weakref
from typing import AbstractSet, Sequence

import numpy as np

from . import _lib

# Cores that are still alive when the interpreter shuts down are destroyed by an atexit handler -- i.e. BEFORE the HIP runtime's own static
# destructors run at process exit.  A tk_destroy from a late __del__ (module teardown, or later still) would free streams and page-locked
# buffers of a runtime that is already gone.
_live_cores: "weakref.WeakSet[CoreBPE]" = weakref.WeakSet()


@atexit.register
def _destroy_live_cores():
    for core in sorted(_live_cores, key=lambda c: getattr(c, "_group", None) is None):  # (a group before the replicas it refers to)
        try:
            core.close()
        except Exception:
            pass

# char::is_whitespace (Rust, lib.rs:583) = the Unicode White_Space property
_WHITE_SPACE = frozenset(map(chr, [9, 10, 11, 12, 13, 32, 0x85, 0xA0, 0x1680, *range(0x2000, 0x200B), 0x2028, 0x2029, 0x202F, 0x205F, 0x3000]))


def _pack_pairs(items: Sequence[tuple[bytes, int]]):
    blob = b"".join(k for k, _ in items)
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    if items:
        np.cumsum(np.fromiter((len(k) for k, _ in items), dtype=np.uint64, count=len(items)), out=off[1:])
    ids = np.fromiter((v for _, v in items), dtype=np.uint32, count=len(items))
    data = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(1, dtype=np.uint8)
    return data, off, ids if len(items) else np.zeros(1, dtype=np.uint32)


class _OwnedBuffer:
    """A library-owned result buffer (page-locked host memory for large results) seen through the array interface: numpy views it in
    place, and it goes back to the library (tk_free) when the last view is gone.  The counterpart of the reference's TiktokenBuffer
    (src/py.rs:186-249)."""

    def __init__(self, ptr: int, n: int, typestr: str = "<u4"):
        self._ptr = ptr
        self.__array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, True), "version": 3}

    def __del__(self):
        ptr, self._ptr = getattr(self, "_ptr", None), None
        if ptr:
            try:
                _lib.lib().tk_free(ptr)
            except Exception:
                pass


def _take_u32(ptr: ctypes.c_void_p, n: int) -> np.ndarray:
    """numpy array of a library-owned uint32 result: large ones in place (no second copy), small ones copied and released."""
    if not n:
        _lib.lib().tk_free(ptr)
        return np.zeros(0, dtype=np.uint32)
    if n >= (1 << 16):
        return np.asarray(_OwnedBuffer(ptr.value, n))
    out = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint32)), shape=(n,)).copy()
    _lib.lib().tk_free(ptr)
    return out


def _check_packed(blob: np.ndarray, doc_off: np.ndarray) -> None:
    """The C ABI reads doc_off[n_docs] bytes of the blob: refuse offsets that do not describe it."""
    if doc_off.ndim != 1 or len(doc_off) < 1:
        raise ValueError("doc_off must hold n_docs + 1 offsets (at least one)")
    if int(doc_off[0]) != 0 or int(doc_off[-1]) != len(blob):
        raise ValueError("doc_off[0] must be 0 and doc_off[-1] must equal len(blob)")
    if len(doc_off) > 1 and bool(np.any(doc_off[1:] < doc_off[:-1])):
        raise ValueError("doc_off must be non-decreasing")


def invalid_utf8_at(data: bytes) -> int | None:
    """Offset of the first byte of `data` that is not part of a well-formed UTF-8 char, or None (tk_validate_utf8: for callers that hand the
    C ABI bytes they cannot vouch for -- a Python str is valid by construction, as the reference's &str is)."""
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    pos = ctypes.c_uint64()
    return None if _lib.lib().tk_validate_utf8(buf.ctypes.data, len(data), ctypes.byref(pos)) == 0 else int(pos.value)


def default_devices() -> list[int]:
    """Devices a CoreBPE uses when none are named: $TIKTOKEN_AMD_DEVICES (comma-separated ordinals, or "all"), else device 0."""
    import os

    v = os.environ.get("TIKTOKEN_AMD_DEVICES", "").strip()
    if not v:
        return [0]
    if v == "all":
        return list(range(max(_lib.device_count(), 1)))
    return [int(x) for x in v.split(",") if x.strip()]


class CoreBPE:
    def __init__(self, encoder: dict[bytes, int], special_tokens_encoder: dict[str, int], pattern: str, *,
                 device: int | None = None, devices: Sequence[int] | None = None):
        """`devices`: several GPUs of this node (one replica of the tables per device; batches are split by documents into
        contiguous ranges of about equal byte counts -- tk_group_encode_batch).  A device may be named twice (virtual ranks)."""
        L = _lib.lib()
        if devices is None:
            devices = [device] if device is not None else default_devices()
        devices = list(devices)
        if not devices:
            raise ValueError("devices must name at least one GPU")
        device = devices[0]
        self._replicas: list[CoreBPE] = []
        self._group = None
        self._specials = dict(special_tokens_encoder)
        sb, so, si = _pack_pairs([(k.encode("utf-8"), v) for k, v in self._specials.items()])
        # vocab_io.RankTable: the arrays straight from the native parser (its dict may not even be filled yet: nothing here walks it)
        packed = getattr(encoder, "packed", None)
        fresh = packed is not None and getattr(encoder, "_pending", None) is not None
        h = ctypes.c_void_p()
        for attempt in (0, 1):
            if packed is not None and (fresh or len(packed[2]) == len(encoder)):
                rb, ro, ri = packed
            else:
                for v in encoder.values():
                    if not 0 <= v <= 0xFFFFFFFF:
                        raise OverflowError("rank does not fit in u32")  # PyO3 would refuse the conversion too
                rb, ro, ri = _pack_pairs(list(encoder.items()))
            rc = L.tk_create(rb.ctypes.data, ro.ctypes.data, ri.ctypes.data, len(ro) - 1, sb.ctypes.data, so.ctypes.data,
                             si.ctypes.data, len(self._specials), pattern.encode("utf-8"), device, ctypes.byref(h))
            if rc != _lib.TK_OK and fresh and attempt == 0 and "duplicate key" in _lib.last_error():
                # a file that lists a token twice: the dict keeps the later rank (as the reference's does) -- build from the dict
                encoder.materialize()
                packed, fresh = encoder.packed, False
                continue
            break
        _lib.raise_for(rc)
        if fresh:
            encoder._distinct = True  # (tk_create refuses a token listed twice)
        self._h = h
        self._L = L
        _live_cores.add(self)
        self.device = device
        self.devices = devices
        if len(devices) > 1:
            self._replicas = [CoreBPE(encoder, special_tokens_encoder, pattern, devices=[d]) for d in devices[1:]]
            handles = (ctypes.c_void_p * len(devices))(h, *[r._h for r in self._replicas])
            grp = ctypes.c_void_p()
            _lib.raise_for(L.tk_group_create(handles, len(devices), ctypes.byref(grp)))
            self._group = grp

    def __del__(self):
        self.close()

    def close(self):
        """Releases the native core (tables, workspace, streams).  Idempotent; the object is unusable afterwards."""
        g, self._group = getattr(self, "_group", None), None
        if g:
            try:
                self._L.tk_group_destroy(g)
            except Exception:
                pass
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._L.tk_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ helpers
    def _allowed_ids(self, allowed_special: AbstractSet[str]) -> tuple[np.ndarray, int]:
        if isinstance(allowed_special, str):
            if allowed_special != "all":
                raise ValueError("allowed_special must be a set of special-token strings or 'all'")
            allowed_special = self._specials.keys()
        ids = [self._specials[s] for s in allowed_special if s in self._specials]
        return np.asarray(ids if ids else [0], dtype=np.uint32), len(ids)

    @staticmethod
    def _as_u8(data: bytes) -> np.ndarray:
        return np.frombuffer(data, dtype=np.uint8) if data else np.zeros(1, dtype=np.uint8)

    def _encode_np(self, data: bytes, allowed_special: AbstractSet[str] | None) -> np.ndarray:
        buf = self._as_u8(data)
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        if allowed_special is None:
            rc = self._L.tk_encode_ordinary(self._h, buf.ctypes.data, len(data), ctypes.byref(out), ctypes.byref(n))
        else:
            ids, k = self._allowed_ids(allowed_special)
            rc = self._L.tk_encode(self._h, buf.ctypes.data, len(data), ids.ctypes.data, k, ctypes.byref(out), ctypes.byref(n))
        _lib.raise_for(rc)
        return _take_u32(out, n.value)

    # ------------------------------------------------------------------ encoding (src/py.rs:29-131)
    def encode_ordinary(self, text: str) -> list[int]:
        # str.encode raises UnicodeEncodeError on lone surrogates, as PyO3's &str extraction does;
        # Encoding.encode_ordinary relies on that to trigger its repair path (core.py:77-80)
        return self._encode_np(text.encode("utf-8"), None).tolist()

    def encode(self, text: str, allowed_special: AbstractSet[str]) -> list[int]:
        return self._encode_np(text.encode("utf-8"), allowed_special).tolist()

    def encode_to_tiktoken_buffer(self, text: str, allowed_special: AbstractSet[str]):
        """Object with the buffer protocol: read-only, 1-D, itemsize 4, format 'I' (src/py.rs:186-249)."""
        arr = self._encode_np(text.encode("utf-8"), allowed_special)
        arr.setflags(write=False)
        return arr

    def encode_batch_packed(self, blob: np.ndarray, doc_off: np.ndarray, allowed_special: AbstractSet[str] | None = None):
        """One GPU batch: `blob` = documents packed back to back (uint8), `doc_off` = uint64[n+1].
        Returns (tokens uint32[T], tok_off uint64[n+1])."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        _check_packed(blob, doc_off)
        n_docs = len(doc_off) - 1
        tok_off = np.empty(n_docs + 1, dtype=np.uint64)
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        src = blob if len(blob) else np.zeros(1, dtype=np.uint8)
        if allowed_special is None:
            ids, k, mode = np.zeros(1, dtype=np.uint32), 0, 0
        else:
            ids, k = self._allowed_ids(allowed_special)
            mode = 1
        if self._group is not None and n_docs > 1:
            rc = self._L.tk_group_encode_batch(self._group, src.ctypes.data, doc_off.ctypes.data, n_docs, mode, ids.ctypes.data, k,
                                               ctypes.byref(out), ctypes.byref(n), tok_off.ctypes.data)
        else:
            rc = self._L.tk_encode_batch(self._h, src.ctypes.data, doc_off.ctypes.data, n_docs, mode, ids.ctypes.data, k,
                                         ctypes.byref(out), ctypes.byref(n), tok_off.ctypes.data)
        _lib.raise_for(rc)
        return _take_u32(out, n.value), tok_off

    def encode_batch_gathered(self, blob: np.ndarray, doc_off: np.ndarray, allowed_special: AbstractSet[str] | None = None):
        """Multi-GPU batch with the token ids gathered on the FIRST device (peer copies over xGMI): returns
        (d_tokens_ptr, n_tokens, d_tok_off_ptr), device pointers owned by this CoreBPE and valid until its next call."""
        if self._group is None:
            raise ValueError("encode_batch_gathered needs a CoreBPE built with several devices")
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        _check_packed(blob, doc_off)
        if allowed_special is None:
            ids, k, mode = np.zeros(1, dtype=np.uint32), 0, 0
        else:
            ids, k = self._allowed_ids(allowed_special)
            mode = 1
        src = blob if len(blob) else np.zeros(1, dtype=np.uint8)
        dt, dn, do = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_void_p()
        rc = self._L.tk_group_encode_batch_device(self._group, src.ctypes.data, doc_off.ctypes.data, len(doc_off) - 1, mode, ids.ctypes.data, k,
                                                  ctypes.byref(dt), ctypes.byref(dn), ctypes.byref(do))
        _lib.raise_for(rc)
        return dt.value, dn.value, do.value

    def group_stat(self, name: str) -> int:
        return int(self._L.tk_group_stat(self._group, name.encode())) if self._group is not None else 0

    def encode_batch_device(self, d_text_ptr: int, n_bytes: int, d_doc_off_ptr: int, h_doc_off: np.ndarray | None,
                            n_docs: int, allowed_special: AbstractSet[str] | None = None, stream: int = 0):
        """Device-resident batch (tk_encode_batch_device): inputs already in HBM, results stay in HBM.
        Returns (d_tokens_ptr, n_tokens, d_tok_off_ptr); the pointers are owned by this CoreBPE and valid
        until its next encode call.  d_text must be readable 64 bytes past n_bytes."""
        dt, dn, do = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_void_p()
        if allowed_special is None:
            ids, k, mode = np.zeros(1, dtype=np.uint32), 0, 0
        else:
            ids, k = self._allowed_ids(allowed_special)
            mode = 1
        h_ptr = None
        if h_doc_off is not None:
            h_doc_off = np.ascontiguousarray(h_doc_off, dtype=np.uint64)
            h_ptr = h_doc_off.ctypes.data
        rc = self._L.tk_encode_batch_device(self._h, d_text_ptr, n_bytes, d_doc_off_ptr, h_ptr, n_docs, mode, ids.ctypes.data, k,
                                            stream or None, ctypes.byref(dt), ctypes.byref(dn), ctypes.byref(do))
        _lib.raise_for(rc)
        return dt.value, dn.value, do.value

    def pretokenize_packed(self, blob: np.ndarray, doc_off: np.ndarray, allowed_special: AbstractSet[str] | None = None) -> np.ndarray:
        """Piece start offsets (uint32, ascending, plus a final sentinel = total bytes) of a packed batch --
        what `regex.find_iter` yields at src/lib.rs:365/405, computed by the GPU pre-tokeniser.  Chars the pattern does not match
        (generic engine only) are steps of their own; their offsets are also left in `self.last_gaps`."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        doc_off = np.ascontiguousarray(doc_off, dtype=np.uint64)
        _check_packed(blob, doc_off)
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        src = blob if len(blob) else np.zeros(1, dtype=np.uint8)
        if allowed_special is None:
            ids, k, mode = np.zeros(1, dtype=np.uint32), 0, 0
        else:
            ids, k = self._allowed_ids(allowed_special)
            mode = 1
        rc = self._L.tk_pretokenize_batch(self._h, src.ctypes.data, doc_off.ctypes.data, len(doc_off) - 1, mode,
                                          ids.ctypes.data, k, ctypes.byref(out), ctypes.byref(n))
        _lib.raise_for(rc)
        starts = _take_u32(out, n.value)
        # bit 31: a char at which a pat_str of the generic engine matches nothing (find_iter skips it: no token) -- kept in self.last_gaps
        self.last_gaps = (starts[starts >= 0x80000000] & 0x7FFFFFFF).astype(np.uint32)
        return starts & np.uint32(0x7FFFFFFF) if len(self.last_gaps) else starts

    def encode_single_token(self, piece: bytes) -> int:
        tok = ctypes.c_uint32()
        rc = self._L.tk_encode_single_token(self._h, self._as_u8(piece).ctypes.data, len(piece), ctypes.byref(tok))
        _lib.raise_for(rc, key=piece)
        return tok.value

    def encode_single_piece(self, piece: bytes) -> list[int]:
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        rc = self._L.tk_encode_single_piece(self._h, self._as_u8(piece).ctypes.data, len(piece), ctypes.byref(out), ctypes.byref(n))
        _lib.raise_for(rc)
        return _take_u32(out, n.value).tolist()

    def _byte_pair_encode(self, piece: bytes) -> list[int]:
        """byte_pair_encode (src/lib.rs:198-211): the merge loop without the whole-piece shortcut."""
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        rc = self._L.tk_byte_pair_encode(self._h, self._as_u8(piece).ctypes.data, len(piece), ctypes.byref(out), ctypes.byref(n))
        _lib.raise_for(rc)
        return _take_u32(out, n.value).tolist()

    # -- byte-level / unstable entry points: host logic over the primitives above
    def _last_piece_token_len(self, data: bytes, allowed_special: AbstractSet[str] | None = None) -> int:
        """Tokens contributed by the last regex piece (second value of CoreBPE::encode, src/lib.rs:439-441);
        0 when the text ends with an allowed special token (lib.rs:433)."""
        if not data:
            return 0
        starts = self.pretokenize_packed(np.frombuffer(data, dtype=np.uint8), np.array([0, len(data)], dtype=np.uint64),
                                         allowed_special)
        tail = data[int(starts[-2]):]
        if allowed_special and any(tail == s.encode("utf-8") for s in allowed_special if s in self._specials):
            return 0
        return len(self.encode_single_piece(tail))

    def _token_is_all_space(self, token: int) -> bool:
        try:
            b = self.decode_single_token_bytes(token)
        except KeyError:
            return False
        return all(c in b" \n\t" for c in b)

    def _increase_last_piece_token_len(self, tokens: list[int], last: int) -> int:
        # src/lib.rs:444-481
        if last > 0 and self._token_is_all_space(tokens[len(tokens) - last]):
            while last < len(tokens) and self._token_is_all_space(tokens[len(tokens) - last - 1]):
                last += 1
        return last

    def _encode_bytes(self, data: bytes) -> list[int]:
        """src/py.rs:72-115: bytes that may end in (or contain) invalid UTF-8."""
        try:
            data.decode("utf-8")
            return self._encode_np(data, None).tolist()
        except UnicodeDecodeError as e:
            valid = e.start
        head = data[:valid]
        tokens = self._encode_np(head, None).tolist()
        last = self._increase_last_piece_token_len(tokens, self._last_piece_token_len(head)) if tokens else 0
        if tokens and last > 0:
            unstable = self.decode_bytes(tokens[len(tokens) - last:]) + data[valid:]
            del tokens[len(tokens) - last:]
        else:
            unstable = data[valid:]
        if unstable:
            tokens.extend(self.encode_single_piece(unstable))
        return tokens

    def encode_with_unstable(self, text: str, allowed_special: AbstractSet[str]):
        """src/lib.rs:483-599 (`_encode_unstable_native`), restated on the host over the GPU primitives."""
        data = text.encode("utf-8")
        tokens = self._encode_np(data, allowed_special).tolist()
        last = self._last_piece_token_len(data, allowed_special)
        if last == 0:
            return tokens, []
        last = self._increase_last_piece_token_len(tokens, last)
        unstable = self.decode_bytes(tokens[len(tokens) - last:])
        del tokens[len(tokens) - last:]
        completions: set[tuple[int, ...]] = set()
        if not unstable:
            return tokens, []
        import bisect

        sorted_tokens = self._sorted_tokens()
        point = bisect.bisect_left(sorted_tokens, unstable)
        while point < len(sorted_tokens) and sorted_tokens[point].startswith(unstable):
            completions.add((self.encode_single_token(sorted_tokens[point]),))
            point += 1
        for i in range(1, len(unstable)):
            prefix, suffix = unstable[:i], unstable[i:]
            point = bisect.bisect_left(sorted_tokens, suffix)
            while point < len(sorted_tokens) and sorted_tokens[point].startswith(suffix):
                possibility = prefix + sorted_tokens[point]
                try:
                    possibility.decode("utf-8")
                    encoded = self._encode_np(possibility, None).tolist()
                except UnicodeDecodeError:
                    encoded = self._byte_pair_encode(possibility)  # lib.rs:555
                seq, seq_len = [], 0
                for t in encoded:
                    seq.append(t)
                    seq_len += len(self.decode_single_token_bytes(t))
                    if seq_len >= len(unstable):
                        break
                completions.add(tuple(seq))
                point += 1
        if len(unstable) > 1:
            # last code point of the unstable bytes: bstr::decode_last_utf8 (lib.rs:581-596); char::is_whitespace is the
            # Unicode White_Space property (not str.isspace(), which also accepts U+001C..U+001F)
            ch, k = None, 1
            for start in range(max(0, len(unstable) - 4), len(unstable)):
                try:
                    cand = unstable[start:].decode("utf-8")
                except UnicodeDecodeError:
                    continue
                if len(cand) == 1:
                    ch, k = cand, len(unstable) - start
                    break
            if ch is not None and len(unstable) - k > 0 and ch in _WHITE_SPACE:
                re = self._byte_pair_encode(unstable[:len(unstable) - k]) + self._byte_pair_encode(unstable[len(unstable) - k:])
                completions.add(tuple(re))
        return tokens, [list(c) for c in completions]

    # ------------------------------------------------------------------ decoding (src/py.rs:156-183)
    def decode_bytes(self, tokens: Sequence[int]) -> bytes:
        arr = np.asarray(tokens, dtype=np.uint32) if len(tokens) else np.zeros(1, dtype=np.uint32)
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        rc = self._L.tk_decode_bytes(self._h, arr.ctypes.data, len(tokens), ctypes.byref(out), ctypes.byref(n))
        _lib.raise_for(rc)
        data = ctypes.string_at(out, n.value)
        self._L.tk_free(out)
        return data

    def decode_batch_packed(self, tokens: np.ndarray, tok_off: np.ndarray, as_array: bool = False):
        """One GPU call for a packed batch: (all bytes back to back, byte_off uint64[n_docs + 1])  -- tk_decode_batch.
        as_array: the bytes as a read-only uint8 array over the library's (page-locked) result buffer instead of a `bytes` copy."""
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        tok_off = np.ascontiguousarray(tok_off, dtype=np.uint64)
        if tok_off.ndim != 1 or len(tok_off) < 1 or int(tok_off[-1]) != len(tokens):
            raise ValueError("tok_off must hold n_docs + 1 offsets ending at len(tokens)")
        byte_off = np.empty(len(tok_off), dtype=np.uint64)
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        src = tokens if len(tokens) else np.zeros(1, dtype=np.uint32)
        rc = self._L.tk_decode_batch(self._h, src.ctypes.data, tok_off.ctypes.data, len(tok_off) - 1, ctypes.byref(out), ctypes.byref(n),
                                     byte_off.ctypes.data)
        _lib.raise_for(rc)
        if as_array and n.value >= (1 << 16):
            return np.asarray(_OwnedBuffer(out.value, n.value, "|u1")), byte_off
        data = ctypes.string_at(out, n.value)
        self._L.tk_free(out)
        return (np.frombuffer(data, dtype=np.uint8) if as_array else data), byte_off

    def decode_batch_device(self, d_tokens: int, n_tokens: int, d_tok_off: int, n_docs: int, stream: int = 0):
        """Device-resident decode (tk_decode_batch_device): pointers to uint32 ids and uint64 token offsets (0: none) on this core's device ->
        (device pointer to the bytes, number of bytes, device pointer to the n_docs + 1 byte offsets or 0).  The buffers are the library's and
        stay valid until the next decode call."""
        db, nb, do = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_void_p()
        rc = self._L.tk_decode_batch_device(self._h, d_tokens or None, n_tokens, d_tok_off or None, n_docs, stream or None, ctypes.byref(db),
                                            ctypes.byref(nb), ctypes.byref(do))
        _lib.raise_for(rc)
        return db.value or 0, nb.value, do.value or 0

    def decode_single_token_bytes(self, token: int) -> bytes:
        if not 0 <= token <= 0xFFFFFFFF:
            raise KeyError(str(token))
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        rc = self._L.tk_decode_single_token_bytes(self._h, token, ctypes.byref(p), ctypes.byref(n))
        _lib.raise_for(rc, key=str(token))
        return ctypes.string_at(p, n.value)

    def token_byte_values(self) -> list[bytes]:
        return list(self._sorted_tokens())

    def _sorted_tokens(self) -> list[bytes]:
        """All token byte strings in lexicographic order (lib.rs:648-650), fetched once per CoreBPE in one call."""
        cached = getattr(self, "_sorted_cache", None)
        if cached is None:
            pb, po, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
            _lib.raise_for(self._L.tk_sorted_tokens_packed(self._h, ctypes.byref(pb), ctypes.byref(po), ctypes.byref(n)))
            off = np.ctypeslib.as_array(ctypes.cast(po, ctypes.POINTER(ctypes.c_uint64)), shape=(n.value + 1,))
            blob = ctypes.string_at(pb, int(off[-1]))
            bounds = off.tolist()
            cached = self._sorted_cache = [blob[a:b] for a, b in zip(bounds[:-1], bounds[1:])]
        return cached

    def set_output_buffers(self, n: int):
        """`encode_batch_device` alternates between n (1 or 2) pairs of result buffers: with 2 the result of a call stays valid while the
        next call runs (tk_set_output_buffers; the several-process gather of tiktoken_amd.distributed sends from it without a copy)."""
        _lib.raise_for(self._L.tk_set_output_buffers(self._h, int(n)))

    # ------------------------------------------------------------------ instrumentation
    def set_profiling(self, on: bool):
        self._L.tk_set_profiling(self._h, 1 if on else 0)

    def reset_kernel_ms(self):
        self._L.tk_reset_kernel_ms(self._h)

    def kernel_ms(self, name: str) -> tuple[float, int]:
        ms, n = ctypes.c_double(), ctypes.c_uint64()
        self._L.tk_get_kernel_ms(self._h, name.encode(), ctypes.byref(ms), ctypes.byref(n))
        return ms.value, n.value

    def last_stats(self) -> dict:
        v = [ctypes.c_uint64() for _ in range(6)]
        self._L.tk_last_stats(self._h, *[ctypes.byref(x) for x in v])
        return dict(zip(("bytes", "pieces", "tokens", "docs", "medium_pieces", "long_pieces"), (x.value for x in v)))

    def stat(self, name: str) -> int:
        """One named figure of the last encode call (tk_stat): "chunks", "small_calls", "mid_calls", "workspace_bytes", ..."""
        return int(self._L.tk_stat(self._h, name.encode())) if hasattr(self._L, "tk_stat") else 0
