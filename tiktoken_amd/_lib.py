"""ctypes binding of csrc/libtiktoken_amd.so (the C ABI declared in include/tiktoken_amd.h).

This is the only door from Python into the HIP encode path.  There is deliberately no
fallback: if the shared library is missing or no MI355X is visible, construction of a CoreBPE
fails loudly (RuntimeError) instead of silently encoding on the CPU.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# ($TIKTOKEN_AMD_LIB: another build of the same library -- kernel experiments with other compile-time parameters, tools/build_variant.sh)
_SO = os.environ.get("TIKTOKEN_AMD_LIB") or os.path.join(_HERE, "csrc", "libtiktoken_amd.so")
_lock = threading.Lock()
_lib = None

TK_OK, TK_VALUE_ERROR, TK_KEY_ERROR, TK_RUNTIME_ERROR, TK_UNSUPPORTED = range(5)


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "libtiktoken_amd.so"])
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_SO):
            raise ImportError(
                f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  tiktoken_amd has no CPU implementation to fall back to."
            )
        L = ctypes.CDLL(_SO)
        vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
        P = ctypes.POINTER
        L.tk_last_error.restype = ctypes.c_char_p
        L.tk_device_count.restype = i32
        L.tk_pattern_id.restype = i32
        L.tk_pattern_id.argtypes = [ctypes.c_char_p]
        L.tk_create.restype = i32
        L.tk_create.argtypes = [vp, vp, vp, u64, vp, vp, vp, u64, ctypes.c_char_p, i32, P(vp)]
        L.tk_destroy.argtypes = [vp]
        L.tk_encode_batch.restype = i32
        L.tk_encode_batch.argtypes = [vp, vp, vp, u64, i32, vp, u64, P(vp), P(u64), vp]
        L.tk_encode_batch_device.restype = i32
        L.tk_encode_batch_device.argtypes = [vp, vp, u64, vp, vp, u64, i32, vp, u64, vp, P(vp), P(u64), P(vp)]
        L.tk_pretokenize_batch.restype = i32
        L.tk_pretokenize_batch.argtypes = [vp, vp, vp, u64, i32, vp, u64, P(vp), P(u64)]
        L.tk_encode_ordinary.restype = i32
        L.tk_encode_ordinary.argtypes = [vp, vp, u64, P(vp), P(u64)]
        L.tk_encode.restype = i32
        L.tk_encode.argtypes = [vp, vp, u64, vp, u64, P(vp), P(u64)]
        L.tk_encode_single_piece.restype = i32
        L.tk_encode_single_piece.argtypes = [vp, vp, u64, P(vp), P(u64)]
        L.tk_byte_pair_encode.restype = i32
        L.tk_byte_pair_encode.argtypes = [vp, vp, u64, P(vp), P(u64)]
        L.tk_encode_single_token.restype = i32
        L.tk_encode_single_token.argtypes = [vp, vp, u64, P(u32)]
        L.tk_decode_bytes.restype = i32
        L.tk_decode_bytes.argtypes = [vp, vp, u64, P(vp), P(u64)]
        L.tk_decode_batch.restype = i32
        L.tk_decode_batch.argtypes = [vp, vp, vp, u64, P(vp), P(u64), vp]
        L.tk_decode_batch_device.restype = i32
        L.tk_decode_batch_device.argtypes = [vp, vp, u64, vp, u64, vp, P(vp), P(u64), P(vp)]
        L.tk_decode_single_token_bytes.restype = i32
        L.tk_decode_single_token_bytes.argtypes = [vp, u32, P(vp), P(u64)]
        L.tk_n_tokens.restype = u64
        L.tk_n_tokens.argtypes = [vp]
        L.tk_sorted_token.restype = i32
        L.tk_sorted_token.argtypes = [vp, u64, P(vp), P(u64), P(u32)]
        L.tk_sorted_tokens_packed.restype = i32
        L.tk_sorted_tokens_packed.argtypes = [vp, P(vp), P(vp), P(u64)]
        L.tk_group_create.restype = i32
        L.tk_group_create.argtypes = [vp, u32, P(vp)]
        L.tk_group_destroy.argtypes = [vp]
        L.tk_group_size.restype = u32
        L.tk_group_size.argtypes = [vp]
        L.tk_group_encode_batch.restype = i32
        L.tk_group_encode_batch.argtypes = [vp, vp, vp, u64, i32, vp, u64, P(vp), P(u64), vp]
        L.tk_group_encode_batch_device.restype = i32
        L.tk_group_encode_batch_device.argtypes = [vp, vp, vp, u64, i32, vp, u64, P(vp), P(u64), P(vp)]
        L.tk_group_stat.restype = u64
        L.tk_group_stat.argtypes = [vp, ctypes.c_char_p]
        L.tk_parse_tiktoken_bpe.restype = i32
        L.tk_parse_tiktoken_bpe.argtypes = [vp, u64, P(vp), P(vp), P(vp), P(u64)]
        L.tk_validate_utf8.restype = i32
        L.tk_validate_utf8.argtypes = [vp, u64, P(u64)]
        L.tk_free.argtypes = [vp]
        L.tk_set_output_buffers.restype = i32
        L.tk_set_output_buffers.argtypes = [vp, ctypes.c_uint32]
        L.tk_set_profiling.argtypes = [vp, i32]
        L.tk_reset_kernel_ms.argtypes = [vp]
        L.tk_get_kernel_ms.restype = i32
        L.tk_get_kernel_ms.argtypes = [vp, ctypes.c_char_p, P(ctypes.c_double), P(u64)]
        L.tk_last_stats.argtypes = [vp, P(u64), P(u64), P(u64), P(u64), P(u64), P(u64)]
        if hasattr(L, "tk_stat"):  # (absent from older builds selected through $TIKTOKEN_AMD_LIB)
            L.tk_stat.restype = u64
            L.tk_stat.argtypes = [vp, ctypes.c_char_p]
        _lib = L
    return _lib


def last_error() -> str:
    return lib().tk_last_error().decode("utf-8", "replace")


def device_count() -> int:
    try:
        return int(lib().tk_device_count())
    except ImportError:
        return 0


def raise_for(rc: int, key=None):
    """Map a C-ABI status to the exception type the reference raises (src/py.rs:21-22,46,142,160,171)."""
    if rc == TK_OK:
        return
    msg = last_error()
    if rc in (TK_VALUE_ERROR, TK_UNSUPPORTED):
        raise ValueError(msg)
    if rc == TK_KEY_ERROR:
        raise KeyError(key if key is not None else msg)
    raise RuntimeError(msg)
