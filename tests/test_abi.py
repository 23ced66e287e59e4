"""The C-ABI library loads on a GPU-less host, exports every symbol include/tiktoken_amd.h declares,
and refuses to work without a device (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import pytest

import helpers as h

ROOT = h.ROOT


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "tiktoken_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from tiktoken_amd import _lib

    _lib.build()
    L = ctypes.CDLL(os.path.join(ROOT, "tiktoken_amd", "csrc", "libtiktoken_amd.so"))
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n


def test_code_object_targets_gfx950():
    so = os.path.join(ROOT, "tiktoken_amd", "csrc", "libtiktoken_amd.so")
    out = subprocess.run(["strings", "-a", so], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_pattern_ids():
    from tiktoken_amd import _lib
    from tiktoken_ext import openai_public as pub
    from oracle import py_oracle as po

    L = _lib.lib()
    assert L.tk_pattern_id(pub.r50k_pat_str.encode()) == 0
    assert L.tk_pattern_id(po.GPT2_ORIG_PAT.encode()) == 0
    assert L.tk_pattern_id(pub.cl100k_pat_str.encode()) == 1
    assert L.tk_pattern_id(pub.o200k_pat_str.encode()) == 2
    assert L.tk_pattern_id(b"\\w+") == 3  # the generic engine (tk_regex.cpp)
    assert L.tk_pattern_id(b"(?<=a+)c") == -1  # (look-behind of variable length)
    # the plugin module spells the same patterns as the reference (compared via the oracle's copies)
    assert (pub.r50k_pat_str, pub.cl100k_pat_str, pub.o200k_pat_str) == (po.R50K_PAT, po.CL100K_PAT, po.O200K_PAT)


def test_fails_loudly_without_a_device():
    from tiktoken_amd import CoreBPE, _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        CoreBPE({bytes([b]): b for b in range(256)}, {}, h.PAT_STR[0])
    with pytest.raises(ValueError):  # argument errors are still reported as ValueError (src/py.rs:21-22)
        CoreBPE({bytes([b]): b for b in range(256)}, {}, r"(?<=a+)\w+")


def test_product_never_imports_the_oracle():
    """Only tests/, tools/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tiktoken_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|tk_oracle|libtk_oracle|hostsim", txt, flags=re.M):
                    bad.append(f)
    for f in os.listdir(os.path.join(ROOT, "tiktoken_ext")):
        if f.endswith(".py") and "oracle" in open(os.path.join(ROOT, "tiktoken_ext", f)).read():
            bad.append(f)
    assert not bad, bad
