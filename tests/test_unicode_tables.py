"""The Unicode class table (tk_unicode_tables.inc, shared by the HIP kernels and the C oracle) checked code point
by code point -- all 0x110000 of them -- against two independent sources, without going through any scanner:

  * Python `regex` property predicates (\\p{L} \\p{N} \\p{M} \\s, Lu/Lt/Ll/Lm/Lo), evaluated with findall over one
    string that holds every code point (a different formulation from tools/gen_unicode_tables.py, which matches
    one character at a time);
  * the standard library's `unicodedata.category` for every code point that is assigned in ITS Unicode version
    (13.0 here, older than `regex`'s 17.0).  Categories of assigned code points almost never change; the one
    change between the two versions is U+0295, re-classified from Ll to Lo in Unicode 16.0, listed below.

These are the classes fancy-regex resolves for the reference at src/lib.rs:365 (tiktoken_ext/openai_public.py
patterns); a wrong class for a rare script would change piece boundaries without any corpus test noticing.
"""
import os
import re
import unicodedata

import numpy as np
import regex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = [os.path.join(ROOT, "tiktoken_amd", "csrc", "tk_unicode_tables.inc"), os.path.join(ROOT, "oracle", "tk_unicode_tables.inc")]
NL, SP, WSO, LU, LL, LC, MK, NU, AP, SL, OT = range(1, 12)


def _load(path):
    src = open(path).read()
    arrs = {}
    for name, body in re.findall(r"static const unsigned char (\w+)\[[^\]]*\] = \{([^}]*)\}", src):
        arrs[name] = np.array([int(x) for x in body.replace("\n", " ").split(",") if x.strip()], dtype=np.uint8)
    s1, s2 = arrs["tk_uc_stage1"], arrs["tk_uc_stage2"]
    assert len(s1) == 0x1100 and len(s2) % 256 == 0 and int(s1.max()) < len(s2) // 256
    cps = np.arange(0x110000)
    return s2[s1[cps >> 8].astype(np.int64) * 256 + (cps & 255)]


def test_both_copies_are_the_same_table():
    assert open(INC[0]).read() == open(INC[1]).read()


def _members(pattern: str, everything: str) -> np.ndarray:
    m = np.zeros(0x110000, bool)
    for ch in regex.findall(pattern, everything):
        m[ord(ch)] = True
    return m


def test_every_code_point_against_regex_properties():
    table = _load(INC[0])
    everything = "".join(chr(c) for c in range(0x110000))  # (lone surrogates included: Python strings hold them)
    L, N, M, WS = (_members(p, everything) for p in (r"\p{L}", r"\p{N}", r"\p{M}", r"\s"))
    lu_lt, ll, lm_lo = (_members(p, everything) for p in (r"[\p{Lu}\p{Lt}]", r"\p{Ll}", r"[\p{Lm}\p{Lo}]"))
    assert int(WS.sum()) == 25  # White_Space is frozen at 25 code points
    assert not (L & N).any() and not (L & M).any() and not (N & M).any() and not (WS & (L | N | M)).any()
    assert ((lu_lt.astype(int) + ll + lm_lo) == L).all()  # the three letter sub-classes partition \p{L}
    want = np.full(0x110000, OT, np.uint8)
    want[M] = MK
    want[N] = NU
    want[lu_lt] = LU
    want[ll] = LL
    want[lm_lo] = LC
    want[WS] = WSO
    want[ord(" ")] = SP
    want[ord("\r")] = want[ord("\n")] = NL
    want[ord("'")] = AP
    want[ord("/")] = SL
    want[0xD800:0xE000] = OT  # surrogates never occur in UTF-8 text
    bad = np.flatnonzero(table != want)
    assert len(bad) == 0, [(hex(int(c)), int(table[c]), int(want[c])) for c in bad[:20]]


def test_assigned_code_points_against_unicodedata():
    table = _load(INC[0])
    by_cat = {"Lu": LU, "Lt": LU, "Ll": LL, "Lm": LC, "Lo": LC, "Mn": MK, "Mc": MK, "Me": MK, "Nd": NU, "Nl": NU, "No": NU}
    ws = {0x9, 0xA, 0xB, 0xC, 0xD, 0x20, 0x85, 0xA0, 0x1680, *range(0x2000, 0x200B), 0x2028, 0x2029, 0x202F, 0x205F, 0x3000}
    recategorised = {0x295: LC}  # LATIN LETTER PHARYNGEAL VOICED FRICATIVE: Ll until Unicode 15.1, Lo from 16.0
    checked = 0
    for cp in range(0x110000):
        cat = unicodedata.category(chr(cp))
        if cat in ("Cn", "Cs"):
            continue  # unassigned in the stdlib's (older) Unicode version: `regex` may know it already
        if cp in ws:
            want = NL if cp in (0xA, 0xD) else (SP if cp == 0x20 else WSO)
        elif cp == 0x27:
            want = AP
        elif cp == 0x2F:
            want = SL
        else:
            want = by_cat.get(cat, OT)
        want = recategorised.get(cp, want) if unicodedata.unidata_version < "16" else want
        assert table[cp] == want, (hex(cp), cat, int(table[cp]), want)
        checked += 1
    assert checked > 140000
