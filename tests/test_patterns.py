"""pat_str handling (reference: the pattern is compiled once per Encoding, src/lib.rs:623): the parser of the supported family
(tiktoken_amd/csrc/tk_pattern.cpp) and the scanners parametrised by it, against Python `regex` -- the engine the reference's own
educational implementation uses for the same patterns (tiktoken/_educational.py).

CPU part: the device headers compiled for the host (tests/hostsim) run the byte-walking scanner, the run-query scanner, the
bit-parallel scanners and the tile rule on generated texts for every variation; `regex.findall` is the expected split.
GPU part (-m gpu): the same through tk_pretokenize_batch and a whole encode through the Python oracle's regex path."""
import random

import pytest
import regex

import helpers as h

R50K, CL100K, O200K = (h.PAT_STR[i] for i in range(3))
CL100K_PLAIN = r"""'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s+$|\s*[\r\n]|\s+(?!\S)|\s"""  # no possessive markers
QWEN2 = r"""(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+"""

# variations inside the family: (name, pattern)
VARIANTS = [
    ("qwen2 (cl100k with single digits)", QWEN2),
    ("cl100k, groups of two digits", CL100K.replace("{1,3}", "{1,2}")),
    ("cl100k, unbounded digits", CL100K.replace(r"\p{N}{1,3}+", r"\p{N}+")),
    ("cl100k without possessive markers, seven digits", CL100K_PLAIN.replace("{1,3}", "{1,7}")),
    ("cl100k, other contractions", CL100K.replace("[sdmt]|ll|ve|re", "[sdm]|ll|re|nt")),
    ("cl100k, case-sensitive contractions", CL100K.replace("(?i:", "(?:")),
    ("cl100k, no contractions suffix set with slash", CL100K.replace(r"[\r\n]*+", r"[\r\n/]*")),
    ("cl100k, no suffix set", CL100K.replace(r"[\r\n]*+", "")),
    ("cl100k without the end-of-text rule", CL100K.replace(r"\s++$|", "")),
    ("cl100k without the newline rule", CL100K.replace(r"\s*[\r\n]|", "")),
    ("o200k, single digits", O200K.replace("{1,3}", "{1,1}")),
    ("o200k, groups of four digits", O200K.replace("{1,3}", "{1,4}")),
    ("o200k, unbounded digits", O200K.replace(r"\p{N}{1,3}", r"\p{N}+")),
    ("o200k, fewer contractions", O200K.replace("'s|'t|'re|'ve|'m|'ll|'d", "'s|'t|'ll")),
    ("o200k, no contractions", O200K.replace("(?i:'s|'t|'re|'ve|'m|'ll|'d)?", "")),
    ("o200k, case-sensitive contractions", O200K.replace("(?i:'s", "(?:'s")),
    ("o200k, newline-only suffix set", O200K.replace(r"[\r\n/]*", r"[\r\n]*")),
    ("o200k, slash-only suffix set", O200K.replace(r"[\r\n/]*", r"/*")),
    ("o200k with the end-of-text rule", O200K.replace(r"|\s*[\r\n]+", r"|\s+$|\s*[\r\n]+")),
    ("r50k, other contractions", R50K.replace("[sdmt]|ll|ve|re", "[st]|ve|nt|em")),
    ("r50k, case-insensitive contractions", R50K.replace("'(?:", "'(?i:")),
    ("r50k, no contractions", R50K.replace("'(?:[sdmt]|ll|ve|re)|", "")),
]

ALPHABET = list("abcdeflmnrstvxABDELMNRSTVX") + list("0123456789") * 2 + [" "] * 8 + ["\n", "\r", "\t", " ", "　", "'", "'", "'", "/", "/", "!", ".",
            "-", "€", "é", "É", "中", "́", "ǅ", "ʰ", "²", "٣", "ſ", "K", "\U0001F600"]


def _texts(seed, n, lo=0, hi=40):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        k = rng.randint(lo, hi)
        s = []
        while len(s) < k:
            r = rng.random()
            if r < 0.15:
                s.extend(rng.choice(["'s", "'S", "'ll", "'LL", "'nt", "'Ve", "'re", "'em", "'d", "'m", "'t", "'ſ"]))
            elif r < 0.25:
                s.extend(rng.choice(ALPHABET) * rng.randint(2, 6))
            else:
                s.append(rng.choice(ALPHABET))
        out.append("".join(s))
    return out


def _ends_regex(pat, docs, off):
    ref = []
    for d, t in enumerate(docs):
        pos = 0
        for piece in regex.findall(pat, t):
            pos += len(piece.encode())
            ref.append(int(off[d]) + pos)
        assert pos == len(t.encode()), (pat, t)
    return ref


TINY = {bytes([b]): b for b in range(256)}


def test_stock_spellings_map_to_the_stock_scanners():
    from tiktoken_amd import _lib
    from oracle import py_oracle as po

    L = _lib.lib()
    for fam, p in ((0, R50K), (0, po.GPT2_ORIG_PAT), (1, CL100K), (2, O200K)):
        assert L.tk_pattern_id(p.encode()) == fam
    # spelled without possessive markers, with the other forms of the contraction list and of the white-space tail
    assert L.tk_pattern_id(CL100K_PLAIN.encode()) == 1
    assert L.tk_pattern_id(CL100K.replace("'(?i:[sdmt]|ll|ve|re)", "(?i:'s|'t|'re|'ve|'m|'ll|'d)").encode()) == 1
    assert L.tk_pattern_id(O200K.replace(r"\s*[\r\n]+", r"\s*[\r\n]").encode()) == 2
    for _, p in VARIANTS:
        assert L.tk_pattern_id(p.encode()) in (0, 1, 2), p


@pytest.mark.parametrize("bad,why", [
    (r"\w+|\s+", "not the letter alternative"),
    (CL100K.replace("[sdmt]|ll|ve|re", "[sdmt]|ll|ve|re|ing"), "one or two letters"),
    (CL100K.replace("[sdmt]|ll|ve|re", "[sdmt]|ll|ve|re|st"), "beginning of a two-letter one"),
    (CL100K.replace("[sdmt]", "[sdmtk]"), "U+212A"),
    (CL100K.replace("{1,3}", "{2,3}"), "digit group"),
    (CL100K.replace(r"[\r\n]*+", r"[\r\n\\]*"), "suffix set"),
    (CL100K + r"|x", "unexpected alternative"),
    (O200K.replace(r"\p{N}{1,3}|", ""), "digit group"),
    (R50K.replace(r"| ?\p{N}++", ""), r"\p{N}+"),
    (CL100K.replace(r"\p{L}++", r"[\p{L}\p{M}]+"), "letter alternative"),
])
def test_patterns_outside_the_families_run_on_the_generic_engine(bad, why):
    """What the hand-written scanners do not cover (`why`: the family parser's reason) is compiled for the generic engine (tk_regex.cpp)
    instead of being refused; its split is Python `regex`'s."""
    from tiktoken_amd import _lib

    assert _lib.lib().tk_pattern_id(bad.encode()) == 3
    rx = h.RxSim(bad)
    texts = [t for t in _texts(len(why), 400) + ["x'ing y'st z'k 'K", "12 345 6789", "a\\\n\\b", "don't stop"] if _covers(bad, t)]
    assert len(texts) > 40
    blob, off = h.pack([d.encode() for d in texts])
    ref = _ends_regex(bad, texts, off)
    starts = rx.split([d.encode() for d in texts])
    assert starts[1:] + [len(blob)] == ref


def _covers(pat, text):
    at = 0
    for m in regex.finditer(pat, text):
        if m.start() != at:
            return False
        at = m.end()
    return at == len(text)


@pytest.mark.parametrize("name,pat", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_scanners_of_a_variation_equal_regex(name, pat):
    sim = h.HostSim(pat, TINY, {})
    before = h.sim_lib().tks_runs_mismatches()
    texts = _texts(hash(name) & 0xFFFF, 1500) + ["", "a", "'", " ", "\n", "1", "12345678901234567890", "x'll'LL'll", "  \n  \n  ", "a  \n\n  b", "   ", "a/\n//\nb",
                                               "!!\r\n/x", "Don't DON'T don'T", "I'ſ 'ſx", "'K", "1 22 333 4444 55555", "٣٣٣٣٣٣٣", "²²²²²"]
    for i in range(0, len(texts), 7):
        docs = texts[i:i + 7]
        blob, off = h.pack([d.encode() for d in docs])
        ref = _ends_regex(pat, docs, off)
        assert sim.piece_ends(blob, off)[0].tolist() == ref, (name, docs)             # byte-walking scanner from certain starts
        assert sim.piece_ends(blob, off, bits=True)[0].tolist() == ref, (name, docs)  # bit-parallel scanners
    assert h.sim_lib().tks_runs_mismatches() == before                                 # run-query scanner agreed at every piece
    # the tile rule of the front kernel (tiny tiles, so that pieces and contexts cross tile edges all the time) and long runs
    long_texts = _texts(hash(name) & 0xFFF, 60, 200, 1200) + ["7" * 700 + "x" * 300 + " " * 400 + "\n" * 90 + "/" * 50 + "!" * 333 + "中" * 200 + "'ll" * 100]
    for t in long_texts:
        blob, off = h.pack([t.encode()])
        ref = _ends_regex(pat, [t], off)
        for tile, left in ((16, 16), (64, 32), (256, 64)):
            assert sim.piece_ends_tiled(blob, off, tile, left)[0].tolist() == ref, (name, tile, t[:80])
        assert sim.piece_ends(blob, off, bits=True)[0].tolist() == ref


def _pattern_from(family, contr, ci, digits, suffix, dollar, nl_rule, spell):
    """A pattern string of the supported family from its parameters (spell: which of the equivalent spellings to use)."""
    items = sorted(contr)
    flag = "i" if ci else ""
    if spell % 3 == 0:
        singles = "".join(c for c in items if len(c) == 1)
        alts = ([f"[{singles}]"] if singles else []) + [c for c in items if len(c) == 2]
        clist = f"'(?{flag}:{'|'.join(alts)})" if alts else ""
    elif spell % 3 == 1:
        clist = f"(?{flag}:{'|'.join(chr(39) + c for c in items)})" if items else ""
    else:
        clist = "|".join("'" + c for c in items) if not ci else (f"(?i:{'|'.join(chr(39) + c for c in items)})" if items else "")
    poss = "+" if spell & 4 else ""
    dig = r"\p{N}+" if digits == 0 else (r"\p{N}" if digits == 1 and spell & 8 else r"\p{N}{1,%d}" % digits)
    suf = {0: "", 1: r"[\r\n]*", 2: "/*", 3: r"[\r\n/]*"}[suffix]
    tail = ([r"\s+" + poss + "$"] if dollar else []) + ([r"\s*[\r\n]" + ("+" if spell & 16 else "")] if nl_rule else []) + [r"\s+(?!\S)", r"\s+" if spell & 32 else r"\s"]
    if family == 0:
        alts = ([clist] if clist else []) + [r" ?\p{L}+" + poss, r" ?\p{N}+" + poss, r" ?[^\s\p{L}\p{N}]+" + poss] + [t for t in tail if "[\\r\\n]" not in t]
    elif family == 1:
        alts = ([clist] if clist else []) + [r"[^\r\n\p{L}\p{N}]?" + poss + r"\p{L}+" + poss, dig, r" ?[^\s\p{L}\p{N}]+" + poss + suf] + tail
    else:
        csuf = (f"(?{flag}:{'|'.join(chr(39) + c for c in items)})?" if items else "")
        w1 = r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]*[\p{Ll}\p{Lm}\p{Lo}\p{M}]+" + csuf
        w2 = r"[^\r\n\p{L}\p{N}]?[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]+[\p{Ll}\p{Lm}\p{Lo}\p{M}]*" + csuf
        alts = [w1, w2, dig, r" ?[^\s\p{L}\p{N}]+" + suf] + tail
    return "|".join(alts)


def test_generated_patterns_equal_regex():
    """Patterns generated from random parameters (contraction list, case sensitivity, digit group, suffix set, white-space rules, and the
    equivalent spellings of each part): whatever the family parser accepts must split exactly as Python `regex` does under the
    hand-written scanners; what it refuses (e.g. a one-letter contraction that begins a two-letter one) runs on the generic engine, with
    the same split."""
    from tiktoken_amd import _lib

    rng = random.Random(20260921)
    letters1, letters2 = list("sdmtnxe"), ["ll", "ve", "re", "nt", "em", "dx", "ar"]
    accepted = refused = 0
    texts = _texts(99, 500, 0, 60) + ["x'll'LL'nt'em y'S z'T 12345 6 78/\n/ a  \n\n  b   ", "I'ſ 'ſx 'K DON'T don'T He'Ll"]
    for it in range(60):
        family = rng.randrange(3)
        contr = set(rng.sample(letters1, rng.randrange(0, 5))) | set(rng.sample(letters2, rng.randrange(0, 5)))
        pat = _pattern_from(family, contr, rng.random() < 0.6, rng.choice([0, 1, 1, 2, 3, 3, 5, 12]), rng.choice([0, 1, 1, 2, 3, 3]),
                            rng.random() < 0.5, rng.random() < 0.75, rng.randrange(64))
        regex.compile(pat)  # (a well-formed pattern whatever the parser thinks of it)
        if _lib.lib().tk_pattern_id(pat.encode()) == 3:  # not a member of the scanner families: the generic engine (tk_regex.cpp) takes it
            refused += 1
            rx = h.RxSim(pat)
            blob, off = h.pack([d.encode() for d in texts])
            assert rx.split([d.encode() for d in texts])[1:] + [len(blob)] == _ends_regex(pat, texts, off), pat
            continue
        sim = h.HostSim(pat, TINY, {})
        accepted += 1
        for i in range(0, len(texts), 9):
            docs = texts[i:i + 9]
            blob, off = h.pack([d.encode() for d in docs])
            ref = _ends_regex(pat, docs, off)
            assert sim.piece_ends(blob, off)[0].tolist() == ref, (pat, docs)
            assert sim.piece_ends(blob, off, bits=True)[0].tolist() == ref, (pat, docs)
        t = "".join(texts[:120])
        blob, off = h.pack([t.encode()])
        assert sim.piece_ends_tiled(blob, off, 32, 16)[0].tolist() == _ends_regex(pat, [t], off), pat
    assert accepted >= 25, (accepted, refused)


# ---------------------------------------------------------------- on the device
@pytest.mark.gpu
@pytest.mark.parametrize("name,pat", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_gpu_split_of_a_variation_equals_regex(name, pat):
    from tiktoken_amd import CoreBPE

    core = CoreBPE(TINY, {}, pat)
    texts = _texts(hash(name) & 0xFFFF, 4000, 0, 120) + _texts(7, 40, 3000, 9000) + ["x" * 20000 + "7" * 9001 + " " * 7000 + "\n" * 300 + "ab'll" * 900]
    blob, off = h.pack([t.encode() for t in texts])
    ref = _ends_regex(pat, texts, off)
    starts = core.pretokenize_packed(blob, off)
    assert starts[1:].tolist() == ref
    # a whole encode of short calls (the one-launch path scans with the same pattern)
    for t in texts[:300]:
        if t:
            assert core.encode_ordinary(t) == list(t.encode())  # (byte vocabulary: one token per byte, in piece order)


@pytest.mark.gpu
def test_gpu_encode_with_a_variation_equals_the_python_restatement():
    """A real (shaped) vocabulary under Qwen2's pattern: token ids against the Python restatement of the reference's encode loop with
    `regex` doing the split (what tiktoken/_educational.py does)."""
    from oracle import py_oracle as po
    from tiktoken_amd import CoreBPE

    ranks = h.load_vocab("cl100k_shaped")
    core = CoreBPE(ranks, {}, QWEN2)
    texts = _texts(3, 400, 0, 200)
    got = core.encode_batch_packed(*h.pack([t.encode() for t in texts]))
    toks, toff = got
    for i, t in enumerate(texts):
        want = []
        for piece in regex.findall(QWEN2, t):
            want += po.byte_pair_encode(piece.encode(), ranks)
        assert toks[int(toff[i]):int(toff[i + 1])].tolist() == want, t


@pytest.mark.gpu
@pytest.mark.parametrize("pat", [QWEN2, O200K.replace("{1,3}", "{1,2}"), R50K.replace("[sdmt]|ll|ve|re", "[st]|ve|nt|em")], ids=["qwen2", "o200k-2-digits", "r50k-contractions"])
def test_gpu_variation_with_special_tokens(pat):
    """A pattern outside the stock three together with special tokens (the generic kernels' special-token instance): the text is cut at
    the allowed specials (each a piece of its own, lib.rs:375-442), `regex` splits the stretches between them, and the ids are the
    specials' ids or, with a byte vocabulary, the bytes."""
    from tiktoken_amd import CoreBPE

    specials = {"<|endoftext|>": 300, "<|im_start|>": 301, "<|im_end|>": 302}
    core = CoreBPE(TINY, specials, pat)
    rng = random.Random(8)
    texts = []
    for t in _texts(21, 600, 0, 80):
        parts = []
        for ch in t:
            parts.append(ch)
            if rng.random() < 0.04:
                parts.append(rng.choice(list(specials) + ["<|endoftext", "<|im_"]))
        texts.append("".join(parts))
    allowed = {"<|endoftext|>", "<|im_start|>"}  # (<|im_end|> stays ordinary text)
    sp = regex.compile("|".join(regex.escape(x) for x in sorted(allowed, key=len, reverse=True)))
    blob, off = h.pack([t.encode() for t in texts])
    want_ends, want_tokens = [], []
    for d, t in enumerate(texts):
        pos, byte_pos = 0, int(off[d])
        for m in list(sp.finditer(t)) + [None]:
            seg = t[pos:m.start()] if m else t[pos:]
            for piece in regex.findall(pat, seg):
                byte_pos += len(piece.encode())
                want_ends.append(byte_pos)
                want_tokens += list(piece.encode())
            if m:
                byte_pos += len(m.group().encode())
                want_ends.append(byte_pos)
                want_tokens.append(specials[m.group()])
                pos = m.end()
    starts = core.pretokenize_packed(blob, off, allowed)
    assert starts[1:].tolist() == want_ends
    toks, _ = core.encode_batch_packed(blob, off, allowed)
    assert toks.tolist() == want_tokens


def test_pattern_parsers_under_the_sanitizers(tmp_path):
    """tests/hostsim/pat_sanitize.cpp: tk_compile_pattern -- the family parser, then the generic compiler -- on random, mostly ill-formed
    strings and on mutations of the stock patterns, built with AddressSanitizer and UBSan: whatever a caller hands to tk_create is refused
    or compiled, never a crash (20 000 strings ran clean; 8 000 here); then the `.tiktoken` parser on damaged files and the table builder on
    vocabularies it has to refuse (missing byte, duplicate or oversized ranks) or accept (sparse ranks)."""
    import os
    import subprocess

    exe = str(tmp_path / "pat_sanitize")
    d, csrc = os.path.join(h.ROOT, "tests", "hostsim"), os.path.join(h.ROOT, "tiktoken_amd", "csrc")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-D_GLIBCXX_SANITIZE_VECTOR", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                         "-Wno-unused-function", os.path.join(d, "pat_sanitize.cpp"), os.path.join(csrc, "tk_tables.cpp"), os.path.join(csrc, "tk_pattern.cpp"),
                         os.path.join(csrc, "tk_regex.cpp"), "-pthread", "-o", exe], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe, R50K, CL100K, O200K], capture_output=True, text=True, timeout=900, env={**os.environ, "TK_SAN_ROUNDS": "8000"})
    assert run.returncode == 0 and run.stdout.startswith("ok 8000 "), (run.stdout[-300:], run.stderr[-3000:])


@pytest.mark.parametrize("name,pat", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_mid_size_cuts_are_piece_starts_under_a_variation(name, pat):
    """encode_mid (tk_api.hip) cuts a document of 2 .. 128 KiB at "ASCII letter, then space" whenever the PATTERN's table of certain starts says
    that a space behind a letter always starts a piece -- for a custom pattern of the families that table is derived by tk_pattern.cpp.  The
    premise, with Python `regex` on both sides: the pieces of the segments, one after the other, are the pieces of the document."""
    sim = h.HostSim(pat, TINY, {})
    rng = random.Random(len(name))
    words = ["the", "of", "Lorem", "ipsum", "x", "I", "don't", "DON'T", "we'LL", "a's", "12", "3.14", "2024", "é", "中文", "naïve", "(see", "note)", "...", "!?", "#tag",
             "a/b", "CamelCase", "ALLCAPS", "tail/", "end.\n", "x'", "'em", "'nt", "ǅ", "ʰ", "²", "ſ"]
    seps = [" "] * 10 + ["\n", "\n\n", "  ", ", ", ". ", "\t", " \n", "\r\n", " - ", "   ", " \r", "/ ", "' "]
    taken = 0
    for _ in range(12):
        parts, size, want = [], 0, rng.choice([2100, 3000, 6000, 20000])
        while size < want:
            w = rng.choice(words) + rng.choice(seps)
            parts.append(w)
            size += len(w.encode())
        text = "".join(parts)
        doc = text.encode()
        cuts, why = sim.mid_plan(doc)
        if cuts is None:
            assert why in ("no cut in a window", "letter -> space is not a certain start of this pattern"), why
            continue
        taken += 1
        whole = regex.findall(pat, text)
        assert "".join(whole) == text
        split = [p for a, b in zip(cuts, cuts[1:]) for p in regex.findall(pat, doc[a:b].decode())]
        assert split == whole, (name, cuts)
    assert taken >= 6, taken  # (every variation here keeps the rule: a blank behind a letter starts a piece)
