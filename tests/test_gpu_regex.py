"""GPU tests of the generic pat_str engine (tk_regex_kernels.h): a pat_str outside the three scanner families, split and encoded on the
device through the C ABI, against Python `regex` (the split) + the C oracle's byte_pair_encode (the tokens).  Bit-exact."""
import random

import numpy as np
import pytest
import regex

import helpers as h
from test_regex_engine import PATTERNS, py_starts, py_starts_gaps, random_text

pytestmark = pytest.mark.gpu
NAME = "o200k_shaped"


def make_core(pat, specials=None):
    from tiktoken_amd import CoreBPE

    return CoreBPE(h.golden_vocab(NAME), specials if specials is not None else h.load_golden(NAME)["special_tokens"], pat)


def make_docs(rng, count, big=0, cap=20000):
    """Short adversarial strings, awkward documents (cut at `cap` chars), `big` long ones with runs."""
    docs = [random_text(rng, rng.choice([0, 1, 3, 30, 300, 3000])) if rng.random() < 0.75 else h.fuzz_doc(rng)[:cap] for _ in range(count)]
    for _ in range(big):
        parts = []
        while sum(map(len, parts)) < 1_500_000:
            parts.append(rng.choice(["hello ", "World", " 12345", "\n", "x" * 5000, " " * 900, "中文", "é", "...", "CamelCase", " don't", "\r\n\r\n", "3.14"]))
        docs.append("".join(parts))
    return docs


def oracle_tokens(py_pat, docs, C, cache):
    toks, off = [], [0]
    for d in docs:
        for m in regex.finditer(py_pat, d):
            p = m.group().encode()
            t = cache.get(p)
            if t is None:
                t = cache[p] = C.encode_piece(p)
            toks += t
        off.append(len(toks))
    return np.array(toks, np.uint32), np.array(off, np.uint64)


# (patterns 9 and 13 leave gaps on most texts: test_text_the_pattern_does_not_match_yields_no_tokens)
@pytest.mark.parametrize("idx", [5, 6, 7, 8, 10, 11, 12, 14, 15, 16, 17, 18, 19, 20, 23, 25, 26, 27])  # (23: binary properties, 25: POSIX classes, 26: \h \H \O, 15 / 27: (?i) beyond ASCII)
def test_split_and_tokens_equal_python_regex_plus_oracle(idx):
    pat, py = PATTERNS[idx]
    py = py or pat
    core, C = make_core(pat), h.c_oracle_for(NAME)
    rng = random.Random(1000 + idx)
    docs = make_docs(rng, 200, big=1 if idx in (5, 6) else 0)
    enc = [d.encode() for d in docs]
    blob, off = h.pack(enc)
    want_starts, base = [], 0
    for d, e in zip(docs, enc):
        want_starts += [base + s for s in py_starts(py, d)]
        base += len(e)
    got = core.pretokenize_packed(blob, off)
    assert got.tolist() == want_starts + [len(blob)]
    rt, ro = oracle_tokens(py, docs, C, {})
    toks, toff = core.encode_batch_packed(blob, off)
    assert np.array_equal(toff, ro)
    assert np.array_equal(toks, rt)


def test_encoding_api_with_a_pattern_of_its_own():
    """The reference's "Extending tiktoken" recipe (README.md:83-94) with a pat_str the library has no hand-written scanner for."""
    from tiktoken_amd import Encoding

    pat = r"\p{Lu}?\p{Ll}+|\p{Lu}+(?!\p{Ll})|\d{1,3}|[^\s\p{L}\d]+|\s+|."
    specials = {"<|endoftext|>": 199999, "<|sep|>": 200000}
    enc = Encoding("camel", pat_str=pat, mergeable_ranks=h.golden_vocab(NAME), special_tokens=specials)
    C = h.c_oracle_for(NAME)
    text = "parseHTTPRequest42 isDone\n\n  fooBar_baz 1234567"
    want = [t for m in regex.finditer(r"\p{Lu}?\p{Ll}+|\p{Lu}+(?!\p{Ll})|\d{1,3}|[^\s\p{L}\d]+|\s+|(?s:.)", text) for t in C.encode_piece(m.group().encode())]
    assert enc.encode_ordinary(text) == want
    assert enc.encode(text) == want
    assert enc.decode(want) == text
    both = text + "<|sep|>" + text + "<|endoftext|>"
    assert enc.encode(both, allowed_special="all") == want + [200000] + want + [199999]
    assert enc.encode_batch([both, "", text], allowed_special={"<|sep|>"}, disallowed_special=()) == [
        want + [200000] + want + enc.encode_ordinary("<|endoftext|>"), [], want]
    with pytest.raises(ValueError):
        enc.encode(both)  # disallowed special, as in the reference (core.py:120-124)
    assert enc.encode_ordinary("hello world") == [t for p in (b"hello", b" ", b"world") for t in C.encode_piece(p)]


def test_special_tokens_batch():
    pat, py = PATTERNS[6]
    g = h.load_golden(NAME)
    core, C = make_core(pat), h.c_oracle_for(NAME)
    rng = random.Random(77)
    sp = list(g["special_tokens"])
    docs = []
    for _ in range(200):
        parts = []
        for _ in range(rng.randrange(0, 12)):
            parts.append(rng.choice(sp) if rng.random() < 0.3 else random_text(rng, rng.choice([0, 1, 5, 80, 700])))
        docs.append("".join(parts))
    alt = "(" + "|".join(regex.escape(s) for s in sorted(sp, key=len, reverse=True)) + ")"
    toks, off, cache = [], [0], {}
    for d in docs:
        for part in regex.split(alt, d):
            if part in g["special_tokens"]:
                toks.append(g["special_tokens"][part])
            else:
                t, _ = oracle_tokens(py, [part], C, cache)
                toks += t.tolist()
        off.append(len(toks))
    blob, doff = h.pack([d.encode() for d in docs])
    got, goff = core.encode_batch_packed(blob, doff, "all")
    assert np.array_equal(goff, np.array(off, np.uint64))
    assert np.array_equal(got, np.array(toks, np.uint32))


def test_text_the_pattern_does_not_match_yields_no_tokens():
    """find_iter (src/lib.rs:365,405) goes on behind text the pattern does not match: the reference drops it.  Split and tokens against
    Python `regex.finditer` + the oracle's byte_pair_encode, for patterns that leave gaps on most texts."""
    C = h.c_oracle_for(NAME)
    core = make_core(r"\w+|\s+")
    blob, off = h.pack([b"fine words only", b"hello, world", "¡hola! ¿qué?".encode(), b"", b"!!", b"..."])
    toks, toff = core.encode_batch_packed(blob, off)
    rt, ro = oracle_tokens(r"\w+|\s+", ["fine words only", "hello, world", "¡hola! ¿qué?", "", "!!", "..."], C, {})
    assert np.array_equal(toff, ro) and np.array_equal(toks, rt)
    assert core.pretokenize_packed(blob, off).tolist()[:8] == [0, 4, 5, 10, 11, 15, 20, 21] and 20 in core.last_gaps.tolist()
    assert core.encode_ordinary("fine words only") == make_core(r"\w+|\s+|[^\w\s]+").encode_ordinary("fine words only")
    assert core.encode_ordinary("?!") == []
    for idx, pat in ((9, None), (13, None), (24, None), (-1, r"\p{L}+|\d"), (-2, r"[a-m]+(?=[n-z])|\s")):  # (24: binary properties, titlecase letters are gaps)
        pat, py = (PATTERNS[idx][0], PATTERNS[idx][1] or PATTERNS[idx][0]) if idx >= 0 else (pat, pat)
        core = make_core(pat)
        rng = random.Random(4000 + idx)
        # (documents cut at 1000 chars: a pattern like ` ?\p{L}+(?='s)` fails at every char of a run of letters after scanning it to its end --
        #  quadratic in the run for any backtracking matcher, the reference's included, and a GPU lane is a slow place for that)
        docs = make_docs(rng, 150, big=1 if idx == -1 else 0, cap=1000)
        blob, off = h.pack([d.encode() for d in docs])
        want, wgap, base = [], [], 0
        for d in docs:
            st, gp = py_starts_gaps(py, d)
            want += [base + v for v in st]
            wgap += [base + v for v in gp]
            base += len(d.encode())
        assert len(wgap) > 100
        got = core.pretokenize_packed(blob, off)
        assert got.tolist() == want + [len(blob)] and core.last_gaps.tolist() == wgap, pat
        rt, ro = oracle_tokens(py, docs, C, {})
        toks, toff = core.encode_batch_packed(blob, off)
        assert np.array_equal(toff, ro) and np.array_equal(toks, rt), pat


def test_deep_backtracking_is_refused_loudly(monkeypatch):
    """A backtracking repeated group in the middle of an alternative: the program keeps a frame per repetition and gives up loudly when its
    stack is full.  The pattern's DFA (what runs unless $TIKTOKEN_AMD_RX_MATCHER says otherwise; the reference hands a pattern without
    look-around to the `regex` crate, which does not backtrack either) has no stack to exhaust and splits the text."""
    monkeypatch.setenv("TIKTOKEN_AMD_RX_MATCHER", "program")
    core = make_core(r"(?:\w\w)*\w!|\w|!")
    assert core.encode_ordinary("abc!abc!") == make_core(r"\w\w\w!").encode_ordinary("abc!abc!")
    with pytest.raises(ValueError, match="possessive"):
        core.encode_ordinary("ab" * 500)
    monkeypatch.delenv("TIKTOKEN_AMD_RX_MATCHER")
    table = make_core(r"(?:\w\w)*\w!|\w|!")
    assert table.encode_ordinary("abc!abc!") == core.encode_ordinary("abc!abc!")
    assert table.encode_ordinary("ab" * 500) == make_core(r"\w").encode_ordinary("ab" * 500)
    # (a pattern without a DFA -- look-behind of two chars -- keeps the program and its limits)
    behind = make_core(r"(?:\w\w)*\w!|(?<=ab)a|\w|!")
    with pytest.raises(ValueError, match="possessive"):
        behind.encode_ordinary("ab" * 500)
    from tiktoken_amd import CoreBPE

    with pytest.raises(ValueError, match="look-behind"):
        CoreBPE(h.golden_vocab(NAME), {}, r"(?<=a+b)c|.")


def test_one_large_document_and_multi_chunk(monkeypatch):
    """A single 6 MiB document (the resolving pass is one lane: it must live off the speculative pass) and the same batch cut into chunks."""
    pat, py = PATTERNS[5]
    rng = random.Random(5)
    words = ["alpha", "Beta", " ", "  ", "\n", "42", "...", "中文", "don't", "é", "_", "x" * 300]
    doc = "".join(rng.choice(words) for _ in range(1_200_000))
    docs = [doc, "tail doc", ""]
    C = h.c_oracle_for(NAME)
    rt, ro = oracle_tokens(py or pat, docs, C, {})
    blob, off = h.pack([d.encode() for d in docs])
    core = make_core(pat)
    toks, toff = core.encode_batch_packed(blob, off)
    assert np.array_equal(toff, ro) and np.array_equal(toks, rt)
    many = make_docs(rng, 300)
    rt, ro = oracle_tokens(py or pat, many, C, {})
    blob, off = h.pack([d.encode() for d in many])
    monkeypatch.setenv("TIKTOKEN_AMD_CHUNK_BYTES", "65536")
    small_chunks = make_core(pat)
    toks, toff = small_chunks.encode_batch_packed(blob, off)
    assert np.array_equal(toff, ro) and np.array_equal(toks, rt)


@pytest.mark.parametrize("name,mix,nbytes", [("gpt2_shaped", 2, 4 << 20), ("cl100k_shaped", 0, 8 << 20), ("o200k_shaped", 1, 16 << 20)])
def test_stock_patterns_through_the_generic_engine_equal_the_scanners(name, mix, nbytes, monkeypatch):
    """The stock pat_str compiled for the generic engine (TIKTOKEN_AMD_DEBUG=1048576) against the hand-written scanners of its family on
    the bench corpora, with and without special tokens: two independent implementations of the split, every token equal (and equal to the
    oracle: test_gpu_parity.py holds the scanners to it on the same corpora)."""
    from tiktoken_amd import CoreBPE

    g = h.load_golden(name)
    blob, off = h.gen_corpus(0x5EED0100 + mix, mix, nbytes)
    blob, off = h.insert_specials(blob, off)
    top = max(max(g["special_tokens"].values()), max(h.golden_vocab(name).values()))
    specials = {**g["special_tokens"], **{f"<|custom_{i}|>": top + 1 + i for i in range(8)}}
    scanners = CoreBPE(h.golden_vocab(name), specials, g["pat_str"])
    monkeypatch.setenv("TIKTOKEN_AMD_DEBUG", "1048576")
    generic = {}
    for form in ("flat", "dfa", "program"):  # the pattern's DFA with the one-loop speculative pass (the default), piece by piece, the backtracking program
        monkeypatch.setenv("TIKTOKEN_AMD_RX_MATCHER", form)
        generic[form] = CoreBPE(h.golden_vocab(name), specials, g["pat_str"])
    monkeypatch.delenv("TIKTOKEN_AMD_RX_MATCHER")
    monkeypatch.delenv("TIKTOKEN_AMD_DEBUG")
    for allowed in (None, "all"):
        t1, o1 = scanners.encode_batch_packed(blob, off, allowed)
        p1 = scanners.pretokenize_packed(blob, off, allowed)
        for form, core in generic.items():
            t2, o2 = core.encode_batch_packed(blob, off, allowed)
            assert np.array_equal(o1, o2), (allowed, form)
            assert np.array_equal(t1, t2), (allowed, form)
            assert np.array_equal(p1, core.pretokenize_packed(blob, off, allowed)), (allowed, form)


def test_staged_speculative_pass_with_several_stretches_per_workgroup(monkeypatch):
    """tk_k_rx_speculate_staged walks the chunk with the stride of its grid: a workgroup takes a second stretch of 256 segments only in
    chunks of more than 2 GiB (65 536 workgroups x 32 KiB).  $TIKTOKEN_AMD_RX_GRID_CAP makes that loop run on a small input: three
    workgroups over 6 MiB (64 stretches each; the codes in LDS and the barrier between two stretches are what this covers), special tokens
    and documents that end inside a stretch included -- every token and every piece start equal to the default grid's and the scanners'."""
    from tiktoken_amd import CoreBPE

    name = "o200k_shaped"
    g = h.load_golden(name)
    blob, off = h.gen_corpus(0x5EED0177, 1, (6 << 20) + 777)
    blob, off = h.insert_specials(blob, off)
    top = max(max(g["special_tokens"].values()), max(h.golden_vocab(name).values()))
    specials = {**g["special_tokens"], **{f"<|custom_{i}|>": top + 1 + i for i in range(8)}}
    scanners = CoreBPE(h.golden_vocab(name), specials, g["pat_str"])
    monkeypatch.setenv("TIKTOKEN_AMD_DEBUG", "1048576")
    wide = CoreBPE(h.golden_vocab(name), specials, g["pat_str"])
    monkeypatch.setenv("TIKTOKEN_AMD_RX_GRID_CAP", "3")
    narrow = CoreBPE(h.golden_vocab(name), specials, g["pat_str"])
    monkeypatch.setenv("TIKTOKEN_AMD_RX_GRID_CAP", "1")
    one = CoreBPE(h.golden_vocab(name), specials, g["pat_str"])
    monkeypatch.delenv("TIKTOKEN_AMD_RX_GRID_CAP")
    monkeypatch.delenv("TIKTOKEN_AMD_DEBUG")
    for allowed in (None, "all"):
        t1, o1 = scanners.encode_batch_packed(blob, off, allowed)
        p1 = scanners.pretokenize_packed(blob, off, allowed)
        for core in (wide, narrow, one):
            t2, o2 = core.encode_batch_packed(blob, off, allowed)
            assert np.array_equal(o1, o2) and np.array_equal(t1, t2), allowed
            assert np.array_equal(p1, core.pretokenize_packed(blob, off, allowed)), allowed


def test_exploding_backtracking_is_stopped(monkeypatch):
    """Nested quantifiers on a text that makes them explode ((?:a+)+b on a run of a's): fancy-regex gives up after 1 000 000 backtracks
    (Error::BacktrackLimitExceeded -- a panic in the reference, src/lib.rs:365); the GPU program has the same kind of budget, so the call
    fails instead of hanging the device.  A pattern that has a DFA never backtracks (nor does the reference for a pattern without
    look-around: fancy-regex hands it to the `regex` crate): linear, and right."""
    monkeypatch.setenv("TIKTOKEN_AMD_RX_MATCHER", "program")
    core = make_core(r"(?:a+)+b|[\s\S]")
    assert core.encode_ordinary("aaab aab") == make_core(r"a+b|[\s\S]").encode_ordinary("aaab aab")
    with pytest.raises(ValueError, match="backtrack limit"):
        core.encode_ordinary("a" * 26)
    monkeypatch.delenv("TIKTOKEN_AMD_RX_MATCHER")
    table = make_core(r"(?:a+)+b|[\s\S]")
    assert table.encode_ordinary("a" * 26 + " aab") == make_core(r"[\s\S]").encode_ordinary("a" * 26 + " ") + make_core(r"a+b").encode_ordinary("aab")
    with pytest.raises(ValueError, match="backtrack limit"):  # (look-ahead of two chars: no DFA)
        make_core(r"(?:a+)+b(?!xy)|[\s\S]").encode_ordinary("a" * 26)


def test_golden_vectors_of_generic_patterns():
    """tests/golden/generic_patterns.json.gz: ten pat_str outside the scanner families, encoded by the reference's own Python code
    (tiktoken/_educational.py SimpleBytePairEncoding: regex.findall + bpe_encode) with the vocabulary its own bpe_train produced
    (tools/gen_golden_generic.py).  The GPU path has to give the same token ids."""
    from tiktoken_amd import CoreBPE
    from test_regex_engine import _generic_golden

    ranks = h.golden_vocab("edu600")
    n = 0
    for p in _generic_golden():
        core = CoreBPE(ranks, {}, p["pat_str"])
        blob, off = h.pack([c["text"] for c in p["cases"]])
        toks, toff = core.encode_batch_packed(blob, off)
        want = [t for c in p["cases"] for t in c["tokens"]]
        assert toff.tolist() == np.cumsum([0] + [len(c["tokens"]) for c in p["cases"]]).tolist(), p["pat_str"]
        assert toks.tolist() == want, p["pat_str"]
        for c in p["cases"][:12]:  # (and one by one: the single-document entry)
            assert core.encode_ordinary(c["text"].decode()) == c["tokens"], (p["pat_str"], c["text"])
        n += len(want)
    assert n > 100_000


def generated_patterns_on_the_device(seed: int, n_patterns: int, with_tokens_every: int = 4, table_form: bool = False):
    """Random patterns over the whole supported syntax (the generator of the CPU test, tests/test_regex_engine.py::_gen_pattern) compiled
    and run ON THE DEVICE: split and gap chars of a batch of short texts against Python `regex`, tokens of every few patterns against the
    oracle's byte_pair_encode.  table_form: only patterns that have a DFA (the kernels' table form).  Returns (patterns run, patterns the
    device gave up on loudly)."""
    from test_regex_engine import _gen_pattern

    rng = random.Random(seed)
    alphabet = list("abcxABCX12 \n\t'.,sSkK") + ["ſ", "K", "é", "中", "É", "٣", "\r\n", "  ", "ab", "'s"]
    texts = ["".join(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 2, 4, 8, 20, 60, 400]))) for _ in range(150)]
    texts += ["a" * 300, " " * 200 + "x", "ab" * 20, "x1" * 25 + "\n", "'s" * 12]
    C = h.c_oracle_for(NAME)
    ran = gave_up = 0
    while ran + gave_up < n_patterns:
        eng, py = _gen_pattern(rng, table_form)
        if "(?i:" in eng and r"[^a\s]" in eng:  # (a scoping bug of `regex` 2026.7.19: see the CPU test)
            continue
        pyc = regex.compile(py)
        try:
            core = make_core(eng, {})
        except ValueError:
            continue  # refused by the compiler, with a reason (the CPU test checks the reasons)
        good, want, wgap, base = [], [], [], 0
        for t in texts:
            try:
                st, gp = py_starts_gaps(pyc, t, timeout=0.25)
            except (TimeoutError, LookupError):
                continue
            good.append(t)
            want += [base + s for s in st]
            wgap += [base + s for s in gp]
            base += len(t.encode())
        blob, off = h.pack([t.encode() for t in good])
        try:
            got = core.pretokenize_packed(blob, off)
        except ValueError as e:  # deep backtracking in a repeated group, or more of it than the budget: loud, as in the reference
            assert "possessive" in str(e) or "backtrack limit" in str(e), (eng, str(e))
            gave_up += 1
            continue
        assert got.tolist() == want + [len(blob)] and core.last_gaps.tolist() == wgap, (eng, py)
        if ran % with_tokens_every == 0:
            rt, ro = oracle_tokens(pyc, good, C, {})
            toks, toff = core.encode_batch_packed(blob, off)
            assert np.array_equal(toff, ro) and np.array_equal(toks, rt), (eng, py)
        ran += 1
    return ran, gave_up


def test_generated_patterns_on_the_device():
    ran, gave_up = generated_patterns_on_the_device(20260923, 40)
    assert ran >= 30, (ran, gave_up)
    ran, gave_up = generated_patterns_on_the_device(20260924, 40, table_form=True)
    # (a DFA has neither a stack nor a budget to exhaust; the odd generated pattern whose table would be too large keeps the program)
    assert ran + gave_up == 40 and ran >= 35, (ran, gave_up)
