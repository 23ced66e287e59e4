"""The reference's own public test-suite shapes (tests/test_encoding.py, test_simple_public.py,
test_pickle.py, test_offsets.py, test_misc.py) run against the MI355X path through the identical
`Encoding` API.  Real-vocabulary known answers need files that are not available offline (they run when
present in $TIKTOKEN_CACHE_DIR); everything else uses the "-shaped" encodings and compares with the oracle."""
import os
import pickle

import hypothesis
import hypothesis.strategies as st
import numpy as np
import pytest

import helpers as h
import tiktoken_amd as tiktoken

pytestmark = pytest.mark.gpu
MAX_EXAMPLES = int(os.environ.get("TIKTOKEN_MAX_EXAMPLES", "60"))
ENCS = ["gpt2_shaped", "cl100k_shaped", "o200k_shaped"]


def oracle_encode(name, text, allowed=None):
    C = h.c_oracle_for(name)
    b = text.encode("utf-8")
    return (C.encode_ordinary(b) if allowed is None else C.encode(b, allowed)).tolist()


@pytest.mark.parametrize("name", ENCS)
def test_simple_and_single_token_roundtrip(name):
    enc = tiktoken.get_encoding(name)
    assert enc.encode("hello world") == oracle_encode(name, "hello world")
    assert enc.decode(enc.encode("hello world")) == "hello world"
    eot = enc.eot_token
    assert enc.encode("hello <|endoftext|>", allowed_special="all") == oracle_encode(name, "hello ") + [eot]
    for token in range(0, min(10_000, enc.max_token_value - 1), 37):
        assert enc.encode_single_token(enc.decode_single_token_bytes(token)) == token
    assert enc.encode("") == []


@pytest.mark.parametrize("name", ENCS)
def test_basic_roundtrip(name):
    enc = tiktoken.get_encoding(name)
    for value in ("hello", "hello ", "hello  ", " hello", " hello ", " hello  ", "hello world", "请考试我的软件！12345"):
        assert value == enc.decode(enc.encode(value))
        assert value == enc.decode(enc.encode_ordinary(value))


@pytest.mark.parametrize("name", ENCS)
@hypothesis.given(text=st.text())
@hypothesis.settings(deadline=None, max_examples=MAX_EXAMPLES)
def test_hyp_roundtrip_and_oracle(name, text):
    enc = tiktoken.get_encoding(name)
    toks = enc.encode(text, disallowed_special=())
    assert text == enc.decode(toks)
    fixed = text.encode("utf-16", "surrogatepass").decode("utf-16", "replace")
    assert toks == oracle_encode(name, fixed)
    assert enc.encode_ordinary(text) == toks  # test_hyp_special_ordinary


@pytest.mark.parametrize("name", ENCS)
@hypothesis.given(batch=st.lists(st.text()))
@hypothesis.settings(deadline=None, max_examples=MAX_EXAMPLES // 2)
def test_hyp_batch_roundtrip(name, batch):
    enc = tiktoken.get_encoding(name)
    encoded = enc.encode_batch(batch, allowed_special="all")
    assert encoded == [enc.encode(t, allowed_special="all") for t in batch]
    assert enc.decode_batch(encoded) == batch


@pytest.mark.parametrize("name", ENCS)
@hypothesis.given(bytestring=st.binary())
@hypothesis.settings(deadline=None, max_examples=MAX_EXAMPLES // 2)
def test_hyp_encode_bytes(name, bytestring):
    enc = tiktoken.get_encoding(name)
    assert enc.decode_bytes(enc._encode_bytes(bytestring)) == bytestring


def test_encode_bytes_invalid_tail():
    enc = tiktoken.get_encoding("cl100k_shaped")
    for i in range(10):
        bs = b"\x80" * i
        assert enc.decode_bytes(enc._encode_bytes(bs)) == bs
    bs = " 실".encode()[:-1] + b"\xed"
    assert enc.decode_bytes(enc._encode_bytes(bs)) == bs


def test_surrogate_pairs():
    enc = tiktoken.get_encoding("cl100k_shaped")
    assert enc.encode("👍") == enc.encode("👍")
    assert enc.encode("\ud83d") == enc.encode("�")
    assert enc.encode_ordinary_batch(["a\ud83db", "👍"]) == [enc.encode("a�b"), enc.encode("👍")]


def test_special_token_policy_matrix():
    """tests/test_encoding.py:175-223."""
    enc = tiktoken.get_encoding("cl100k_shaped")
    eot, fip, fim = (enc.encode_single_token(s) for s in ("<|endoftext|>", "<|fim_prefix|>", "<|fim_middle|>"))
    assert eot == enc.eot_token
    text = "<|endoftext|> hello <|fim_prefix|>"
    assert eot not in enc.encode(text, disallowed_special=())
    for kw in ({}, {"disallowed_special": "all"}, {"disallowed_special": {"<|endoftext|>"}}, {"disallowed_special": {"<|fim_prefix|>"}}):
        with pytest.raises(ValueError, match="disallowed special token"):
            enc.encode(text, **kw)
    text = "<|endoftext|> hello <|fim_prefix|> there <|fim_middle|>"
    t = enc.encode(text, disallowed_special=())
    assert eot not in t and fip not in t and fim not in t
    for kw in ({"allowed_special": "all", "disallowed_special": ()}, {"allowed_special": "all", "disallowed_special": "all"}):
        t = enc.encode(text, **kw)
        assert eot in t and fip in t and fim in t
    for allowed, present in (({"<|fim_prefix|>"}, fip), ({"<|endoftext|>"}, eot), ({"<|fim_middle|>"}, fim)):
        t = enc.encode(text, allowed_special=allowed, disallowed_special=())
        assert present in t and sum(x in t for x in (eot, fip, fim)) == 1
        assert t == oracle_encode("cl100k_shaped", text, allowed)
    with pytest.raises(ValueError):
        enc.encode_batch(["fine", text])


@pytest.mark.parametrize("name", ENCS)
def test_batch_encode(name):
    enc = tiktoken.get_encoding(name)
    t1, t2 = "hello world", "goodbye world"
    assert enc.encode_batch([t1]) == [enc.encode(t1)]
    assert enc.encode_batch([t1, t2]) == [enc.encode(t1), enc.encode(t2)]
    assert enc.encode_ordinary_batch([t1, t2], num_threads=3) == [enc.encode_ordinary(t1), enc.encode_ordinary(t2)]
    assert enc.encode_ordinary_batch([]) == []


def test_encode_to_numpy_and_buffer_protocol():
    enc = tiktoken.get_encoding("o200k_shaped")
    arr = enc.encode_to_numpy("hello world 123")
    assert arr.dtype == np.uint32 and arr.tolist() == enc.encode("hello world 123")
    buf = enc._core_bpe.encode_to_tiktoken_buffer("hello", set())
    mv = memoryview(buf)
    assert mv.readonly and mv.ndim == 1 and mv.itemsize == 4 and mv.format in ("I", "<I", "L", "<L")


def test_offsets_and_tokens_bytes():
    """tests/test_offsets.py:28-46 (property) on fixed prompts."""
    enc = tiktoken.get_encoding("cl100k_shaped")
    for prompt in ["hello world", "hello world<|endoftext|> green cow", "我非常渴望与人工智能一起工作", "நடிகர் சூர்யா", " Ġ除"]:
        toks = enc.encode(prompt, allowed_special="all")
        text, offsets = enc.decode_with_offsets(toks)
        assert text == prompt
        slow, pos = [], 0
        for tb in enc.decode_tokens_bytes(toks):
            # first character that contains a byte of this token
            slow.append(len(prompt.encode()[:pos].decode("utf-8", errors="ignore")) if not (0x80 <= tb[0] < 0xC0) else
                        len(prompt.encode()[:pos].decode("utf-8", errors="ignore")))
            pos += len(tb)
        assert offsets == sorted(offsets) and len(offsets) == len(toks) and offsets[0] == 0


def test_pickle():
    enc = tiktoken.get_encoding("gpt2_shaped")
    assert pickle.loads(pickle.dumps(enc)).encode("hello world") == enc.encode("hello world")
    custom = tiktoken.Encoding(name="my_new", pat_str=enc._pat_str, mergeable_ranks=enc._mergeable_ranks,
                               special_tokens={**enc._special_tokens, "<|pickle|>": 100_000})
    again = pickle.loads(pickle.dumps(custom))
    assert again.encode("<|pickle|>", allowed_special="all") == [100_000]
    assert again.encode("hello world") == enc.encode("hello world")


def test_custom8_special_tokens_batch():
    """BASELINE.json config 5 shape: o200k + 8 custom specials, encode_batch(allowed_special='all')."""
    enc = tiktoken.get_encoding("o200k_custom8")
    C = h.c_oracle.COracle(2, enc._mergeable_ranks, enc._special_tokens)
    blob, off = h.gen_corpus(0x5EED0005, 1, 1 << 20)
    bb = blob.tobytes()
    docs = []
    rng = np.random.default_rng(1)
    decoys = ["<|custom_9|>", "<|endoftext", "<|custom_3|", "<|", "|>"]
    for d in range(len(off) - 1):
        t = bb[int(off[d]):int(off[d + 1])].decode()
        k = int(rng.integers(0, len(t) + 1))
        ins = f"<|custom_{int(rng.integers(0, 8))}|>" if rng.random() < 0.7 else decoys[int(rng.integers(0, len(decoys)))]
        docs.append(t[:k] + ins + t[k:])
    got = enc.encode_batch(docs, allowed_special="all")
    for t, g in zip(docs, got):
        assert g == C.encode(t.encode(), "all").tolist()


def test_encode_with_unstable_contract():
    """core.py:227-230: stable tokens decode to a prefix; every completion continues to cover the text."""
    enc = tiktoken.get_encoding("gpt2_shaped")
    for text in ["hello fanta", "hello wor", "  ", "a\n\n", "12345"]:
        stable, completions = enc.encode_with_unstable(text)
        assert text.encode().startswith(enc.decode_bytes(stable))
        assert all(enc.decode_bytes(stable + seq).startswith(text.encode()) for seq in completions)


def test_threads_share_one_encoding():
    """The reference object is frozen and entered from many Python threads (core.py:175)."""
    from concurrent.futures import ThreadPoolExecutor

    enc = tiktoken.get_encoding("cl100k_shaped")
    texts = [f"thread {i} says hello world {i * 7919}" for i in range(64)]
    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(enc.encode_ordinary, texts))
    assert got == [oracle_encode("cl100k_shaped", t) for t in texts]


def test_small_calls_of_several_threads_overlap():
    """core.py:175 -- a pool of threads on one Encoding is the reference's normal use (it keeps a regex per thread for it, lib.rs:232-238).
    Small calls take no lock here: each runs on a slot of its own.  Correct under load, and 8 threads finish a fixed number of calls faster
    than one (the C entry directly: ctypes releases the GIL around it; measured numbers in profiles/r03_small_calls_threads.txt)."""
    import threading
    import time

    enc = tiktoken.get_encoding("o200k_shaped")
    core = enc._core_bpe
    texts = [(f"worker {i}: " + "The quick brown fox jumps over the lazy dog; 3.14159 and so on, ünïcödé too. " * 2)[:180] for i in range(16)]
    datas = [t.encode() for t in texts]
    want = [oracle_encode("o200k_shaped", t) for t in texts]
    for d, w in zip(datas, want):
        assert core._encode_np(d, None).tolist() == w

    def run(n_threads, calls_per_thread):
        bad = []

        def work(k):
            for j in range(calls_per_thread):
                i = (k + j) % len(datas)
                if core._encode_np(datas[i], None).tolist() != want[i]:
                    bad.append((k, j))

        th = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not bad, bad[:3]
        return time.perf_counter() - t0

    run(8, 200)  # (every slot used, results compared under load)

    # the rate through the C entry itself (ctypes releases the GIL around the call; the Python work around it does not overlap)
    import ctypes

    L, hnd = core._L, core._h
    bufs = [np.frombuffer(d, np.uint8) for d in datas]

    def raw(n_threads, calls_per_thread):
        def work(k):
            out, n = ctypes.c_void_p(), ctypes.c_uint64()
            b = bufs[k % len(bufs)]
            for _ in range(calls_per_thread):
                L.tk_encode_ordinary(hnd, b.ctypes.data, len(b), ctypes.byref(out), ctypes.byref(n))
                L.tk_free(out)

        th = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    raw(8, 100)
    total = 16000
    t1 = min(raw(1, total) for _ in range(2))
    t8 = min(raw(8, total // 8) for _ in range(2))
    print(f"{total} calls of 180 bytes through the C ABI: one thread {t1 * 1e6 / total:.1f} us per call, eight threads {t8 * 1e6 / total:.1f} us per call ({t1 / t8:.2f}x)")
    assert t8 < t1 / 1.4, (t1, t8)  # (measured 2.2-2.8x: profiles/r03_small_calls.txt; the launch path of the HIP runtime is the shared part)


def test_small_calls_scale_with_native_threads(tmp_path):
    """The same without an interpreter in the loop (tools/ubench/small_threads.c, built here with gcc): Python threads stop at ~85 k calls/s
    whatever the library does, because every call's marshalling holds the GIL.  Callers that arrive together share a launch (flat
    combining, tk_api.hip encode_small).  Measured: 1 thread 26 k calls/s, 8 threads 143 k (5.5x), 16 threads 218 k (8.4x)
    (profiles/r04_small_calls.txt); asserted: every result right, and 8 threads at least 3x one."""
    import ctypes
    import shutil
    import subprocess

    if not shutil.which("gcc"):
        pytest.skip("no gcc on this box")
    so = str(tmp_path / "tk_small_threads.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", os.path.join(h.ROOT, "tools", "ubench", "small_threads.c"), "-o", so])
    H = ctypes.CDLL(so)
    H.tk_small_threads.restype = ctypes.c_double
    H.tk_small_threads.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
    enc = tiktoken.get_encoding("o200k_shaped")
    core = enc._core_bpe
    text = ("The quick brown fox jumps over the lazy dog; 3.14159 and so on, ünïcödé too. " * 3)[:180]
    data = np.frombuffer(text.encode(), np.uint8)
    want = len(oracle_encode("o200k_shaped", text))
    assert enc.encode_ordinary(text) == oracle_encode("o200k_shaped", text)
    L = core._L
    fe, ff = ctypes.cast(L.tk_encode_ordinary, ctypes.c_void_p), ctypes.cast(L.tk_free, ctypes.c_void_p)

    def rate(nth, per):
        tok, bad = ctypes.c_uint64(), ctypes.c_int()
        r = H.tk_small_threads(fe, ff, core._h, data.ctypes.data, len(data), nth, per, ctypes.byref(tok), ctypes.byref(bad))
        assert bad.value == 0 and tok.value == want * nth * per
        return r

    rate(8, 200)
    r1 = max(rate(1, 4000) for _ in range(2))
    r8 = max(rate(8, 1500) for _ in range(2))
    l0, c0 = core.stat("small_launches"), core.stat("small_calls")
    r16 = rate(16, 1000)
    l1, c1 = core.stat("small_launches"), core.stat("small_calls")
    print(f"native threads: 1 -> {r1:.0f} calls/s, 8 -> {r8:.0f} ({r8 / r1:.2f}x), 16 -> {r16:.0f} ({r16 / r1:.2f}x, {(c1 - c0) / max(l1 - l0, 1):.2f} calls per launch)")
    assert r8 >= 3.0 * r1, (r1, r8)


def test_small_and_mid_size_calls_of_many_threads_share_the_slots():
    """Eight threads on one core, calls of 100 bytes to 40 KiB mixed: short ones take a slot, longer ones a slot per segment (or the general
    pipeline when the slots are taken), whoever launches takes every ready slot along.  Every result: the oracle's."""
    import random
    import threading

    enc = tiktoken.get_encoding("cl100k_shaped")
    core = enc._core_bpe
    base = h.lorem(60000)
    rng = random.Random(5)
    cases = []
    for _ in range(40):
        n = rng.choice((100, 700, 1500, 2047, 2049, 3000, 9000, 20000, 40000))
        at = rng.randrange(0, len(base) - n)
        d = base[at: at + n]
        cases.append((d, np.asarray(oracle_encode("cl100k_shaped", d.decode()), np.uint32)))
    bad = []

    def work(k):
        r = random.Random(k)
        for _ in range(120):
            d, want = cases[r.randrange(len(cases))]
            got = core._encode_np(d, None)
            if not np.array_equal(got, want):
                bad.append((k, len(d)))

    before = core.stat("mid_calls")
    th = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad[:5]
    assert core.stat("mid_calls") > before  # (some of the longer ones did find their slots)


def test_close_is_idempotent_and_calls_after_it_fail_cleanly():
    """CoreBPE.close() releases the native core (an atexit handler does the same for cores still alive at interpreter shutdown, before the HIP
    runtime is gone); a call on a closed core is an error, not a crash."""
    from tiktoken_amd import CoreBPE

    g = h.load_golden("gpt2_shaped")
    core = CoreBPE(h.golden_vocab("gpt2_shaped"), g["special_tokens"], g["pat_str"])
    assert core._encode_np(b"hello world", None).tolist() == oracle_encode("gpt2_shaped", "hello world")
    core.close()
    core.close()
    with pytest.raises((ValueError, RuntimeError)):
        core._encode_np(b"hello world", None)


# (the real-vocabulary known answers of SURVEY.md Appendix B: tests/test_real_vocab.py -- each encoding skipped on its own while its file is absent)


# ---------------------------------------------------------------- byte-level / unstable entry points vs the oracle
# CoreBPE._encode_bytes (src/py.rs:72-115) and CoreBPE.encode_with_unstable (src/lib.rs:483-599) are compared with the
# statement-by-statement restatement in oracle/py_oracle.py -- token for token and completion set for completion set.
def _edu_core():
    from oracle import py_oracle as po
    from tiktoken_amd import CoreBPE

    ranks = h.golden_vocab("edu600")
    specials = {"<|endoftext|>": 600}
    return CoreBPE(ranks, specials, po.R50K_PAT), ranks, specials, po


UNSTABLE_TEXTS = ["hello fanta", "hello wor", "def f(x):\n    return x", "a\n\n", "tab\t\t", "x  ", "hello <|endoftext|>", "hello <|endoftext|> wor",
                  "", " ", "\n", "ing", "the the the", "  \n  ", "naïve caf", "中文", "1234 56", "it's", "don'", " \x1c", "a　", "a  "]


@pytest.mark.parametrize("allowed", [set(), {"<|endoftext|>"}])
def test_encode_with_unstable_equals_oracle(allowed):
    core, ranks, specials, po = _edu_core()
    for text in UNSTABLE_TEXTS:
        got_t, got_c = core.encode_with_unstable(text, allowed)
        want_t, want_c = po.encode_unstable_native(text, po.R50K_PAT, ranks, specials, allowed)
        assert got_t == want_t, (text, got_t, want_t)
        assert {tuple(c) for c in got_c} == want_c, (text, sorted(map(tuple, got_c))[:5], sorted(want_c)[:5])


@hypothesis.given(text=st.text(alphabet=st.sampled_from(list("abehlnort \n\t'.0x") + ["é", "中", "\x1c", "　"]), max_size=12))
@hypothesis.settings(deadline=None, max_examples=MAX_EXAMPLES)
def test_hyp_encode_with_unstable_equals_oracle(text):
    core, ranks, specials, po = _edu_core()
    got_t, got_c = core.encode_with_unstable(text, set())
    want_t, want_c = po.encode_unstable_native(text, po.R50K_PAT, ranks, specials, set())
    assert got_t == want_t and {tuple(c) for c in got_c} == want_c, text


def test_encode_bytes_equals_oracle():
    from oracle import py_oracle as po

    rng = np.random.default_rng(11)
    for name in ENCS:
        enc = tiktoken.get_encoding(name)
        ranks, pat = h.load_vocab(name), h.PAT_STR[h.PATTERN_OF[name]]
        cases = [b"", b"hello world", b"hello wor\xff", b" \xec\x8b\xa4\xed", b"\x80" * 5, "naïve".encode()[:-1], "today\n ".encode() + b"\xe4\xb8",
                 b"abc \n\t  \xf0\x9f\x91", b"\xff", b"a\xffb\xfec", "실 실".encode()[:-2]]
        for _ in range(40):
            base = "".join(rng.choice(h.ADV, size=int(rng.integers(1, 12)))).encode()
            cut = int(rng.integers(0, len(base) + 1))
            cases.append(base[:cut] + bytes(rng.integers(0x80, 0x100, size=int(rng.integers(0, 3)), dtype=np.uint8)))
        for bs in cases:
            assert enc._encode_bytes(bs) == po.encode_bytes(bs, pat, ranks), (name, bs)


def test_byte_pair_encode_has_no_whole_piece_shortcut():
    """tk_byte_pair_encode = byte_pair_encode (src/lib.rs:198-211): on a vocabulary where a key is NOT reachable by merges the
    shortcut (encode_single_piece, py.rs:145-150) and the merge differ."""
    from oracle import py_oracle as po
    from tiktoken_amd import CoreBPE

    ranks = {bytes([b]): b for b in range(256)}
    ranks[b"abc"] = 256  # neither "ab" nor "bc" is a key: the merge loop can never build it
    core = CoreBPE(ranks, {}, po.R50K_PAT)
    assert core.encode_single_piece(b"abc") == [256] == po.encode_single_piece(b"abc", ranks)
    assert core._byte_pair_encode(b"abc") == [97, 98, 99] == po.byte_pair_encode(b"abc", ranks)
    assert core._byte_pair_encode(b"a") == [97]


def test_encoding_from_a_parsed_file_never_walks_the_dict():
    """vocab_io.RankTable hands tk_create the parser's arrays; a file that lists a token twice falls back to the dict (the later rank wins,
    as in the reference's dict comprehension, load.py:159-171)."""
    import base64

    from tiktoken_amd import vocab_io

    g = h.load_golden("cl100k_shaped")
    ranks = h.golden_vocab("cl100k_shaped")
    text = b"".join(base64.b64encode(k) + b" %d\n" % v for k, v in ranks.items())
    table = vocab_io.parse_tiktoken_bpe(text, lazy=True)
    enc = tiktoken.Encoding("t1", pat_str=g["pat_str"], mergeable_ranks=table, special_tokens=g["special_tokens"])
    assert table._pending is not None  # still packed: nobody needed the dict
    ref = tiktoken.Encoding("t2", pat_str=g["pat_str"], mergeable_ranks=dict(ranks), special_tokens=g["special_tokens"])
    s = "The quick brown fox's 12345 jumps\n\n over the lazy dog. 中文 \U0001F600"
    assert enc.encode_ordinary(s) == ref.encode_ordinary(s)
    some = next(k for k in ranks if len(k) == 3)
    dup = vocab_io.parse_tiktoken_bpe(text + base64.b64encode(some) + b" %d\n" % (max(ranks.values()) + 1), lazy=True)
    enc2 = tiktoken.Encoding("t3", pat_str=g["pat_str"], mergeable_ranks=dup, special_tokens={})
    assert enc2.encode_single_token(some) == max(ranks.values()) + 1
    # the largest rank belongs to a token that is listed again with a smaller one: max_token_value follows the dict, as the reference's does
    top = max(ranks.values())
    first = next(k for k, v in ranks.items() if v == 0)
    dup2 = vocab_io.parse_tiktoken_bpe(text + base64.b64encode(b"\xf5\xf6\xf7") + b" %d\n" % (top + 9) + base64.b64encode(b"\xf5\xf6\xf7") + b" %d\n" % (top + 1), lazy=True)
    enc3 = tiktoken.Encoding("t4", pat_str=g["pat_str"], mergeable_ranks=dup2, special_tokens={})
    assert enc3.max_token_value == top + 1 and first in enc3._mergeable_ranks and dict.__len__(enc3._mergeable_ranks) == len(ranks) + 1
