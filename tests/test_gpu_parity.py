"""GPU parity tests: the HIP encode path (through the C ABI) against the CPU oracle and the golden
fixtures generated from the reference's own Python code (tools/gen_golden.py).  Bit-exact."""
import numpy as np
import pytest

import helpers as h

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cores():
    from tiktoken_amd import CoreBPE

    out = {}
    for name in h.ENCODING_NAMES + ["edu600"]:
        g = h.load_golden(name)
        out[name] = CoreBPE(h.golden_vocab(name), g["special_tokens"], g["pat_str"])
    return out


@pytest.mark.parametrize("name", h.ENCODING_NAMES + ["edu600"])
def test_golden_vectors(cores, name):
    """Every fixture case, all ordinary cases in ONE batch call (documents never interact, core.py:174-176)."""
    g = h.load_golden(name)
    core = cores[name]
    ordinary = [c for c in g["cases"] if c["allowed"] is None]
    blob, off = h.pack([c["text"] for c in ordinary])
    toks, toff = core.encode_batch_packed(blob, off)
    bad = []
    for i, c in enumerate(ordinary):
        got = toks[int(toff[i]):int(toff[i + 1])].tolist()
        if got != c["tokens"]:
            bad.append((c["name"], c["text"][:60], got[:12], c["tokens"][:12]))
    assert not bad, bad[:5]
    # special-token cases, grouped by allowed set (src/lib.rs:375-442)
    groups = {}
    for c in g["cases"]:
        if c["allowed"] is not None:
            groups.setdefault(tuple(c["allowed"]), []).append(c)
    for allowed, cs in groups.items():
        blob, off = h.pack([c["text"] for c in cs])
        toks, toff = core.encode_batch_packed(blob, off, set(allowed))
        for i, c in enumerate(cs):
            got = toks[int(toff[i]):int(toff[i + 1])].tolist()
            assert got == c["tokens"], (c["name"], c["text"], allowed, got, c["tokens"])


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_pretokenizer_matches_oracle_split(cores, name):
    """Piece boundaries from tk_k_front == the sequential scanner == regex.findall (src/lib.rs:365)."""
    core, C = cores[name], h.c_oracle_for(name)
    blob, off = h.gen_corpus(0xABC0 + h.PATTERN_OF[name], h.PATTERN_OF[name] % 2, 2 << 20)
    starts = core.pretokenize_packed(blob, off)
    bb = blob.tobytes()
    ref = []
    for d in range(len(off) - 1):
        a, b = int(off[d]), int(off[d + 1])
        ref += [a] + [a + e for e in C.split(bb[a:b])[:-1]] if b > a else []
    ref.append(len(bb))
    assert starts.tolist() == ref


@pytest.mark.parametrize("name,mix,nbytes", [("gpt2_shaped", 2, 1 << 20), ("cl100k_shaped", 0, 8 << 20), ("o200k_shaped", 1, 8 << 20)])
def test_corpus_batch_equals_oracle(cores, name, mix, nbytes):
    core, C = cores[name], h.c_oracle_for(name)
    if name == "gpt2_shaped":  # config C1: one 1 MiB ASCII Lorem-ipsum document
        data = h.lorem(nbytes)
        blob, off = h.pack([data])
    else:
        blob, off = h.gen_corpus(0x5EED0000 + mix + 7, mix, nbytes)
    toks, toff = core.encode_batch_packed(blob, off)
    rt, ro = C.encode_batch(blob, off, None, 8)
    assert np.array_equal(toff, ro)
    assert np.array_equal(toks, rt)


def test_edge_cases(cores):
    core, C = cores["o200k_shaped"], h.c_oracle_for("o200k_shaped")
    # empty batch members, empty batch, ragged sizes (tests/test_encoding.py:81-83, 239-252)
    docs = [b"", b"hello world", b"", b"", "goodbye world 中文".encode(), b"x", b""]
    blob, off = h.pack(docs)
    toks, toff = core.encode_batch_packed(blob, off)
    for i, d in enumerate(docs):
        assert toks[int(toff[i]):int(toff[i + 1])].tolist() == C.encode_ordinary(d).tolist()
    toks, toff = core.encode_batch_packed(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert len(toks) == 0 and toff.tolist() == [0]
    assert core.encode_ordinary("") == []
    # documents must not interact: "'" + "s", " " + "x", digits across a boundary
    docs = [b"it'", b"s", b"a ", b" x", b"12", b"345", b"\n", b"\n", b"A", b"b"]
    blob, off = h.pack(docs)
    toks, toff = core.encode_batch_packed(blob, off)
    for i, d in enumerate(docs):
        assert toks[int(toff[i]):int(toff[i + 1])].tolist() == C.encode_ordinary(d).tolist(), d


@pytest.mark.parametrize("name", ["cl100k_shaped", "o200k_shaped", "gpt2_shaped"])
def test_catastrophically_repetitive(cores, name):
    """tests/test_encoding.py:113-124 at the reference's size (10_000 repeats)."""
    core, C = cores[name], h.c_oracle_for(name)
    docs = []
    for c in ["^", "0", "a", "'s", " ", "\n"]:
        big = c * 10_000
        docs += [big.encode(), (" " + big).encode(), (" " + big + "\n").encode()]
    blob, off = h.pack(docs)
    toks, toff = core.encode_batch_packed(blob, off)
    for i, d in enumerate(docs):
        got = toks[int(toff[i]):int(toff[i + 1])]
        assert np.array_equal(got, C.encode_ordinary(d)), d[:20]
        assert core.decode_bytes(got.tolist()) == d


def test_large_repeated(cores):
    """tests/test_encoding.py:52-57: 'x' * 1_000_000 on the o200k pattern is one giant piece."""
    core, C = cores["o200k_shaped"], h.c_oracle_for("o200k_shaped")
    data = b"x" * 1_000_000
    toks = core._encode_np(data, None)
    assert len(toks) > 0
    assert np.array_equal(toks, C.encode_ordinary(data))


def test_long_pieces(cores):
    """Pieces in every merge path: <=16 (lane), 17..64 (wave), >64 (tree)."""
    rng = np.random.default_rng(5)
    for name in h.ENCODING_NAMES:
        core, C = cores[name], h.c_oracle_for(name)
        for n in [2, 3, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 128, 129, 1000, 4097, 70000]:
            for alphabet in ("ab", "abcdefgh", "中文字", "etaoinshr"):
                s = "".join(rng.choice(list(alphabet), size=n))
                piece = s.encode()[: n if alphabet != "中文字" else 3 * (n // 3 + 1)]
                assert core.encode_single_piece(piece) == C.encode_piece(piece), (name, n, alphabet)


def test_vocab_free_vectors():
    """src/lib.rs:685-701: {ab:0, cd:1} -- byte_pair_split("abcd") = [ab, cd]; ("abab") = [ab, ab].
    The reference test uses a 2-entry map; a CoreBPE needs every single byte too, so they are appended
    after the two merges (which keeps ab/cd the lowest ranks)."""
    from tiktoken_amd import CoreBPE

    ranks = {b"ab": 0, b"cd": 1}
    for b in range(256):
        ranks[bytes([b])] = 2 + b
    core = CoreBPE(ranks, {}, h.PAT_STR[0])
    assert core.encode_single_piece(b"abcd") == [0, 1]
    assert core.encode_single_piece(b"abab") == [0, 0]
    assert core.encode_ordinary("abcd abab") == [0, 1, ranks[b" "], 0, 0]


def test_single_token_and_decode(cores):
    core = cores["cl100k_shaped"]
    ranks = h.load_vocab("cl100k_shaped")
    for tb, r in list(ranks.items())[::997]:
        assert core.encode_single_token(tb) == r
        assert core.decode_single_token_bytes(r) == tb
    assert core.encode_single_token(b"<|endoftext|>") == 100257
    with pytest.raises(KeyError):
        core.encode_single_token(b"\xff\xfe\xfd not a token")
    with pytest.raises(KeyError):
        core.decode_bytes([4_000_000_000])
    vals = core.token_byte_values()
    assert vals == sorted(ranks.keys())


@pytest.mark.parametrize("dbg", ["512", "256"])
def test_dedup_collision_and_disabled_paths(dbg, monkeypatch):
    """The in-call de-duplication of missed pieces must never change results: 512 truncates the table hash
    to 12 bits so that different pieces collide constantly (every duplicate candidate is byte-verified and
    real collisions are re-encoded on their own); 256 switches the table off."""
    from tiktoken_amd import CoreBPE

    monkeypatch.setenv("TIKTOKEN_AMD_DEBUG", dbg)
    g = h.load_golden("o200k_shaped")
    core = CoreBPE(h.golden_vocab("o200k_shaped"), g["special_tokens"], g["pat_str"])
    monkeypatch.delenv("TIKTOKEN_AMD_DEBUG")
    C = h.c_oracle_for("o200k_shaped")
    blob, off = h.gen_corpus(0xDED0, 0, 4 << 20)
    toks, toff = core.encode_batch_packed(blob, off)
    rt, ro = C.encode_batch(blob, off, None, 8)
    assert np.array_equal(toff, ro) and np.array_equal(toks, rt)


def test_text_at_any_device_address():
    """tk_encode_batch_device takes the text where the caller has it: a buffer that starts at an odd address (a slice of a tensor)
    gives the same tokens, also for the inputs that go through the workgroup-wide scanner (long runs)."""
    import torch
    from tiktoken_amd import CoreBPE

    g = h.load_golden("o200k_shaped")
    core = CoreBPE(h.golden_vocab("o200k_shaped"), g["special_tokens"], g["pat_str"])
    C = h.c_oracle_for("o200k_shaped")
    blob, off = h.gen_corpus(0xA11E, 1, 2 << 20)
    docs = [blob[int(off[i]):int(off[i + 1])].tobytes() for i in range(len(off) - 1)]
    docs[3:3] = [("x" * 20000).encode(), ("\u4e2d" * 9000).encode(), (" " * 15000 + "a").encode(), ("Ab" * 3000).encode(), ("\u00e9" * 7000 + "\n\n").encode()]
    blob = np.frombuffer(b"".join(docs), np.uint8)
    off = np.zeros(len(docs) + 1, np.uint64)
    off[1:] = np.cumsum([len(d) for d in docs])
    rt, ro = C.encode_batch(blob, off, None, 8)
    d_off = torch.from_numpy(off.view(np.int64)).cuda()
    for shift in (1, 2, 7, 13):
        host = np.zeros(len(blob) + shift + 64, np.uint8)
        host[shift:shift + len(blob)] = blob
        d = torch.from_numpy(host).cuda()
        dt, nt, do = core.encode_batch_device(d.data_ptr() + shift, len(blob), d_off.data_ptr(), off, len(docs))
        toks = h.dev_u32(dt, nt)
        toff = h.dev_u64(do, len(docs) + 1)
        assert np.array_equal(toff, ro) and np.array_equal(toks, rt), shift


def test_mid_size_documents_as_segments_in_one_launch(cores):
    """One document of 2 .. 128 KiB without special tokens is cut at piece starts that are certain whatever stands on either side (an ASCII
    letter followed by a space) into segments of at most 2 KiB, which go out as so many small calls in ONE launch (tk_api.hip, encode_mid);
    text without such cuts, and segments the small kernel does not do, take the general path.  Either way: the oracle's tokens."""
    for name in ("o200k_shaped", "cl100k_shaped", "gpt2_shaped"):
        core = cores[name]
        C = h.c_oracle_for(name)
        blob, off = h.gen_corpus(0x51D0 + len(name), 1, 1 << 20)
        text = blob[: int(off[-1])].tobytes()
        lorem = h.lorem(140000)
        before = core.stat("mid_calls")
        taken = 0
        for base, src in ((0, lorem), (0, text), (100_000, text)):
            for n in (2049, 2500, 4096, 5000, 10_000, 33_333, 65_536, 100_000, 131_072, 131_073):
                data = src[base: base + n].decode("utf-8", errors="ignore").encode()  # (a whole number of chars: the boundary is &str)
                got = core._encode_np(data, None)
                assert np.array_equal(got, C.encode_ordinary(data)), (name, base, n)
        taken = core.stat("mid_calls") - before
        assert taken >= 6, taken  # (the Lorem ipsum cases; the corpus is full of long pieces that are not tokens, which a segment leaves to the
        # general pipeline -- the whole document then: after such a call the next 16 .. 64 do not even try)
        # random documents made of vocabulary words, numbers, punctuation, newlines, contractions and short non-ASCII words: nothing the small
        # kernel leaves out, so the segments are what is tested -- every cut, every size from 2 to 128 KiB
        import random
        rng = random.Random(0x51D0 ^ len(name))
        vocab = [t for t in h.golden_vocab(name) if 2 <= len(t) <= 10 and t.isalpha() and t.isascii()]
        rng.shuffle(vocab)
        words = [w.decode() for w in vocab[:3000]] + ["don't", "I'll", "we've", "IT'S", "x'Re", "caf\u00e9", "na\u00efve", "\u4e2d\u6587", "\u043f\u0440\u0438\u0432\u0435\u0442", "3.14", "12345", "2024",
                                                       "a", "I", "e.g.", "(see", "note)", "\u2014", "...", "!?", "#tag", "@you", "x_y", "CamelCase", "ALLCAPS", "\U0001F600"]
        seps = [" "] * 12 + ["\n", "\n\n", "  ", ", ", ". ", "; ", ": ", "\t", " \n", "\r\n", " - "]
        before = core.stat("mid_calls")
        n_docs = 0
        for _ in range(120):
            target = rng.choice((2100, 3000, 4096, 6000, 9000, 15000, 30000, 65536, 100000, 131072))
            parts, size = [], 0
            while size < target:
                w = rng.choice(words) + rng.choice(seps)
                parts.append(w)
                size += len(w.encode())
            data = "".join(parts).encode()[:131072].decode("utf-8", errors="ignore").encode()
            assert np.array_equal(core._encode_np(data, None), C.encode_ordinary(data)), (name, len(data), data[:60])
            n_docs += 1
        assert core.stat("mid_calls") - before >= n_docs // 3, (core.stat("mid_calls") - before, n_docs)  # (up to 64 of them still skip after the corpus)
        # no space anywhere / no ASCII letter before a space / one word of 3 KB / spaces only: the general path, the same tokens
        for data in (("\u4e2d\u6587" * 2000).encode(), ("\u00e9t\u00e9 " * 900).encode(), b"x" * 3000, b" " * 5000, ("ab " * 20000).encode()[:70000],
                     ("word " * 300 + "y" * 3000 + " tail" * 300).encode()):
            assert np.array_equal(core._encode_np(data, None), C.encode_ordinary(data)), data[:20]


def test_alternating_result_buffers(cores):
    """tk_set_output_buffers(core, 2): the ids and offsets of a device-resident call stay where they are while the NEXT call runs (a consumer
    on another stream -- the gather of the several-process bench -- reads them without a copy); with one pair (the default) the next call
    writes over them."""
    import torch

    core = cores["o200k_shaped"]
    C = h.c_oracle_for("o200k_shaped")
    batches = []
    for seed in (0xB0F1, 0xB0F2, 0xB0F3):
        blob, off = h.gen_corpus(seed, 1, 3 << 20)
        host = np.zeros(len(blob) + 64, np.uint8)
        host[: len(blob)] = blob
        batches.append((torch.from_numpy(host).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), off, int(off[-1]), C.encode_batch(blob[: int(off[-1])], off, None, 8)))
    core.set_output_buffers(2)
    try:
        prev = None
        for d_text, d_off, off, n, (rt, ro) in batches:
            dt, nt, do = core.encode_batch_device(d_text.data_ptr(), n, d_off.data_ptr(), off, len(off) - 1)
            if prev is not None:  # the call before this one: still intact, and not the same buffers
                pdt, pnt, pdo, prt, pro, pnd = prev
                assert pdt != dt and pdo != do
                assert np.array_equal(h.dev_u32(pdt, pnt), prt) and np.array_equal(h.dev_u64(pdo, pnd + 1), pro)
            assert np.array_equal(h.dev_u32(dt, nt), rt) and np.array_equal(h.dev_u64(do, len(off)), ro)
            prev = (dt, nt, do, rt, ro, len(off) - 1)
    finally:
        core.set_output_buffers(1)
    a = core.encode_batch_device(batches[0][0].data_ptr(), batches[0][3], batches[0][1].data_ptr(), batches[0][2], len(batches[0][2]) - 1)
    b = core.encode_batch_device(batches[1][0].data_ptr(), batches[1][3], batches[1][1].data_ptr(), batches[1][2], len(batches[1][2]) - 1)
    assert a[0] == b[0]  # one pair again
    with pytest.raises(ValueError):
        core.set_output_buffers(3)


def test_multi_chunk_batches(monkeypatch):
    """Batches larger than the per-launch chunk are cut at document boundaries (tk_api.hip); force a tiny
    chunk so that a 3 MiB batch needs many launches, incl. documents larger than the chunk itself."""
    from tiktoken_amd import CoreBPE

    monkeypatch.setenv("TIKTOKEN_AMD_CHUNK_BYTES", "65536")
    g = h.load_golden("cl100k_shaped")
    core = CoreBPE(h.golden_vocab("cl100k_shaped"), g["special_tokens"], g["pat_str"])
    monkeypatch.delenv("TIKTOKEN_AMD_CHUNK_BYTES")
    C = h.c_oracle_for("cl100k_shaped")
    blob, off = h.gen_corpus(0xC4A1, 0, 3 << 20)
    toks, toff = core.encode_batch_packed(blob, off)
    rt, ro = C.encode_batch(blob, off, None, 8)
    assert np.array_equal(toff, ro) and np.array_equal(toks, rt)
    toks, toff = core.encode_batch_packed(blob, off, {"<|endoftext|>"})
    rt, ro = C.encode_batch(blob, off, {"<|endoftext|>"}, 8)
    assert np.array_equal(toff, ro) and np.array_equal(toks, rt)


def test_size_independent_properties_at_scale(cores):
    """At a size the oracle would need minutes for: decode(encode(x)) == x per document (sampled), token
    offsets are non-decreasing and end at the token count, and re-encoding a document alone gives the same
    tokens as inside the 256 MiB batch (documents never interact)."""
    core = cores["o200k_shaped"]
    blob, off = h.gen_corpus(0x5CA1E, 1, 256 << 20, threads=16)
    toks, toff = core.encode_batch_packed(blob, off)
    assert toff[0] == 0 and toff[-1] == len(toks) and np.all(np.diff(toff.astype(np.int64)) >= 0)
    bb = blob.tobytes()
    rng = np.random.default_rng(3)
    for d in rng.integers(0, len(off) - 1, size=200):
        a, b = int(off[d]), int(off[d + 1])
        t = toks[int(toff[d]):int(toff[d + 1])]
        assert core.decode_bytes(t.tolist()) == bb[a:b]
    for d in rng.integers(0, len(off) - 1, size=20):
        a, b = int(off[d]), int(off[d + 1])
        assert np.array_equal(core._encode_np(bb[a:b], None), toks[int(toff[d]):int(toff[d + 1])])
    total_bytes = sum(len(core.decode_single_token_bytes(int(t))) for t in toks[:200000])
    assert total_bytes == int(off[np.searchsorted(toff, 200000, side="right") - 1]) or total_bytes > 0


@pytest.mark.parametrize("cfg", ["C1", "C2", "C5", "C3", "C4r0", "C4r1", "N1"])
def test_baseline_configs_at_full_size(cfg):
    """Every BASELINE.json configuration at its FULL size (1 MiB / 64 MiB / 256 MiB with special tokens / 1 GiB), every token and
    every offset compared with the C oracle (all host threads) -- not a sample.  C4 (8 GiB doc-sharded over 8 GPUs) is represented by the
    shards of ranks 0 and 1 at their full size, generated with the seeds bench.py --gpus N gives those ranks: a shard is all a GPU sees.
    N1 (not a BASELINE configuration): 256 MiB of text with a natural share of pieces that are not tokens (helpers.natural_corpus)."""
    import os

    from tiktoken_amd import CoreBPE

    vocab, pat, specials, blob, off, allowed = h.baseline_config(cfg, threads=min(os.cpu_count() or 8, 32))
    core = CoreBPE(h.load_vocab(vocab), specials, h.PAT_STR[pat])
    C = h.c_oracle.COracle(pat, h.load_vocab(vocab), specials)
    toks, toff = core.encode_batch_packed(blob, off, allowed)
    rt, ro = C.encode_batch(blob, off, allowed, os.cpu_count() or 8)
    assert len(toks) == len(rt), (cfg, len(toks), len(rt))
    if not np.array_equal(toff, ro):
        d = int(np.flatnonzero(toff != ro)[0]) - 1
        a, b = int(off[d]), int(off[d + 1])
        raise AssertionError((cfg, "first differing document", d, blob[a:b].tobytes()[:200]))
    if not np.array_equal(toks, rt):
        i = int(np.flatnonzero(toks != rt)[0])
        d = int(np.searchsorted(ro, i, side="right")) - 1
        raise AssertionError((cfg, "first differing token", i, "document", d, blob[int(off[d]):int(off[d + 1])].tobytes()[:200]))
    if allowed:
        assert np.isin(toks, np.array(sorted(specials.values()), np.uint32)).sum() > 0  # (specials were really exercised)


# ---------------------------------------------------------------- long pieces (reference tests/test_encoding.py:52-57, CHANGELOG v0.13.0)
@pytest.mark.parametrize("unit", ["x", " ", "中"])
def test_megabyte_runs_are_fast(cores, unit):
    """A million repetitions of one character is ONE piece (or two): pre-tokenised by the workgroup-wide scanner, merged in rounds.
    Bit-exact, and under 20 ms per call -- the reference added _byte_pair_merge_large precisely so that such inputs are cheap."""
    import time

    core, C = cores["o200k_shaped"], h.c_oracle_for("o200k_shaped")
    data = (unit * 1_000_000).encode()
    want = C.encode_ordinary(data)
    assert np.array_equal(core._encode_np(data, None), want)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        got = core._encode_np(data, None)
        best = min(best, time.perf_counter() - t0)
    assert np.array_equal(got, want)
    assert best < 0.020, f"{unit!r} * 1e6 took {best * 1e3:.1f} ms"


@pytest.mark.parametrize("unit,name", [("Ab", "o200k_shaped"), ("x'll", "o200k_shaped"), ("x'll", "cl100k_shaped"), ("a'S b'Ll", "gpt2_shaped"),
                                        ("x'll\u4e2d'd", "o200k_shaped"), ("1a", "cl100k_shaped")])
def test_chains_of_uncertain_boundaries_do_not_take_seconds(cores, unit, name):
    """A megabyte of short pieces without a single certain piece start ("camelCase" chains; contraction chains, where whether 'll ends a
    piece depends on unbounded left context).  The first kind has certain starts by a rule with context (lower -> upper without an
    apostrophe near), the second is walked a 4 KiB window at a time (tk_coop_window_walk): 4 ms and 20 ms here; round 1: 6.5 s and 3 s."""
    import time

    core, C = cores[name], h.c_oracle_for(name)
    data = (unit * (1_000_000 // len(unit))).encode()
    want = C.encode_ordinary(data)
    assert np.array_equal(core._encode_np(data, None), want)
    t0 = time.perf_counter()
    got = core._encode_np(data, None)
    dt = time.perf_counter() - t0
    assert np.array_equal(got, want)
    assert dt < 0.15, f"{unit!r}: {dt * 1e3:.1f} ms"


def test_ten_megabytes_without_a_certain_start_take_the_generic_way():
    """A stretch of short pieces without certain starts used to be quadratic (every deferred tile walked from its start: 23 ms for 1 MB,
    seconds for 10 MB).  Now a tile gives up after TKF_WALK_BUDGET windows, the generic engine splits the chunk under the same pat_str
    (linear on such text), and the tiles that gave up run again with every piece start a hard start (tk_api.hip, stage_deferred)."""
    import time

    import tiktoken_amd

    name = "o200k_shaped"
    core, C = tiktoken_amd.get_encoding(name)._core_bpe, h.c_oracle_for(name)
    data = ("x'll" * 2_500_000).encode()
    want = C.encode_ordinary(data)
    before = core.stat("fallbacks")
    assert np.array_equal(core._encode_np(data, None), want)
    assert core.stat("fallbacks") == before + 1
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        got = core._encode_np(data, None)
        ts.append(time.perf_counter() - t0)
    assert np.array_equal(got, want)
    print(f"10 MB of x'll: {min(ts) * 1e3:.1f} ms")
    assert min(ts) < 0.06, ts  # (measured 45 ms: profiles/r03_long_runs.txt has the 1 MB cases)
    # documents around it, special tokens inside it, and a second stretch in a later document
    docs = [b"plain text before. ", ("x'll" * 300_000).encode(), b"", ("y'd" * 200_000 + "<|endoftext|>" + "z'Re" * 100_000).encode(), b"and after"]
    blob, off = h.pack(docs)
    for allowed in (None, "all"):
        toks, toff = core.encode_batch_packed(blob, off, allowed)
        rt, ro = C.encode_batch(blob, off, None if allowed is None else set(h.load_golden(name)["special_tokens"]), 4)
        assert np.array_equal(toff, ro) and np.array_equal(toks, rt), allowed


def test_sparse_non_ascii_chars_by_dense_lanes(cores):
    """Round 6, tk_fused.h phase B (TKF_DENSE_DECODE): a wavefront whose 1024 bytes hold few non-ASCII chars lists their lead bytes and decodes them by
    dense lanes -- the class goes into the class planes by LDS atomics at the char's position, whichever lane holds its bytes -- while a wavefront with more
    than TKF_DENSE_MAX of them keeps the loop per lane.  Chars of two, three and four bytes (letters with and without case, digits, marks, white space,
    punctuation, astral letters and emoji) at EVERY offset against the lanes' sixteen bytes, the wavefronts' 1024 and the planes' 32-bit words, sparse
    (dense lanes), dense (the loop) and changing between the two inside a tile; every token compared with the oracle."""
    chars = ["\u00e9", "\u0416", "\u00a0", "\u0663", "\u0301", "\u2019", "\u4e2d", "\u3000", "\u0e01", "\u2028", "\U00010400", "\U0001f600", "\U0001d7ce", "\u01c5", "\u017f"]
    for name in h.ENCODING_NAMES:
        core, C = cores[name], h.c_oracle_for(name)
        docs = []
        for shift in range(0, 70):
            parts, k = ["x" * shift], shift
            for i in range(260):  # ~ 11 KiB: three tiles, every char at a new offset against 16, 32 and 1024
                c = chars[(i + shift) % len(chars)]
                gap = 5 + (i * 7 + shift) % 23
                k += 1
                filler = (" word" * 8)[: gap] if i % 3 else "a" * gap
                parts.append(filler + c + (chars[(i * 5 + 1) % len(chars)] if i % 4 == 0 else "") + ("'s" if i % 9 == 0 else ""))
            docs.append("".join(parts).encode())
            if shift % 7 == 0:  # sparse text that turns dense and back: the two paths side by side in one tile
                dense = "".join(chars[(j + shift) % len(chars)] for j in range(700))
                docs.append(("".join(parts[:90]) + dense + " " + "".join(parts[90:200]) + dense[:333] + "".join(parts[200:])).encode())
        blob, off = h.pack(docs)
        toks, toff = core.encode_batch_packed(blob, off, None)
        rt, ro = C.encode_batch(blob, off, None, 8)
        assert np.array_equal(toff, ro) and np.array_equal(toks, rt), name
        one = b"\n".join(docs[::5])  # one document: the chars at other offsets again, no hard start between the parts
        assert np.array_equal(core._encode_np(one, None), C.encode_ordinary(one)), name


def test_letter_runs_around_tile_ends(cores):
    """Round 6, tk_fused.h: a tile whose left context holds no certain start begins its scan at a position where the matcher's state is known (a letter
    behind a letter, TKF_SYNC_POINTS), and a letter run that leaves a tile's window is read on to its end (TKF_EXTEND) -- unless what ends the run lets
    the piece go on (o200k: a cased letter, an apostrophe), more than 2 KiB follow, or the run is not of the kind the rule is about: then the tile goes to
    the workgroup-wide scanner as before.  Runs of every such kind and ending, at every offset against the tiles' ends (3840 bytes), documents and special
    tokens inside them; every token compared with the oracle."""
    runs = ["\u5b57", "\u0e01\u0e31", "\u3042\u30a2\u4e9c", "x", "X", "xY", "\u00e9", "\u0416", "\u5b57x", "x\u0301", "a\u5b57", "\u5b57A"]
    ends = ["", " ", "A", "Ab", "a", "'s", "'ll ", "'S", "\u2019s", "1", "12345", "\n", "\r\n\n", "\u3002", "\u0301", ".", "/", "'", "''", " '", "\t", "\u00a0", "\u017f", "'\u017f"]
    for name in h.ENCODING_NAMES:
        core, C = cores[name], h.c_oracle_for(name)
        specials = h.load_golden(name)["special_tokens"]
        docs, k = [], 0
        for unit in runs:
            for reps in (45, 130, 400, 900, 3000):
                for end in ends:
                    k += 1
                    pad = "the quick brown fox " * (k % 7) + "a" * (k % 13) + " "
                    body = unit * (reps // max(1, len(unit.encode()) // 3 or 1)) + end + unit * (k % 5) + " tail"
                    if k % 11 == 0:
                        body = body[: len(body) // 2] + "<|endoftext|>" + body[len(body) // 2:]
                    docs.append((pad + body).encode())
                    if k % 17 == 0:  # the same run split by a document boundary: the run ends with the document
                        docs.append((unit * reps).encode())
        blob, off = h.pack(docs)
        assert len(blob) > (1 << 20)
        for allowed in (None, "all"):
            toks, toff = core.encode_batch_packed(blob, off, allowed)
            rt, ro = C.encode_batch(blob, off, None if allowed is None else set(specials), 8)
            assert np.array_equal(toff, ro) and np.array_equal(toks, rt), (name, allowed)
        # one document: the runs follow each other without a hard start between them
        one = b" ".join(docs[::3])
        assert np.array_equal(core._encode_np(one, None), C.encode_ordinary(one)), name


def test_a_tile_that_gives_up_while_the_host_is_not_waiting_repeats_the_batch():
    """Round 6: the host no longer waits for the deferred tiles' counters between the two kernels (tk_api.hip, stage_deferred).  A core's first
    batch with a stretch that makes a tile give up runs to its end on what the kernels had (the tiles that gave up: empty), is found out by
    chunk_finish and repeated once, waiting; from then on the core waits in every chunk.  Same tokens as the oracle both times."""
    from tiktoken_amd import CoreBPE

    name = "o200k_shaped"
    g = h.load_golden(name)
    core, C = CoreBPE(h.golden_vocab(name), g["special_tokens"], h.PAT_STR[h.ENCODING_NAMES.index(name)]), h.c_oracle_for(name)
    plain = ("The quick brown fox, 12345 times. " * 20_000).encode()
    assert np.array_equal(core._encode_np(plain, None), C.encode_ordinary(plain))
    assert core.stat("resynced") == 0 and core.stat("fallbacks") == 0
    docs = [b"before ", ("x'll" * 300_000).encode(), plain[:100_000], b""]
    blob, off = h.pack(docs)
    rt, ro = C.encode_batch(blob, off, None, 4)
    for k in range(2):
        toks, toff = core.encode_batch_packed(blob, off, None)
        assert np.array_equal(toff, ro) and np.array_equal(toks, rt), k
        assert core.stat("resynced") == 1 and core.stat("fallbacks") == k + 1, (k, core.stat("resynced"), core.stat("fallbacks"))
    assert np.array_equal(core._encode_np(plain, None), C.encode_ordinary(plain))


def test_a_late_chunk_that_gives_up_repeats_the_whole_batch(monkeypatch):
    """The same with a batch of several chunks (1 MiB each): the stretch lies in a late chunk, the chunks before it have been finished and their ids sent
    to the host when chunk_finish finds the tile that gave up; the pass is abandoned and run once more from the first chunk, waiting."""
    from tiktoken_amd import CoreBPE

    monkeypatch.setenv("TIKTOKEN_AMD_CHUNK_BYTES", str(1 << 20))
    name = "o200k_shaped"
    g = h.load_golden(name)
    core, C = CoreBPE(h.golden_vocab(name), g["special_tokens"], h.PAT_STR[h.ENCODING_NAMES.index(name)]), h.c_oracle_for(name)
    monkeypatch.delenv("TIKTOKEN_AMD_CHUNK_BYTES")
    blob, off = h.gen_corpus(0xC0FFEE, 1, 6 << 20)
    docs = [blob[int(off[i]):int(off[i + 1])].tobytes() for i in range(len(off) - 1)]
    docs.insert(len(docs) * 3 // 4, ("x'll" * 200_000).encode())
    blob, off = h.pack(docs)
    rt, ro = C.encode_batch(blob, off, None, 8)
    for k in range(2):
        toks, toff = core.encode_batch_packed(blob, off, None)
        assert np.array_equal(toff, ro) and np.array_equal(toks, rt), k
        assert core.stat("resynced") == 1 and core.stat("chunks") >= 6, (k, core.stat("resynced"), core.stat("chunks"))


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_ordinary_text_through_the_give_up_path(monkeypatch, name):
    """TIKTOKEN_AMD_DEBUG bit 0x20000000: a walk budget of zero windows -- every deferred tile that would walk a window gives up, and the
    chunk goes the generic way of the test above.  Same tokens as the oracle on corpus text with special tokens."""
    from tiktoken_amd import CoreBPE

    monkeypatch.setenv("TIKTOKEN_AMD_DEBUG", str(0x20000000))
    g = h.load_golden(name)
    core, C = CoreBPE(h.golden_vocab(name), g["special_tokens"], h.PAT_STR[h.ENCODING_NAMES.index(name)]), h.c_oracle_for(name)
    blob, off = h.gen_corpus(0xFA11 + h.ENCODING_NAMES.index(name), 1, 8 << 20)
    docs = [blob[int(off[i]):int(off[i + 1])].tobytes() for i in range(len(off) - 1)]
    docs[3] = docs[3] + ("x'll" * 100_000).encode() + b"<|endoftext|>" + ("Ab" * 5000 + "q're" * 60_000).encode()
    blob, off = h.pack(docs)
    for allowed in (None, "all"):
        toks, toff = core.encode_batch_packed(blob, off, allowed)
        rt, ro = C.encode_batch(blob, off, None if allowed is None else set(g["special_tokens"]), 8)
        assert np.array_equal(toff, ro) and np.array_equal(toks, rt), allowed
    # (cl100k: since round 6 a tile of this text never walks -- a letter behind a letter is a position the scan can start from, tk_fused.h TKF_SYNC_POINTS,
    # and every other class pair of "x'll" / "q're" is a certain start under that pattern: the text is encoded under the zero budget without the way out)
    assert core.stat("fallbacks") >= (0 if name == "cl100k_shaped" else 2)


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_long_runs_of_every_kind(cores, name):
    """Documents made of long runs (2..40 KiB) of letters, digits, white space with and without newlines, punctuation, CJK, accented
    letters, alternating case, contractions -- every path of the scanner over run queries, the digit-group jumps, the far-away
    piece starts and the merges in rounds -- against the oracle, every token."""
    core, C = cores[name], h.c_oracle_for(name)
    rng = np.random.default_rng(0xA11)
    units = ["x", "Q", "é", "中", "ก", "1", "٣", " ", "\t", "\n", " \n", "\r\n", "!", "/", "!/\n", "'", "x'll", "Ab", "aB1", "　", "é", "a ", " a", "0 ", "--"]
    docs = []
    for _ in range(60):
        parts = []
        for _ in range(int(rng.integers(1, 5))):
            u = units[int(rng.integers(0, len(units)))]
            parts.append(u * int(rng.integers(2_000, 40_000) // len(u.encode())))
            if rng.random() < 0.5:
                parts.append("".join(rng.choice(h.ADV, size=int(rng.integers(0, 6)))))
        docs.append("".join(parts).encode())
    blob, off = h.pack(docs)
    toks, toff = core.encode_batch_packed(blob, off)
    rt, ro = C.encode_batch(blob, off, None, 8)
    assert np.array_equal(toff, ro), int(np.flatnonzero(toff != ro)[0]) - 1
    assert np.array_equal(toks, rt)
    starts = core.pretokenize_packed(blob, off)
    bb = blob.tobytes()
    ref = []
    for d in range(len(off) - 1):
        a, b = int(off[d]), int(off[d + 1])
        ref += [a] + [a + e for e in C.split(bb[a:b])[:-1]] if b > a else []
    ref.append(len(bb))
    assert starts.tolist() == ref


def test_rounds_and_one_at_a_time_merges_agree(monkeypatch):
    """Long pieces go through tk_k_merge_rounds (all pairs of the lowest rank per round, its assumption checked round by round);
    debug bit 1024 forces tk_k_merge_long (the reference's order literally) for the same pieces."""
    from tiktoken_amd import CoreBPE

    g = h.load_golden("cl100k_shaped")
    rng = np.random.default_rng(5)
    pieces = [("ab" * 3000).encode(), bytes(rng.integers(97, 123, size=5000, dtype=np.uint8)), ("the quick brown fox " * 400).replace(" ", "").encode(),
              ("中文字" * 1500).encode(), bytes(rng.integers(0, 256, size=3000, dtype=np.uint8))]
    a = CoreBPE(h.golden_vocab("cl100k_shaped"), g["special_tokens"], g["pat_str"])
    monkeypatch.setenv("TIKTOKEN_AMD_DEBUG", "1024")
    b = CoreBPE(h.golden_vocab("cl100k_shaped"), g["special_tokens"], g["pat_str"])
    monkeypatch.delenv("TIKTOKEN_AMD_DEBUG")
    C = h.c_oracle_for("cl100k_shaped")
    for p in pieces:
        want = C.encode_piece(p)
        assert a.encode_single_piece(p) == want
        assert b.encode_single_piece(p) == want
    # pieces of 128 KiB and more: all the workgroups of tk_k_merge_rounds_wide on one piece (several such pieces in one call, with
    # short ones between them; lowercase letters only, so that every document is one piece of the pattern)
    big = [bytes(rng.integers(97, 123, size=300_000, dtype=np.uint8)), ("ab" * 100_000).encode(), ("the quick brown fox " * 9000).replace(" ", "").encode(),
           bytes(rng.integers(97, 101, size=150_001, dtype=np.uint8)), b"hello", ("abc" * 50_000).encode()]
    blob = np.frombuffer(b"".join(big), np.uint8)
    off = np.zeros(len(big) + 1, np.uint64)
    off[1:] = np.cumsum([len(x) for x in big])
    toks, toff = a.encode_batch_packed(blob, off)
    rt, ro = C.encode_batch(blob, off, None, 8)
    assert np.array_equal(toff, ro) and np.array_equal(toks, rt)


# ---------------------------------------------------------------- decode on the device (src/lib.rs:345-358)
def test_decode_batch_on_device(cores):
    core = cores["o200k_shaped"]
    blob, off = h.gen_corpus(0xDEC0DE, 1, 4 << 20)
    toks, toff = core.encode_batch_packed(blob, off)
    data, boff = core.decode_batch_packed(toks, toff)
    assert data == blob.tobytes() and np.array_equal(boff, off)
    # one long document through tk_decode_bytes (device path), a short one on the host, special tokens, an unknown id
    assert core.decode_bytes(toks[: int(toff[40])].tolist()) == blob[: int(off[40])].tobytes()
    assert core.decode_bytes(toks[:7].tolist()) == b"".join(core.decode_single_token_bytes(int(t)) for t in toks[:7])
    sp = np.array([199999, 200018] * 5000, np.uint32)
    assert core.decode_batch_packed(sp, np.array([0, len(sp)], np.uint64))[0] == b"<|endoftext|><|endofprompt|>" * 5000
    bad = toks[:20000].copy()
    bad[12345] = 199_998_0
    with pytest.raises(KeyError, match="Invalid token for decoding: 1999980"):
        core.decode_batch_packed(bad, np.array([0, len(bad)], np.uint64))
    data, boff = core.decode_batch_packed(np.zeros(0, np.uint32), np.zeros(3, np.uint64))
    assert data == b"" and boff.tolist() == [0, 0, 0]


def test_decode_in_ranges_and_on_the_device(cores):
    """A batch of more than 16 Mi ids is decoded in ranges whose copies overlap (tk_decode_batch: ids in through page-locked staging, or
    straight from a page-locked caller's buffer; bytes out while the next range is decoded); tk_decode_batch_device leaves ids and bytes
    in HBM.  All of them: the text, byte for byte (src/lib.rs:345-358)."""
    import torch

    core = cores["o200k_shaped"]
    blob, off = h.gen_corpus(0xDEC0DF, 1, 96 << 20)
    n = int(off[-1])
    toks, toff = core.encode_batch_packed(blob, off)  # (views of page-locked memory)
    assert len(toks) > (16 << 20) + 1000
    for src in (toks, np.array(toks)):  # the caller's buffer page-locked, then pageable
        data, boff = core.decode_batch_packed(src, toff, as_array=True)
        assert len(data) == n and np.array_equal(data, blob[:n]) and np.array_equal(boff, off)
        del data
    # an unknown id in the second range: the reference's KeyError with that id
    bad = np.array(toks)
    bad[(16 << 20) + 777] = 1999980
    with pytest.raises(KeyError, match="Invalid token for decoding: 1999980"):
        core.decode_batch_packed(bad, toff)
    del bad
    d_tok = torch.from_numpy(np.array(toks).view(np.int32)).cuda()
    d_off = torch.from_numpy(np.array(toff).view(np.int64)).cuda()
    torch.cuda.synchronize()
    db, nb, do = core.decode_batch_device(d_tok.data_ptr(), len(toks), d_off.data_ptr(), len(toff) - 1)
    assert nb == n and do
    got = torch.as_tensor(h._DevArray(db, nb, "|u1"), device="cuda").cpu().numpy()
    assert np.array_equal(got, blob[:n])
    assert np.array_equal(h.dev_u64(do, len(toff)), off)
    db, nb, do = core.decode_batch_device(d_tok.data_ptr(), 1000, 0, 0)  # no offsets: the bytes only
    assert do == 0 and bytes(torch.as_tensor(h._DevArray(db, nb, "|u1"), device="cuda").cpu().numpy()) == core.decode_bytes(toks[:1000].tolist())
    d_tok[5] = 1999980
    torch.cuda.synchronize()
    with pytest.raises(KeyError, match="Invalid token for decoding: 1999980"):
        core.decode_batch_device(d_tok.data_ptr(), 1000, 0, 0)


# ---------------------------------------------------------------- several devices in one process (virtual ranks on one GPU here)
def test_multi_device_group_equals_single_device():
    """CoreBPE(devices=[...]): documents split into contiguous ranges of about equal bytes, one replica per device (tk_group_encode_batch);
    on the one-GPU test box two and three replicas share device 0.  Same tokens and offsets as the oracle, ordinary and with specials;
    the variant that gathers the ids on the first device (peer copies) returns the same."""
    import torch

    from bench import DevArray
    from tiktoken_amd import CoreBPE

    g = h.load_golden("cl100k_shaped")
    C = h.c_oracle_for("cl100k_shaped")
    blob, off = h.gen_corpus(0x6A7, 0, 6 << 20)
    rt, ro = C.encode_batch(blob, off, None, 8)
    for devices in ([0, 0], [0, 0, 0]):
        core = CoreBPE(h.golden_vocab("cl100k_shaped"), g["special_tokens"], g["pat_str"], devices=devices)
        toks, toff = core.encode_batch_packed(blob, off)
        assert np.array_equal(toff, ro) and np.array_equal(toks, rt)
        dt, nt, do = core.encode_batch_gathered(blob, off)
        assert nt == len(rt)
        assert np.array_equal(torch.as_tensor(DevArray(dt, nt, "<i4"), device="cuda").cpu().numpy().view(np.uint32), rt)
        assert np.array_equal(torch.as_tensor(DevArray(do, len(off), "<i8"), device="cuda").cpu().numpy().astype(np.uint64), ro)
    st, so = C.encode_batch(blob, off, {"<|endoftext|>"}, 8)
    toks, toff = core.encode_batch_packed(blob, off, {"<|endoftext|>"})
    assert np.array_equal(toff, so) and np.array_equal(toks, st)
    # ragged: fewer documents than replicas, empty documents, an empty batch -- through the host gather and through the device gather
    # (whose shards never take the one-launch small path: that one writes to host memory only)
    for docs in ([b"one"], [b"a", b"b"], [b"", b""], [], [b"a", b"", b"bb " * 1000, b""], [b"x" * 5000, b"", b"", b"tail"]):
        b2, o2 = h.pack(docs)
        t2, f2 = core.encode_batch_packed(b2, o2)
        w2, x2 = C.encode_batch(b2, o2, None, 1) if docs else (np.zeros(0, np.uint32), np.zeros(1, np.uint64))
        assert np.array_equal(t2, w2) and np.array_equal(f2, x2), docs
        dt, nt, do = core.encode_batch_gathered(b2, o2)
        assert nt == len(w2), docs
        if nt:
            assert np.array_equal(torch.as_tensor(DevArray(dt, nt, "<i4"), device="cuda").cpu().numpy().view(np.uint32), w2), docs
        assert np.array_equal(torch.as_tensor(DevArray(do, len(o2), "<i8"), device="cuda").cpu().numpy().astype(np.uint64), x2), docs
    assert core.group_stat("gathers_peer") > 0 and core.group_stat("gathers_rccl") == 0  # (virtual ranks share a device: no RCCL here)


def test_distinct_devices_gather_over_rccl():
    """The branch no one-GPU box can take: CoreBPE(devices=[0, 1, ...]) on pairwise distinct devices gathers the token ids with ONE grouped
    ncclSend / ncclRecv over xGMI (tk_group_encode_batch_device).  Skipped below two devices; the first box that has them runs it."""
    import torch

    from bench import DevArray
    from tiktoken_amd import CoreBPE, _lib

    n = min(_lib.device_count(), 8)
    if n < 2:
        pytest.skip(f"{n} HIP device(s) visible: the RCCL gather needs at least two distinct ones")
    g = h.load_golden("o200k_shaped")
    C = h.c_oracle_for("o200k_shaped")
    blob, off = h.gen_corpus(0x5EED0003, 1, 64 << 20)
    rt, ro = C.encode_batch(blob, off, None, 8)
    core = CoreBPE(h.golden_vocab("o200k_shaped"), g["special_tokens"], g["pat_str"], devices=list(range(n)))
    for rep in range(2):  # (the second call re-uses communicators and buffers)
        dt, nt, do = core.encode_batch_gathered(blob, off)
        assert nt == len(rt)
        with torch.cuda.device(0):
            assert np.array_equal(torch.as_tensor(DevArray(dt, nt, "<i4"), device="cuda:0").cpu().numpy().view(np.uint32), rt)
            assert np.array_equal(torch.as_tensor(DevArray(do, len(off), "<i8"), device="cuda:0").cpu().numpy().astype(np.uint64), ro)
    assert core.group_stat("gathers_rccl") >= 2 and core.group_stat("gathers_peer") == 0
    toks, toff = core.encode_batch_packed(blob, off, "all")  # the host-gather form on the same replicas
    st, so = C.encode_batch(blob, off, "all", 8)
    assert np.array_equal(toff, so) and np.array_equal(toks, st)


# ---------------------------------------------------------------- small calls: one launch (tk_k_small)
def test_small_calls_one_launch_same_tokens(monkeypatch):
    """A single document of up to 2 KiB without special tokens is encoded by ONE workgroup in ONE launch (tk_k_small); the result is
    the oracle's, and the general pipeline's (debug bit 2048 switches the short cut off).  Pieces that are not tokens are merged in the
    kernel: one lane each up to 24 bytes, sixteen lanes each up to 256 bytes; longer ones send the call to the general path."""
    from tiktoken_amd import CoreBPE

    rng = np.random.default_rng(11)
    alphabet = list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789   \n\t.,;:!?'\"()-_/\u00e9\u00fc\u00df\u4e2d\u6587\u0416\u0434\U0001F600\u200b\u3000\r")
    texts = ["hello world", "a", " ", "\n", "Hello, World! It's 12345 o'clock.\n\n  def f(x):\n\treturn x's", "x" * 2048, " " * 2048, "ab" * 1000,
             "\u4e2d" * 682, "don't DON'T I'LL we've", "supercalifragilisticexpialidociousantidisestablishmentarianism" * 3, "1234567890" * 20]
    for _ in range(300):
        k = int(rng.integers(1, 400))
        texts.append("".join(alphabet[int(j)] for j in rng.integers(0, len(alphabet), size=k)))
    texts = [t for t in texts if len(t.encode()) <= 2048]
    for name in ("gpt2_shaped", "cl100k_shaped", "o200k_shaped"):
        g = h.load_golden(name)
        a = CoreBPE(h.golden_vocab(name), g["special_tokens"], g["pat_str"])
        monkeypatch.setenv("TIKTOKEN_AMD_DEBUG", "2048")
        b = CoreBPE(h.golden_vocab(name), g["special_tokens"], g["pat_str"])
        monkeypatch.delenv("TIKTOKEN_AMD_DEBUG")
        C = h.c_oracle_for(name)
        a.set_profiling(True)
        a.reset_kernel_ms()
        for t in texts:
            want = C.encode_ordinary(t.encode()).tolist()
            assert a.encode_ordinary(t) == want, (name, t)
            assert b.encode_ordinary(t) == want, (name, t)
        assert a.kernel_ms("tk_k_small")[1] >= len(texts)
        assert a.kernel_ms("tk_k_front")[1] <= 8  # (only "x" * 2048, "ab" * 1000, the Chinese run and " " * 2048 where they are not tokens)
        a.set_profiling(False)


def test_hello_world_latency():
    """The reference's commonest call, end to end through the Python layer (Encoding.encode: special-token check, str -> UTF-8, C ABI,
    one launch, list of ints).  Measured 27-28 us on MI355X (tools/small_call.py, profiles/r02_small_calls.txt); the bounds leave a little
    room for a busy host."""
    import time

    import tiktoken_amd

    enc = tiktoken_amd.get_encoding("o200k_shaped")
    for _ in range(300):
        enc.encode("hello world")
    ts = []
    for _ in range(3000):
        t0 = time.perf_counter_ns()
        enc.encode("hello world")
        ts.append(time.perf_counter_ns() - t0)
    ts.sort()
    p10, med = ts[len(ts) // 10] / 1e3, ts[len(ts) // 2] / 1e3
    print(f"Encoding.encode('hello world'): p10 {p10:.1f} us, median {med:.1f} us")
    assert p10 <= 33.0 and med <= 48.0, (p10, med)


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_fuzzed_awkward_documents(cores, name):
    """Random batches of awkward documents (helpers.fuzz_batch; tools/gpu_fuzz.py runs many more seeds), with and without special tokens,
    every token against the oracle."""
    import zlib

    core, C = cores[name], h.c_oracle_for(name)
    for seed in (1, 2):
        blob, off = h.pack(h.fuzz_batch(zlib.crc32(name.encode()) + seed, 6 << 20))
        for allowed in (None, "all"):
            toks, toff = core.encode_batch_packed(blob, off, allowed)
            rt, ro = C.encode_batch(blob, off, allowed, 8)
            assert np.array_equal(toff, ro), (seed, allowed)
            assert np.array_equal(toks, rt), (seed, allowed)


@pytest.mark.parametrize("name,word", [("o200k_shaped", " a's"), ("cl100k_shaped", " a's"), ("gpt2_shaped", " a's"), ("o200k_shaped", " I'll")])
def test_full_continuation_list_and_a_piece_that_leaves_the_window(cores, name, word):
    """A tile with more scan chains that go on after an uncertain boundary (" a's": whether 's ends the piece is not known from the class
    pair) than the continuation list holds, in which a white-space piece needs text beyond the window ("\\n" + 6000 spaces: `\\s*[\\r\\n]+`
    ends at the last newline of the run, which only the workgroup-wide scanner can know) and ends inside the tile again.  Found by
    tools/gpu_fuzz.py: the chain's re-entry had no room on the list ("scanner list overflow"); now the workgroup walks on itself."""
    core, C = cores[name], h.c_oracle_for(name)
    docs = []
    for k in range(0, 1100, 37):
        docs.append((word * k + "\n" + " " * 6000 + "x" + word * 900 + " \n" + "\t" * 5000 + "\n!").encode())
    blob, off = h.pack(docs)
    toks, toff = core.encode_batch_packed(blob, off)
    rt, ro = C.encode_batch(blob, off, None, 8)
    assert np.array_equal(toff, ro)
    assert np.array_equal(toks, rt)
