"""The product's device functions (tiktoken_amd/csrc/tk_device.h: class bytes, certain-start rule,
tk_piece_end scanner, table probes, per-lane merge) compiled for the HOST and run in a sequential loop
that mirrors tk_k_front (scanners, tile rule, whole-piece probe) and the lane merge, against the oracle.  This is how logic errors are caught in a
container without a GPU; the real kernels are checked by the `-m gpu` tests."""
import itertools
import random

import numpy as np
import pytest

import helpers as h


@pytest.fixture(scope="module")
def sims():
    return {n: h.HostSim(h.PAT_STR[h.PATTERN_OF[n]], h.load_vocab(n), h.SPECIALS[n]) for n in h.ENCODING_NAMES}


def _ref_ends(C, docs, off):
    ref = []
    for d, dd in enumerate(docs):
        ref += [int(off[d]) + e for e in C.split(dd)]
    return ref


@pytest.mark.parametrize("bits", [False, True], ids=["bytewalk", "bitparallel"])
@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_pretok_sim_on_adversarial_batches(sims, name, bits):
    sim, C = sims[name], h.c_oracle_for(name)
    rng = random.Random(11)
    for _ in range(3000):
        docs = ["".join(rng.choice(h.ADV) for _ in range(rng.randint(0, 30))).encode() for _ in range(rng.randint(1, 4))]
        blob, off = h.pack(docs)
        ends, _ = sim.piece_ends(blob, off, bits=bits)
        assert ends.tolist() == _ref_ends(C, docs, off), docs


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_bitparallel_long_runs_and_window_edges(sims, name):
    """Runs longer than one 64-bit window (continued through the extension windows) and runs that reach the
    edge of a tile's LDS window (4096 + 192 bytes: unresolved -> byte-walking fallback), at every alignment."""
    sim, C = sims[name], h.c_oracle_for(name)
    rng = random.Random(29)
    units = ["a", "A", "1", " ", "\n", "!", "中", "́", "'s", "/", "\t", "ก", "é", "-", "="]
    for _ in range(400):
        parts = []
        size = 0
        while size < 9000:
            u = rng.choice(units)
            k = rng.choice([1, 2, 3, 30, 59, 64, 65, 130, 200, 700])
            seg = rng.choice(["", " ", "x", "'", "X"]) + u * k + rng.choice(["", " ", "b", "B", "'ll", "\n", "9", "'S"])
            parts.append(seg)
            size += len(seg.encode())
        doc = "".join(parts).encode()
        pad = b"y" * rng.randint(0, 70)  # shift everything relative to the 4096-byte tiles
        blob, off = h.pack([pad + doc])
        ends, n_fallback = sim.piece_ends(blob, off, bits=True)
        assert ends.tolist() == C.split(pad + doc)


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_tile_rule_of_the_front_kernel(sims, name):
    """tk_k_front's tile rule (tk_fused.h): every tile derives exactly the piece starts inside its own byte range
    from its certain starts plus the last certain start in a bounded left context (else a walk back).  Simulated
    with tiny tiles (many tile boundaries inside pieces) and with the real 4096 / 64 geometry."""
    sim, C = sims[name], h.c_oracle_for(name)
    rng = random.Random(41)
    for _ in range(600):
        docs = ["".join(rng.choice(h.ADV) for _ in range(rng.randint(0, 60))).encode() for _ in range(rng.randint(1, 4))]
        blob, off = h.pack(docs)
        ref = _ref_ends(C, docs, off)
        for tile, left in ((16, 8), (32, 32), (64, 16)):
            ends, _ = sim.piece_ends_tiled(blob, off, tile=tile, left=left)
            assert ends.tolist() == ref, (docs, tile, left)
    units = ["a", "A", "1", " ", "\n", "!", "中", "́", "'s", "/", "\t", "é"]
    for _ in range(40):  # long runs: pieces that span several tiles, left context without any certain start
        doc = "".join(rng.choice(["", " ", "x"]) + rng.choice(units) * rng.choice([1, 3, 70, 300, 5000]) for _ in range(30)).encode()
        blob, off = h.pack([doc])
        for tile, left in ((64, 16), (4096, 64)):
            ends, _ = sim.piece_ends_tiled(blob, off, tile=tile, left=left)
            assert ends.tolist() == C.split(doc), (tile, left)
    blob, off = h.gen_corpus(77, 0, 1 << 20)
    ends, walked = sim.piece_ends_tiled(blob, off)
    ref, _ = sim.piece_ends(blob, off, bits=False)
    assert np.array_equal(ends, ref)
    assert walked < len(blob) // 4096  # the walk-back is the exception, not the rule


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_second_stop_rule_of_the_scanners_short_cut(sims, name):
    """Phase D of tk_k_front decides a piece without a scanner when it starts at a certain start and the next stop is certain -- and (round
    6) when the start is a one-byte char of a prefix class, the next byte a letter and an uncertain stop, and the stop BEHIND that one is
    certain: the piece then ends there (tk_chunk.h, tk_chunk_second_stop).  Checked against the sequential scanner wherever the rule
    applies: every string of up to four class representatives around a prefix char and a letter, 40 000 random adversarial strings, the
    corpora.  (gpt2 / r50k: the rule never applies.)"""
    sim = sims[name]
    alpha = [c for v in REPS.values() for c in v]
    docs = ["".join(t).encode() for k in (2, 3, 4) for t in itertools.product(alpha, repeat=k)] if name != "gpt2_shaped" else []
    rng = random.Random(17)
    docs += ["".join(rng.choice(h.ADV) for _ in range(rng.randint(2, 40))).encode() for _ in range(40000)]
    docs += ["".join(rng.choice(alpha) for _ in range(rng.randint(5, 12))).encode() for _ in range(40000)]
    total_applied = 0
    for i in range(0, len(docs), 4096):
        blob, off = h.pack(docs[i:i + 4096])
        bad, applied = sim.second_stop_check(blob, off)
        assert bad == 0, (name, i)
        total_applied += applied
    for mix in (0, 1):
        blob, off = h.gen_corpus(91 + mix, mix, 2 << 20)
        bad, applied = sim.second_stop_check(blob, off)
        assert bad == 0
        total_applied += applied
    assert (total_applied > 20000) == (name != "gpt2_shaped"), total_applied


REPS = {  # one or two representatives per character class, incl. every contraction letter in both cases
    "NL": ["\n", "\r"], "SP": [" "], "WSO": ["\t", "　"], "LU": ["S", "L", "E", "Z", "ǅ"], "LL": ["s", "l", "e", "ſ", "x", "t"],
    "LC": ["中", "ʰ"], "MK": ["́"], "NU": ["1", "²"], "AP": ["'"], "SL": ["/"], "OT": ["!", "\x1c"],
}


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_certain_start_rule_exhaustive_short_strings(sims, name):
    """Every string of length <= 4 over the class representatives (plus random longer ones): the
    segmentation that starts scanners only at `tk_certain_start` positions must equal the sequential
    split.  A wrong entry in the certain-start table shows up here as a spurious or missing boundary."""
    sim, C = sims[name], h.c_oracle_for(name)
    alpha = [c for v in REPS.values() for c in v]
    docs = []
    for n in (1, 2, 3):
        docs += ["".join(t).encode() for t in itertools.product(alpha, repeat=n)]
    rng = random.Random(5)
    docs += ["".join(rng.choice(alpha) for _ in range(rng.randint(4, 9))).encode() for _ in range(20000)]
    for i in range(0, len(docs), 512):
        chunk = docs[i:i + 512]
        blob, off = h.pack(chunk)
        ref = _ref_ends(C, chunk, off)
        assert sim.piece_ends(blob, off)[0].tolist() == ref
        assert sim.piece_ends(blob, off, bits=True)[0].tolist() == ref


@pytest.mark.parametrize("name,mix", [("gpt2_shaped", 1), ("cl100k_shaped", 0), ("o200k_shaped", 1)])
def test_pretok_and_pieces_on_corpus(sims, name, mix):
    sim, C = sims[name], h.c_oracle_for(name)
    blob, off = h.gen_corpus(0x51 + mix, mix, 1 << 19)
    bb = blob.tobytes()
    docs = [bb[int(off[d]):int(off[d + 1])] for d in range(len(off) - 1)]
    ends, n_certain = sim.piece_ends(blob, off)
    ref = _ref_ends(C, docs, off)
    assert ends.tolist() == ref
    assert n_certain / len(ref) > 0.9  # certain starts must stay dense or the pre-tokeniser loses parallelism
    ends_b, n_fallback = sim.piece_ends(blob, off, bits=True)
    assert ends_b.tolist() == ref
    assert n_fallback / len(ref) < 0.001  # the byte-walking fallback must stay the exception
    p = 0
    for e in ref[:20000]:
        piece = bb[p:e]
        p = e
        assert sim.encode_piece(piece) == C.encode_piece(piece), piece


@pytest.mark.parametrize("name", h.ENCODING_NAMES + ["edu600"])
def test_piece_encode_sim_on_golden_pieces(sims, name):
    """Whole-piece probe + merge through the device tables == golden tokens, piece by piece."""
    g = h.load_golden(name)
    sim = sims[name] if name in sims else h.HostSim(g["pat_str"], h.golden_vocab(name), g["special_tokens"])
    C = h.c_oracle_for(name)
    for c in g["cases"][:200]:
        if c["allowed"] is not None:
            continue
        got, p = [], 0
        for e in C.split(c["text"]):
            got += sim.encode_piece(c["text"][p:e])
            p = e
        assert got == c["tokens"], c["name"]


@pytest.mark.parametrize("name", ["o200k_shaped", "cl100k_shaped"])
def test_identity_lookup_of_long_pieces_is_exact(sims, name):
    """The front kernel looks pieces of 9..23 bytes up by their IDENTITY (tk_common.h tk_ident: the bytes themselves in three words) in a
    table of 32-byte slots, and keys its in-call table of pieces that are not tokens by the same identity: (i) every vocabulary token of
    those lengths is found with its rank, whatever bytes follow it in the text; (ii) strings that are not tokens -- tokens with a byte
    changed, cut short, extended, random bytes -- are not found; (iii) the identity is injective: different byte strings of up to 23 bytes
    never share one, equal strings always do (which is what makes a duplicate in the in-call table a duplicate); (iv) a longer piece's
    identity carries its length and its first and last eight bytes."""
    sim, ranks = sims[name], h.load_vocab(name)
    rng = np.random.default_rng(5)
    toks = [t for t in ranks if 9 <= len(t) <= 23]
    assert len(toks) > 1000
    pick = [toks[i] for i in rng.choice(len(toks), size=min(len(toks), 6000), replace=False)]
    seen = {}
    for t in pick:
        r0, id0 = sim.lookup_xl(t, 0)
        r1, id1 = sim.lookup_xl(t, 0xFF)
        r2, id2 = sim.lookup_xl(t, 0x41)
        assert r0 == r1 == r2 == ranks[t] and id0 == id1 == id2, t
        assert seen.setdefault(id0[:3], t) == t
    n_absent = 0
    for t in pick[:3000]:
        for v in (t[:-1] + bytes([t[-1] ^ 1]), bytes([t[0] ^ 0x20]) + t[1:], t[:-1], t + b"x", t[: len(t) // 2] + b"\x00" + t[len(t) // 2 + 1:], bytes(rng.integers(0, 256, len(t), dtype=np.uint8))):
            if not 9 <= len(v) <= 23:
                continue
            r, idv = sim.lookup_xl(v, int(rng.integers(0, 256)))
            assert r == ranks.get(v, 0xFFFFFFFF), v
            assert seen.setdefault(idv[:3], v) == v  # (injective: an identity seen before belongs to these very bytes)
            n_absent += v not in ranks
    assert n_absent > 5000
    # every length 1..23: identities of different strings differ, a zero byte at the end or in the middle is not padding
    for L in range(1, 24):
        base = bytes(rng.integers(1, 256, L, dtype=np.uint8))
        ids = {}
        for v in {base, base[:-1] + b"\x00", b"\x00" + base[1:], base[: L // 2] + b"\x00" + base[L // 2 + 1:], base[:-1], base + b"\x00"}:
            if 1 <= len(v) <= 23:
                _, idv = sim.lookup_xl(v, 0xA5)
                assert ids.setdefault(idv[:3], v) == v
                assert idv[2] >> 56 == len(v)
    # tokens of more than 23 bytes pass the filter tk_k_bincount asks before it looks a piece up; few other strings do
    longs = [t for t in ranks if len(t) > 23]
    assert all(sim.lookup_xl(t, int(rng.integers(0, 256)))[1][4] == 1 for t in longs[:4000])
    passed = sum(sim.lookup_xl(bytes(rng.integers(0, 256, int(rng.integers(24, 200)), dtype=np.uint8)), 0)[1][4] for _ in range(4000))
    assert passed < 400, passed
    for L in (24, 25, 31, 32, 33, 100, 1024):
        v = bytes(rng.integers(0, 256, L, dtype=np.uint8))
        r, idv = sim.lookup_xl(v, 7)
        assert r == 0xFFFFFFFF and idv[0] == int.from_bytes(v[:8], "little") and idv[1] == int.from_bytes(v[-8:], "little")
        assert idv[2] >> 63 == 1 and (idv[2] >> 32) & 0x7FFFFFFF == L


def test_pair_table_is_exactly_the_set_of_token_splits():
    """(id_a, id_b) -> id_ab must exist iff bytes(a)+bytes(b) is a vocabulary key (tk_common.h)."""
    ranks = h.load_vocab("gpt2_shaped")
    sim = h.HostSim(h.PAT_STR[0], ranks, {})
    n = 0
    for tb in list(ranks)[::17]:
        for s in range(1, len(tb)):
            if tb[:s] in ranks and tb[s:] in ranks:
                n += 1
    assert sim.n_pairs() >= n > 0


def test_rejects_bad_vocabularies():
    with pytest.raises(ValueError):  # duplicate ranks (src/lib.rs:636-641 panics)
        h.HostSim(h.PAT_STR[0], {**{bytes([b]): b for b in range(256)}, b"ab": 5}, {})
    with pytest.raises(ValueError):  # a missing single byte (src/lib.rs:201-203 would panic at encode time)
        h.HostSim(h.PAT_STR[0], {bytes([b]): b for b in range(255)}, {})
    with pytest.raises(ValueError, match="look-behind"):  # a pattern neither the scanner families nor the generic engine take
        h.HostSim(r"(?<!a*b)\w+|\s+", {bytes([b]): b for b in range(256)}, {})


def test_two_special_strings_may_share_an_id():
    """o200k_harmony registers <|endofprompt|> and <|reserved_200018|> under the same id (reference
    tiktoken_ext/openai_public.py:85-94); the reference accepts that (HashMap collect, src/lib.rs:643-646)."""
    ranks = {bytes([b]): b for b in range(256)}
    specials = {"<|endofprompt|>": 200018, "<|reserved_200018|>": 200018, "<|endoftext|>": 199999}
    h.HostSim(h.PAT_STR[2], ranks, specials)  # builds
    with pytest.raises(ValueError):  # (the same STRING twice cannot be expressed in a dict; two spellings that collide can)
        h.HostSim(h.PAT_STR[2], ranks, {"<|a|>": 1000, "": 1001})


# ---------------------------------------------------------------- 16-bytes-per-lane classification (tk_chunk.h)
def _spec_bitmaps(data: bytes, specials):
    """(ss, si) bitmaps of the non-overlapping occurrences of the given special strings, leftmost first."""
    n = len(data)
    ss = np.zeros(n // 32 + 2, np.uint32)
    si = np.zeros(n // 32 + 2, np.uint32)
    pos = 0
    while True:
        hits = [(data.find(s.encode(), pos), s) for s in specials]
        hits = [(i, s) for i, s in hits if i >= 0]
        if not hits:
            break
        i, s = min(hits)
        ss[i >> 5] |= np.uint32(1 << (i & 31))
        for j in range(i + 1, i + len(s.encode())):
            si[j >> 5] |= np.uint32(1 << (j & 31))
        pos = i + len(s.encode())
    return ss, si


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_chunk_classification_equals_per_byte_reference(name):
    """Phases A-C of tk_k_front -- classes, char starts, hard starts and certain starts computed 16 bytes per lane from bit planes
    (tk_chunk.h) -- against tk_class_byte / tk_certain_start position by position, on corpora, on adversarial strings, with
    documents and the text ending at every offset of a chunk, with tiny windows (many window edges) and with special tokens."""
    sim = h.HostSim(h.PAT_STR[h.PATTERN_OF[name]], h.load_vocab(name), h.SPECIALS[name])
    for mix in (0, 1):
        blob, off = h.gen_corpus(0xC0FFEE + mix, mix, 1 << 19)
        assert sim.chunk_check(blob, off) == (0, 0, 0)
        assert sim.chunk_check(blob, off, tile=160, left=32, win=256) == (0, 0, 0)
        assert sim.chunk_check(blob, off, tile=64, left=16, win=128) == (0, 0, 0)
    rng = np.random.default_rng(7)
    docs = ["".join(rng.choice(h.ADV, size=int(rng.integers(0, 40)))).encode() for _ in range(3000)]
    blob, off = h.pack(docs)
    assert sim.chunk_check(blob, off) == (0, 0, 0)
    assert sim.chunk_check(blob, off, tile=64, left=16, win=128) == (0, 0, 0)
    for cut in range(1, 40):  # the text ends at every position of a chunk; 4-byte chars at every alignment
        data = ("😀é中a" * 12).encode()[:cut]
        try:
            data.decode()
        except UnicodeDecodeError:
            continue
        blob, off = h.pack([data])
        assert sim.chunk_check(blob, off) == (0, 0, 0), cut
    # special tokens: first / interior bytes come from the marking kernels as bitmaps
    text = ("hello <|endoftext|> wörld<|endoftext|><|endoftext|>\n 中<|endoftext|>" * 40).encode()
    ss, si = _spec_bitmaps(text, ["<|endoftext|>"])
    blob, off = h.pack([text])
    assert sim.chunk_check(blob, off, ss, si) == (0, 0, 0)
    assert sim.chunk_check(blob, off, ss, si, tile=64, left=16, win=128) == (0, 0, 0)


def test_run_query_scanner_equals_the_byte_walking_one():
    """tk_piece_end_runs (the scanner over run queries that the front kernel uses for pieces that leave a tile's window) against
    tk_piece_end at every piece start of corpora and adversarial documents, all three patterns."""
    rng = np.random.default_rng(11)
    before = h.sim_lib().tks_runs_mismatches()
    never_before = h.sim_lib().tks_never_violations()
    for name in h.ENCODING_NAMES:
        sim = h.HostSim(h.PAT_STR[h.PATTERN_OF[name]], h.load_vocab(name), h.SPECIALS[name])
        for mix in (0, 1):
            blob, off = h.gen_corpus(0xF1A7 + mix, mix, 1 << 19)
            sim.piece_ends(blob, off)
        docs = ["".join(rng.choice(h.ADV, size=int(rng.integers(0, 40)))).encode() for _ in range(5000)]
        sim.piece_ends(*h.pack(docs))
        for rep in ("x", " ", "中", "1", "\n", "!", "a\u0301", "Aa", " \n"):
            sim.piece_ends(*h.pack([(rep * 700).encode(), ("z" + rep * 300 + "'ll").encode()]))
        # the table of impossible boundaries (tk_never_mask: the front kernel skips the scanner on its word) on contraction-heavy text
        apo = list("abdelmrstvxABDELMRSTVX") + ["'"] * 8 + [" ", " ", "\n", "1", "!", "/", "\u017f", "\u0301", "\u4e2d", "\t", "\u01c5"]
        docs = ["".join(rng.choice(apo, size=int(rng.integers(0, 30)))).encode() for _ in range(20000)]
        sim.piece_ends(*h.pack(docs))
    assert h.sim_lib().tks_runs_mismatches() == before
    assert h.sim_lib().tks_never_violations() == never_before


def test_scanners_under_the_sanitizers(tmp_path):
    """tests/hostsim/scan_sanitize.cpp: every scanner form of the device headers and the 16-bytes-per-lane classification on random text
    (valid UTF-8, truncated chars, stray continuation bytes, NULs, long runs, random document starts) in buffers sized as on the device,
    built with AddressSanitizer and UBSan -- an out-of-bounds read that the GPU would answer with garbage ends the run here.
    (3000 rounds per pattern ran clean as well: 60 000 calls.)"""
    import os
    import subprocess

    exe = str(tmp_path / "scan_sanitize")
    d = os.path.join(h.ROOT, "tests", "hostsim")
    csrc = os.path.join(h.ROOT, "tiktoken_amd", "csrc")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-D_GLIBCXX_SANITIZE_VECTOR", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                         "-Wno-unused-function", os.path.join(d, "scan_sanitize.cpp"), os.path.join(csrc, "tk_tables.cpp"), os.path.join(csrc, "tk_pattern.cpp"),
                         os.path.join(csrc, "tk_regex.cpp"), "-pthread", "-o", exe], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert cc.returncode == 0, cc.stderr[-2000:]
    qwen2 = h.PAT_STR[1].replace(r"\p{N}{1,3}", r"\p{N}")
    import gzip

    vocab = tmp_path / "o200k_shaped.tiktoken"  # (with it: the whole-piece probes and the per-lane merge on random pieces, 15 000 of them)
    vocab.write_bytes(gzip.open(os.path.join(h.ROOT, "tiktoken_amd", "vocab", "o200k_shaped.tiktoken.gz")).read())
    run = subprocess.run([exe, h.PAT_STR[0], h.PAT_STR[1], qwen2, h.PAT_STR[2]], capture_output=True, text=True, timeout=900,
                         env={**os.environ, "TK_SAN_ROUNDS": "300", "TK_SAN_VOCAB": str(vocab)})
    assert run.returncode == 0 and run.stdout.startswith("ok "), (run.stdout[-500:], run.stderr[-3000:])


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_scanner_forms_on_fuzzed_documents(name):
    """The awkward documents of the GPU fuzzer (helpers.fuzz_batch: runs of 20 000 units, chains of uncertain boundaries, near-specials) through
    every scanner form of the device headers on the CPU, against the oracle's split."""
    import zlib

    sim, C = h.HostSim(h.load_golden(name)["pat_str"], {bytes([b]): b for b in range(256)}, {}), h.c_oracle_for(name)
    for seed in range(2):
        blob, off = h.pack(h.fuzz_batch(zlib.crc32(name.encode()) + 100 + seed, 2 << 20))
        bb, ref = blob.tobytes(), []
        for d in range(len(off) - 1):
            a, b = int(off[d]), int(off[d + 1])
            if b > a:
                ref += [a + e for e in C.split(bb[a:b])]
        assert sim.piece_ends_tiled(blob, off, 3840, 128)[0].tolist() == ref
        assert sim.piece_ends(blob, off, bits=True)[0].tolist() == ref
        assert sim.piece_ends(blob, off)[0].tolist() == ref
        assert sim.chunk_check(blob, off)[0] == 0


def test_tables_do_not_depend_on_the_thread_count(monkeypatch):
    """the pair table is built by several host threads (tk_tables.cpp); every table the device gets -- and the sorted token
    list -- must be byte for byte what one thread builds."""
    g = h.load_golden("o200k_shaped")
    ranks = h.golden_vocab("o200k_shaped")
    digests = []
    for nth in ("1", "3", "8", "32"):
        monkeypatch.setenv("TIKTOKEN_AMD_BUILD_THREADS", nth)
        sim = h.HostSim(g["pat_str"], ranks, g["special_tokens"])
        digests.append((sim.n_pairs(), h.sim_lib().tks_tables_digest(sim._h)))
    assert len(set(digests)) == 1, digests


# ---------------------------------------------------------------- encode_mid's plan (tk_mid_plan.h)
def _mid_docs():
    """Documents of 2 .. 128 KiB: Lorem ipsum at the sizes around the limits, slices of the synthetic corpora, random awkward text."""
    rng = random.Random(23)
    docs = [h.lorem(n) for n in (2049, 2050, 2100, 3000, 3071, 3072, 3073, 4096, 10000, 65536, 131071, 131072)]
    for mix in (0, 1):
        blob, _ = h.gen_corpus(0x5EED0100 + mix, mix, 1 << 20, 4)
        raw = blob.tobytes()
        for _ in range(40):
            n = rng.choice([2100, 4096, 9000, 20000, 65536, 131072])
            a = rng.randrange(0, len(raw) - n)
            docs.append(raw[a:a + n].decode(errors="ignore").encode())
    for _ in range(120):
        want, parts, size = rng.choice([2100, 3000, 5000, 12000, 40000]), [], 0
        while size < want:
            u = rng.choice(h.ADV + h.FUZZ_UNITS + [" ", "a ", "Z ", "b  ", "c \n", "d's ", "E'LL ", "é ", "f ", "g 　"])
            parts.append(u)
            size += len(u.encode())
        docs.append("".join(parts).encode())
    return docs


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_mid_plan_cuts_only_where_a_piece_starts_whatever_surrounds_it(sims, name):
    """encode_mid (tk_api.hip) encodes the segments of tk_mid_plan as so many small calls and concatenates their tokens: right only if every cut
    is a piece start of the whole document AND the pieces of a segment on its own are the document's (reference src/lib.rs:236-249: pieces
    are encoded one by one).  Checked with the oracle on both sides; plus the plan's own promises."""
    sim, C = sims[name], h.c_oracle_for(name)
    L = h.sim_lib()
    seg_max, slots, planned = L.tks_mid_segment_max(), L.tks_mid_slots(), L.tks_mid_segments()
    assert (seg_max, slots, planned) == (2048, 72, 64)
    taken = 0
    for doc in _mid_docs():
        cuts, why = sim.mid_plan(doc)
        if cuts is None:
            assert why in ("no cut in a window", "more segments than slots"), why
            continue
        taken += 1
        n = len(doc)
        assert cuts[0] == 0 and cuts[-1] == n and 3 <= len(cuts) <= slots + 1
        assert all(0 < b - a <= seg_max for a, b in zip(cuts, cuts[1:])), cuts
        for c in cuts[1:-1]:
            assert doc[c:c + 1] == b" " and doc[c - 1:c].isalpha() and doc[c - 1] < 0x80, (c, doc[c - 2:c + 2])
        starts = {0, *C.split(doc)}  # (split gives piece ends = the next piece's start)
        assert all(c in starts for c in cuts[1:-1])
        whole = C.encode_ordinary(doc).tolist()
        parts = [t for a, b in zip(cuts, cuts[1:]) for t in C.encode_ordinary(doc[a:b]).tolist()]
        assert parts == whole
    assert taken > 100  # (the corpora and Lorem ipsum are all taken)


def test_mid_plan_refuses_what_it_cannot_cut(sims):
    sim = sims["cl100k_shaped"]
    for doc, reason in [(b"a" * 5000, "no cut in a window"), (b"1 " * 3000, "no cut in a window"), ("中文 ".encode() * 900, "no cut in a window"),
                        (b"ab " * 20 + b"x" * 4000, "no cut in a window"), (b"ab " * 600, None), (b"ab " * 683, None),
                        # cuts only in the first half of every 2 KiB; one cut every 1030 bytes: segments of 1030, the slots run out before the text does
                        ((b"ab " * 300 + b"x" * 1148) * 64, None), ((b"x" * 1028 + b"b ") * 127, "more segments than slots")]:
        cuts, why = sim.mid_plan(doc)
        if reason is None:
            assert cuts is not None, why
            assert cuts[-1] == len(doc) and all(b - a <= 2048 for a, b in zip(cuts, cuts[1:]))
        else:
            assert cuts is None and why == reason, (why, cuts)
    # one segment's worth and less never gets here in the product; the plan says "one segment"
    assert sim.mid_plan(b"ab " * 300) == (None, "one segment")


def test_mid_plan_is_off_for_patterns_without_the_rule():
    """A pattern of the generic engine has no class table to vouch for the cut: no plan (tk_core::mid_cut is false; has_rx keeps encode_mid out)."""
    rx = h.HostSim(r"\w+|\s+|[^\w\s]+", h.load_vocab("gpt2_shaped"), {})
    assert rx.mid_plan(h.lorem(5000)) == (None, "a pattern of the generic engine")
