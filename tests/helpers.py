"""Shared test helpers: vocabularies, golden fixtures, corpus generator, oracle handles."""
from __future__ import annotations

import base64
import ctypes
import functools
import gzip
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import c_oracle, py_oracle  # noqa: E402  (tests are allowed to use the oracle)

SPECIALS = {
    "gpt2_shaped": {"<|endoftext|>": 50256},
    "cl100k_shaped": {"<|endoftext|>": 100257, "<|fim_prefix|>": 100258, "<|fim_middle|>": 100259,
                      "<|fim_suffix|>": 100260, "<|endofprompt|>": 100276},
    "o200k_shaped": {"<|endoftext|>": 199999, "<|endofprompt|>": 200018},
}
PATTERN_OF = {"gpt2_shaped": 0, "cl100k_shaped": 1, "o200k_shaped": 2, "edu600": 0}
PAT_STR = {0: py_oracle.R50K_PAT, 1: py_oracle.CL100K_PAT, 2: py_oracle.O200K_PAT}
ENCODING_NAMES = ["gpt2_shaped", "cl100k_shaped", "o200k_shaped"]


@functools.lru_cache(maxsize=None)
def load_vocab(name: str) -> dict[bytes, int]:
    d = {}
    with gzip.open(os.path.join(ROOT, "tiktoken_amd", "vocab", name + ".tiktoken.gz")) as f:
        for line in f.read().splitlines():
            t, r = line.split()
            d[base64.b64decode(t)] = int(r)
    return d


@functools.lru_cache(maxsize=None)
def load_golden(name: str) -> dict:
    with gzip.open(os.path.join(ROOT, "tests", "golden", name + ".json.gz")) as f:
        g = json.loads(f.read())
    for c in g["cases"]:
        c["text"] = base64.b64decode(c["text"])
    if g.get("mergeable_ranks"):
        g["mergeable_ranks"] = {base64.b64decode(k): v for k, v in g["mergeable_ranks"]}
    return g


def golden_vocab(name: str) -> dict[bytes, int]:
    g = load_golden(name)
    return g["mergeable_ranks"] if g.get("mergeable_ranks") else load_vocab(g["vocab"])


@functools.lru_cache(maxsize=None)
def c_oracle_for(name: str) -> c_oracle.COracle:
    g = load_golden(name) if os.path.exists(os.path.join(ROOT, "tests", "golden", name + ".json.gz")) else None
    specials = g["special_tokens"] if g else SPECIALS[name]
    return c_oracle.COracle(PATTERN_OF[name], golden_vocab(name) if g else load_vocab(name), specials)


# ---------------------------------------------------------------- corpus
_corpus_lib = None


def corpus_lib():
    global _corpus_lib
    if _corpus_lib is None:
        path = os.path.join(ROOT, "tiktoken_amd", "csrc", "libtkcorpus.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", os.path.dirname(path), "libtkcorpus.so"])
        lib = ctypes.CDLL(path)
        lib.tkc_generate.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int]
        _corpus_lib = lib
    return _corpus_lib


def gen_corpus(seed: int, mix: int, nbytes: int, threads: int = 8):
    """(blob uint8[nbytes], doc_off uint64[n_docs+1]) -- SURVEY.md 8(d) synthetic corpora."""
    out = np.empty(nbytes + 64, np.uint8)
    out[nbytes:] = 0
    maxd = nbytes // 64 + 2
    off = np.empty(maxd + 1, np.uint64)
    nd = ctypes.c_uint64()
    rc = corpus_lib().tkc_generate(seed, mix, nbytes, out.ctypes.data, off.ctypes.data, maxd, ctypes.byref(nd), threads)
    assert rc == 0
    return out[:nbytes], off[: nd.value + 1].copy()


def natural_corpus(ranks: dict[bytes, int], nbytes: int, seed: int = 0x5EED0006, miss_share: float = 0.02):
    """Text whose share of pieces that are NOT vocabulary tokens is what natural text has (1-3 %; the C3 / C4 / C5 corpora: 18.7 %, their
    lexicon is far larger than the vocabulary): words drawn from the vocabulary itself -- the tokens that are a whole piece of the o200k /
    cl100k patterns, a space and ASCII letters, Zipf over their ranks (a low rank is a frequent word) -- plus `miss_share` words of random
    letters, in sentences with commas, full stops, digits and line breaks, cut into documents of about 2 KiB.  (blob, doc_off) like gen_corpus."""
    rng = np.random.default_rng(seed)
    words = sorted((t for t in ranks if len(t) >= 2 and t[:1] == b" " and t[1:].isalpha() and t[1:].isascii() and (t[1:].islower() or t[1:].istitle())),
                   key=lambda t: ranks[t])
    n_voc = len(words)
    n_rand = max(n_voc // 2, 1000)
    letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8)
    for _ in range(n_rand):  # words of random letters: next to none of them is a token
        words.append(b" " + letters[rng.integers(0, 26, int(rng.integers(7, 14)))].tobytes())
    extra = [b",", b".", b".\n", b" 2024", b" 17", b"?", b"\n\n", b" -", b":"]
    words += extra
    lens = np.array([len(w) for w in words], np.int64)
    maxlen = int(lens.max())
    table = np.zeros((len(words), maxlen), np.uint8)
    for i, w in enumerate(words):
        table[i, : len(w)] = np.frombuffer(w, np.uint8)
    # probabilities: vocabulary words Zipf(1.0) over their order; random words uniform, `miss_share` in all; punctuation 12 %
    p = np.zeros(len(words))
    z = 1.0 / np.arange(1, n_voc + 1)
    p[:n_voc] = z / z.sum() * (1.0 - miss_share - 0.12)
    p[n_voc:n_voc + n_rand] = miss_share / n_rand
    p[n_voc + n_rand:] = 0.12 / len(extra)
    n_words = int(nbytes / float((p * lens).sum()) * 1.02) + 16
    idx = rng.choice(len(words), size=n_words, p=p / p.sum())
    wl = lens[idx]
    ends = np.cumsum(wl)
    n_take = int(np.searchsorted(ends, nbytes, side="right"))
    idx, wl, ends = idx[:n_take], wl[:n_take], ends[:n_take]
    total = int(ends[-1])
    blob = np.zeros(total + 64, np.uint8)
    starts = ends - wl
    for c in range(maxlen):  # (a column of the word table at a time)
        m = wl > c
        blob[starts[m] + c] = table[idx[m], c]
    # documents: a cut every ~2 KiB, at a word boundary
    cuts = starts[np.searchsorted(starts, np.arange(2048, total, 2048))]
    off = np.unique(np.concatenate([[0], cuts, [total]])).astype(np.uint64)
    return blob[:total], off


def insert_specials(blob: np.ndarray, off: np.ndarray, seed: int = 5):
    """Config C5: special tokens <|custom_0..7|> at a mean of one per ~2 KiB plus decoys (an unregistered
    <|custom_9|>, truncated specials, bare delimiters), inserted at char boundaries of every document."""
    bb = blob.tobytes()
    rng = np.random.default_rng(seed)
    decoys = [b"<|custom_9|>", b"<|endoftext", b"<|custom_3|", b"<|", b"|>"]
    parts, lens = [], []
    for d in range(len(off) - 1):
        t = bb[int(off[d]):int(off[d + 1])]
        out, pos = [], 0
        while pos < len(t):
            step = int(rng.integers(512, 3584))
            cut = min(len(t), pos + step)
            while cut < len(t) and (t[cut] & 0xC0) == 0x80:
                cut += 1
            out.append(t[pos:cut])
            if cut < len(t):
                out.append(b"<|custom_%d|>" % rng.integers(0, 8) if rng.random() < 0.8 else decoys[int(rng.integers(0, len(decoys)))])
            pos = cut
        doc = b"".join(out)
        parts.append(doc)
        lens.append(len(doc))
    blob5 = np.frombuffer(b"".join(parts) + b"\0" * 64, np.uint8)
    off5 = np.zeros(len(lens) + 1, np.uint64)
    off5[1:] = np.cumsum(lens)
    return blob5[: int(off5[-1])], off5


CUSTOM8 = {**SPECIALS["o200k_shaped"], **{f"<|custom_{i}|>": 200019 + i for i in range(8)}}


@functools.lru_cache(maxsize=1)
def _n1_corpus():
    return natural_corpus(load_vocab("o200k_shaped"), 256 << 20)


def baseline_config(cfg: str, threads: int = 16):
    """(encoding name for the vocabulary, pattern id, special tokens, blob, doc_off, allowed_special) of a BASELINE.json config at its
    FULL size: C1 gpt2 1 MiB Lorem ipsum (one document); C2 cl100k 64 MiB mixed UTF-8; C3 o200k 1 GiB web text;
    C5 o200k + 8 custom specials, 256 MiB web text, allowed_special='all'."""
    if cfg == "C1":
        blob, off = pack([lorem(1 << 20)])
        return "gpt2_shaped", 0, SPECIALS["gpt2_shaped"], blob, off, None
    if cfg == "C2":
        blob, off = gen_corpus(0x5EED0002, 0, 64 << 20, threads)
        return "cl100k_shaped", 1, SPECIALS["cl100k_shaped"], blob, off, None
    if cfg == "C3":
        blob, off = gen_corpus(0x5EED0003, 1, 1 << 30, threads)
        return "o200k_shaped", 2, SPECIALS["o200k_shaped"], blob, off, None
    if cfg.startswith("C4r"):  # C4: 8 GiB doc-sharded over 8 GPUs = rank r's 1 GiB shard, the seeds bench.py --gpus N uses (0x5EED0004 + rank)
        blob, off = gen_corpus(0x5EED0004 + int(cfg[3:]), 1, 1 << 30, threads)
        return "o200k_shaped", 2, SPECIALS["o200k_shaped"], blob, off, None
    if cfg == "N1":  # not a BASELINE.json configuration: 256 MiB of text whose miss rate is natural (natural_corpus), o200k-shaped
        blob, off = _n1_corpus()
        return "o200k_shaped", 2, SPECIALS["o200k_shaped"], blob, off, None
    if cfg == "N1g":  # the same text at the headline's size (1 GiB), so that the 18.7 %-miss headline corpus and the natural-miss-rate figure stand
        # side by side at one size: N1's 256 MiB four times, each copy with its documents rotated by a quarter (the generator takes 30 s per 256 MiB;
        # its pool of words is finite, so a longer run of it repeats its words just the same)
        b, o = _n1_corpus()
        nd = len(o) - 1
        parts, lens = [], []
        ln = np.diff(o.astype(np.int64))
        for k in range(4):
            d = nd * k // 4
            cut = int(o[d])
            parts += [b[cut:], b[:cut]]
            lens += [ln[d:], ln[:d]]
        off = np.zeros(4 * nd + 1, np.uint64)
        off[1:] = np.cumsum(np.concatenate(lens))
        return "o200k_shaped", 2, SPECIALS["o200k_shaped"], np.concatenate(parts), off, None
    if cfg == "C5":
        blob, off = insert_specials(*gen_corpus(0x5EED0005, 1, 256 << 20, threads))
        return "o200k_shaped", 2, CUSTOM8, blob, off, "all"
    raise KeyError(cfg)


LOREM = ("Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor incididunt ut labore et "
         "dolore magna aliqua. Ut enim ad minim veniam, quis nostrud exercitation ullamco laboris nisi ut aliquip ex "
         "ea commodo consequat. Duis aute irure dolor in reprehenderit in voluptate velit esse cillum dolore eu "
         "fugiat nulla pariatur. Excepteur sint occaecat cupidatat non proident, sunt in culpa qui officia deserunt "
         "mollit anim id est laborum. ")


def lorem(nbytes: int) -> bytes:
    return (LOREM * (nbytes // len(LOREM) + 1))[:nbytes].encode()


ADV = list("aAsStTlLvVeErRdDmMxZ") + ["ſ", "中", "́", "ʰ", "ǅ", "1", "2", "²", "٣", " ", " ", "\t", "\r", "\n", "　",
                                        "\x85", "'", "/", "!", ".", "\x1c", "é", "Ω", "я", "ก", "ั", "😀", "’", "hello",
                                        " world", "ing", "tion", "<|", "|>", "<|endoftext|>"]


# Units of the fuzzed documents (tools/gpu_fuzz.py, test_fuzzed_awkward_documents): runs of one class, chains of uncertain boundaries,
# contractions in every case, digits, white space with and without newlines, CJK, combining marks, special tokens, near-specials.
FUZZ_UNITS = ["x", "X", "Ab", "aB", "x'll", "X'LL", "y's", "'t", "'", "''", "1", "12", "1234567", " ", "  ", "\n", "\r\n", " \n", "\t", "\u00a0",
              "\u3000", "\u4e2d", "\u4e2d\u6587", "\u00e9", "\u00c9", "\u0301", "a\u0301", "\u01c5", "\u02b0", "!", "...", "/", "//", "!/\n", "-", "=",
              "\U0001F600", "\u017f", "\u212a", "hello", "World", " the", " of", ",", ".", "https://example.com/a/b?c=d", "foo_bar",
              "camelCaseWord", "x'sS", "don't", "DON'T", "I'll", " a's", "<|endoftext|>", "<|endofprompt|>", "<|endoftext", "\u0661\u0662\u0663",
              "\u00b2", "\x7f", "\u200b", "\u2028", "\u1680"]


def fuzz_doc(rng) -> str:
    """One awkward document (rng: random.Random)."""
    r = rng.random()
    if r < 0.05:
        return ""
    if r < 0.15:
        return rng.choice(FUZZ_UNITS)
    parts, size, target = [], 0, int(rng.lognormvariate(8.5, 1.6))
    while size < target:
        u = rng.choice(FUZZ_UNITS)
        k = rng.choice([1, 1, 1, 2, 3, 5, 17, 64, 300, 2000, 20000]) if rng.random() < 0.3 else 1
        seg = u * k
        parts.append(seg)
        size += len(seg)
        if rng.random() < 0.5:
            parts.append(rng.choice([" ", "", "", "\n"]))
    return "".join(parts)


def fuzz_batch(seed: int, nbytes: int) -> list[bytes]:
    import random

    rng = random.Random(seed)
    docs, total = [], 0
    while total < nbytes:
        d = fuzz_doc(rng).encode()
        docs.append(d)
        total += len(d)
    return docs


def pack(docs: list[bytes]):
    blob = np.frombuffer(b"".join(docs) + b"\0" * 64, np.uint8)
    off = np.zeros(len(docs) + 1, np.uint64)
    if docs:
        off[1:] = np.cumsum([len(d) for d in docs], dtype=np.uint64)
    return blob[: int(off[-1])] if len(docs) else blob[:0], off


# ---------------------------------------------------------------- CPU simulation of the device logic
_sim_lib = None


def sim_lib():
    global _sim_lib
    if _sim_lib is None:
        d = os.path.join(ROOT, "tests", "hostsim")
        so = os.path.join(d, "libtk_hostsim.so")
        srcs = [os.path.join(d, "tk_hostsim.cpp")] + [os.path.join(ROOT, "tiktoken_amd", "csrc", f)
                                                       for f in ("tk_tables.cpp", "tk_pattern.cpp", "tk_regex.cpp", "tk_device.h", "tk_common.h", "tk_tables.h", "tk_chunk.h", "tk_regex.h",
                                                                 "tk_regex_split.h", "tk_regex_host.h", "tk_regex_dfa.inc", "tk_mid_plan.h", "tk_regex_casefold.inc")]
        def stale():
            return not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs)

        if stale():  # (several pytest-xdist workers may get here at once: one builds, into a file of its own, and renames)
            import fcntl
            with open(so + ".lock", "w") as lk:
                fcntl.flock(lk, fcntl.LOCK_EX)
                if stale():
                    tmp = f"{so}.{os.getpid()}.tmp"
                    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", srcs[0], srcs[1], srcs[2], srcs[3], "-o", tmp])
                    os.replace(tmp, so)
        L = ctypes.CDLL(so)
        vp, u64 = ctypes.c_void_p, ctypes.c_uint64
        L.tks_create.restype = vp
        L.tks_create.argtypes = [vp, vp, vp, u64, vp, vp, vp, u64, ctypes.c_char_p, ctypes.c_char_p, u64]
        L.tks_destroy.argtypes = [vp]
        L.tks_n_pairs.restype = u64
        L.tks_n_pairs.argtypes = [vp]
        L.tks_tables_digest.restype = u64
        L.tks_tables_digest.argtypes = [vp]
        L.tks_pretok.restype = u64
        L.tks_pretok.argtypes = [vp, vp, u64, vp, u64, vp]
        L.tks_pretok_bits.restype = u64
        L.tks_pretok_bits.argtypes = [vp, vp, u64, vp, u64, vp]
        L.tks_pretok_tiles.restype = u64
        L.tks_pretok_tiles.argtypes = [vp, vp, u64, vp, u64, vp, ctypes.c_uint32, ctypes.c_uint32]
        L.tks_second_stop_check.restype = u64
        L.tks_second_stop_check.argtypes = [vp, vp, u64, vp, u64, vp]
        L.tks_runs_mismatches.restype = u64
        L.tks_never_violations.restype = u64
        L.tks_table_stats.argtypes = [vp, vp, vp]
        L.tks_lookup.restype = ctypes.c_uint32
        L.tks_lookup.argtypes = [vp, vp, ctypes.c_uint32]
        L.tks_lookup_xl.restype = ctypes.c_uint32
        L.tks_lookup_xl.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint8, vp]
        L.tks_chunk_check.restype = u64
        L.tks_chunk_check.argtypes = [vp, vp, u64, vp, u64, vp, vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp, vp]
        L.tks_encode_piece.restype = ctypes.c_int64
        L.tks_encode_piece.argtypes = [vp, vp, ctypes.c_uint32, vp]
        L.tks_rx_compile.restype = vp
        L.tks_rx_compile.argtypes = [ctypes.c_char_p, ctypes.c_char_p, u64]
        L.tks_rx_free.argtypes = [vp]
        L.tks_rx_size.restype = u64
        L.tks_rx_size.argtypes = [vp]
        L.tks_rx_dfa.restype = u64
        L.tks_rx_dfa.argtypes = [vp, ctypes.c_char_p, u64]
        L.tks_rx_staged_stats.argtypes = [vp, ctypes.c_int]
        L.tks_rx_split.restype = u64
        L.tks_rx_split.argtypes = [vp, vp, u64, vp, u64, vp, vp, u64, ctypes.c_int, vp, vp]
        L.tks_mid_plan.restype = u64
        L.tks_mid_plan.argtypes = [vp, vp, u64, vp, ctypes.c_char_p, u64]
        for f in (L.tks_mid_slots, L.tks_mid_segment_max, L.tks_mid_segments):
            f.restype = u64
        _sim_lib = L
    return _sim_lib


class HostSim:
    def __init__(self, pat_str: str, ranks: dict[bytes, int], specials: dict[str, int]):
        rb, ro, ri = c_oracle._pack(list(ranks.items()))
        sb, so, si = c_oracle._pack([(k.encode(), v) for k, v in specials.items()])
        err = ctypes.create_string_buffer(512)
        self._h = sim_lib().tks_create(rb.ctypes.data, ro.ctypes.data, ri.ctypes.data, len(ri), sb.ctypes.data,
                                       so.ctypes.data, si.ctypes.data, len(si), pat_str.encode(), err, 512)
        if not self._h:
            raise ValueError(err.value.decode())

    def n_pairs(self):
        return sim_lib().tks_n_pairs(self._h)

    def table_stats(self):
        probes, slots = (ctypes.c_double * 3)(), (ctypes.c_uint64 * 3)()
        sim_lib().tks_table_stats(self._h, probes, slots)
        return list(probes), list(slots)

    def lookup(self, piece: bytes) -> int:
        b = np.frombuffer(piece, np.uint8) if piece else np.zeros(1, np.uint8)
        return sim_lib().tks_lookup(self._h, b.ctypes.data, len(piece))

    def lookup_xl(self, piece: bytes, fill: int = 0):
        """(rank or 0xFFFFFFFF, (w0, w1, w2, hash, filter bit)) of the front kernel's identity lookup (pieces of 9..23 bytes; longer ones: identity
        and whether the filter of the long tokens lets them through)."""
        b = np.frombuffer(piece, np.uint8) if piece else np.zeros(1, np.uint8)
        ident = np.zeros(5, np.uint64)
        r = sim_lib().tks_lookup_xl(self._h, b.ctypes.data, len(piece), fill, ident.ctypes.data)
        return r, tuple(int(x) for x in ident)

    def mid_plan(self, doc: bytes):
        """encode_mid's cuts for one document (tk_mid_plan.h): ([0, c1, ..., n], "") or (None, reason)."""
        L = sim_lib()
        cuts = np.zeros(L.tks_mid_slots() + 1, np.uint32)
        why = ctypes.create_string_buffer(128)
        b = np.frombuffer(doc, np.uint8) if doc else np.zeros(1, np.uint8)
        k = L.tks_mid_plan(self._h, b.ctypes.data, len(doc), cuts.ctypes.data, why, 128)
        return (cuts[:k + 1].tolist(), "") if k else (None, why.value.decode())

    def piece_ends(self, blob: np.ndarray, doc_off: np.ndarray, bits: bool = False):
        """Piece end offsets from the simulated pre-tokeniser; bits=True mirrors the bit-parallel kernel
        (second value = pieces that fell back to the byte-walking scanner), else the byte-walking one
        (second value = number of certain starts)."""
        n = len(blob)
        starts = np.zeros(max(n, 1), np.uint8)
        b = np.ascontiguousarray(blob) if n else np.zeros(1, np.uint8)
        fn = sim_lib().tks_pretok_bits if bits else sim_lib().tks_pretok
        nc = fn(self._h, b.ctypes.data, n, doc_off.ctypes.data, len(doc_off) - 1, starts.ctypes.data)
        idx = np.flatnonzero(starts[:n])
        return np.concatenate([idx[1:], [n]]).astype(np.uint64) if n else np.zeros(0, np.uint64), nc

    def piece_ends_tiled(self, blob: np.ndarray, doc_off: np.ndarray, tile: int = 4096, left: int = 64):
        """Piece ends from the per-tile rule of tk_k_front (second value: tiles that had to walk back)."""
        n = len(blob)
        starts = np.zeros(max(n, 1), np.uint8)
        b = np.ascontiguousarray(blob) if n else np.zeros(1, np.uint8)
        nw = sim_lib().tks_pretok_tiles(self._h, b.ctypes.data, n, doc_off.ctypes.data, len(doc_off) - 1, starts.ctypes.data, tile, left)
        idx = np.flatnonzero(starts[:n])
        return (np.concatenate([idx[1:], [n]]).astype(np.uint64) if n else np.zeros(0, np.uint64)), nw

    def second_stop_check(self, blob: np.ndarray, doc_off: np.ndarray):
        """The "second stop" rule of the front kernel's phase D against the sequential scanner: (violations, places where it applied)."""
        n = len(blob)
        b = np.ascontiguousarray(blob) if n else np.zeros(1, np.uint8)
        applied = ctypes.c_uint64()
        bad = sim_lib().tks_second_stop_check(self._h, b.ctypes.data, n, doc_off.ctypes.data, len(doc_off) - 1, ctypes.byref(applied))
        return int(bad), int(applied.value)

    def chunk_check(self, blob: np.ndarray, doc_off: np.ndarray, ss=None, si=None, tile: int = 3840, left: int = 64, win: int = 4096):
        """Phases A-C of tk_k_front (16 bytes per lane, tk_chunk.h) against the per-byte reference: (mismatches, first position, code)."""
        n = len(blob)
        b = np.ascontiguousarray(blob) if n else np.zeros(1, np.uint8)
        fb, what = ctypes.c_uint64(), ctypes.c_uint32()
        bad = sim_lib().tks_chunk_check(self._h, b.ctypes.data, n, doc_off.ctypes.data, len(doc_off) - 1,
                                        ss.ctypes.data if ss is not None else None, si.ctypes.data if si is not None else None,
                                        tile, left, win, ctypes.byref(fb), ctypes.byref(what))
        return bad, fb.value, what.value

    def encode_piece(self, piece: bytes) -> list[int]:
        out = np.empty(max(len(piece), 1), np.uint32)
        b = np.frombuffer(piece, np.uint8)
        n = sim_lib().tks_encode_piece(self._h, b.ctypes.data, len(piece), out.ctypes.data)
        return out[:n].tolist()


class _DevArray:
    """A device buffer of the library seen through __cuda_array_interface__ (torch.as_tensor reads it in place)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def dev_u32(ptr: int, n: int) -> np.ndarray:
    import torch

    if not n:
        return np.zeros(0, np.uint32)
    return torch.as_tensor(_DevArray(ptr, n, "<i4"), device="cuda").cpu().numpy().view(np.uint32)


def dev_u64(ptr: int, n: int) -> np.ndarray:
    import torch

    return torch.as_tensor(_DevArray(ptr, n, "<i8"), device="cuda").cpu().numpy().astype(np.uint64)


class RxSim:
    """The generic pat_str engine on the CPU (tests/hostsim): the compiler of tk_regex.cpp and the lanes of the two split kernels."""

    def __init__(self, pat_str: str):
        L = sim_lib()
        err = ctypes.create_string_buffer(512)
        self._h = L.tks_rx_compile(pat_str.encode(), err, 512)
        if not self._h:
            raise ValueError(err.value.decode())
        self._L = L
        self.size = int(L.tks_rx_size(self._h))
        self.stats = (0, 0)
        why = ctypes.create_string_buffer(256)
        d = int(L.tks_rx_dfa(self._h, why, 256))
        self.dfa = (d >> 32, d & 0xFFFFFFFF) if d else None  # (states, classes) of the pattern's DFA; None: it has none, dfa_why says why
        self.dfa_why = why.value.decode()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.tks_rx_free(self._h)
            self._h = None

    def split(self, docs: list[bytes], specials: list[tuple[int, int]] = (), speculate: int | bool = True, matcher: str = "both") -> list[int]:
        """Piece starts (byte offsets into the packed batch) of the documents -- gap chars included, listed in self.gaps as well;
        specials: (offset, length) of allowed special tokens.
        speculate: False = the matcher alone walks every document; True / 1 = with the speculative pass over 256-byte segments; 2 = 1 KiB;
        + 4: with the link pass (what the device runs); + 8: documents resolved by groups of 64 lanes (the device's wavefront).
        matcher: "program" = the backtracking program; "dfa" = the pattern's DFA (self.dfa must not be None); "both" = the program, and where
        the pattern has a DFA the same split through it as well, which has to give the same starts and gaps (self.stats are the program's)."""
        if matcher == "dfa":
            assert self.dfa, self.dfa_why
            got = self._split(docs, specials, int(speculate) | 16)
            gaps = self.gaps
            if int(speculate) & 3:  # (the one-loop form of the speculative lanes, which is what the device runs)
                assert self._split(docs, specials, int(speculate) | 48) == got and self.gaps == gaps, "the two forms of the DFA's speculative pass disagree"
                # (a speculative match may look 64 bytes beyond its segment instead of 2 KiB: many more pieces are left to the resolving pass)
                assert self._split(docs, specials, int(speculate) | 48 | (6 << 8)) == got and self.gaps == gaps, "the split depends on the look-ahead limit"
            else:  # (no speculation, documents by groups of lanes: EVERY piece is matched by the group together -- tk_rx_match_dfa_coop)
                assert self._split(docs, specials, 8 | 16) == got and self.gaps == gaps, "the group's matcher and the lane's disagree"
            return got
        got = self._split(docs, specials, int(speculate))
        if matcher == "both" and self.dfa:
            gaps, stats = self.gaps, self.stats
            assert self.split(docs, specials, speculate, matcher="dfa") == got and self.gaps == gaps, "the DFA and the program disagree"
            self.stats = stats
        return got

    def _split(self, docs, specials, speculate: int) -> list[int]:
        blob, off = pack(docs)
        n = len(blob)
        starts = np.zeros(n + 1, np.uint8)
        sa = np.array([a for a, _ in specials], np.uint64)
        sl = np.array([b for _, b in specials], np.uint64)
        stats = np.zeros(2, np.uint64)
        buf = np.ascontiguousarray(blob) if n else np.zeros(1, np.uint8)
        rc = self._L.tks_rx_split(self._h, buf.ctypes.data, n, off.ctypes.data, len(off) - 1, sa.ctypes.data if len(sa) else None,
                                  sl.ctypes.data if len(sl) else None, len(sa), speculate, starts.ctypes.data, stats.ctypes.data)
        self.stats = (int(stats[0]), int(stats[1]))
        if rc:
            raise RuntimeError(f"split error {rc & 255} at byte {rc >> 8}")
        self.gaps = np.flatnonzero(starts[:n] & 2).tolist()  # the starts that are gap chars (the pattern matches nothing there)
        return np.flatnonzero(starts[:n]).tolist()
