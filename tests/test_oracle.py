"""The CPU oracle pinned against (a) the golden fixtures produced by the reference's own Python code
(tools/gen_golden.py: tiktoken/_educational.py bpe_encode + regex.findall), (b) Python `regex` directly,
(c) the vocabulary-free vectors of the reference's Rust unit tests, (d) a third-party implementation: HF `tokenizers` built from the same
vocabulary files."""
import random

import numpy as np
import pytest

import helpers as h
from oracle import c_oracle, py_oracle as po

ALL = h.ENCODING_NAMES + ["edu600"]


@pytest.mark.parametrize("name", ALL)
def test_c_oracle_matches_reference_golden(name):
    g = h.load_golden(name)
    C = h.c_oracle_for(name)
    for c in g["cases"]:
        if c["allowed"] is None:
            got = C.encode_ordinary(c["text"]).tolist()
        else:
            got = C.encode(c["text"], set(c["allowed"])).tolist()
        assert got == c["tokens"], (c["name"], c["text"][:80])


@pytest.mark.parametrize("name", ALL)
def test_py_oracle_matches_reference_golden(name):
    g = h.load_golden(name)
    ranks = h.golden_vocab(name)
    for c in g["cases"][::3]:
        text = c["text"].decode("utf-8")
        if c["allowed"] is None:
            got = po.encode_ordinary(text, g["pat_str"], ranks)
        else:
            got = po.encode(text, g["pat_str"], ranks, g["special_tokens"], set(c["allowed"]))
        assert got == c["tokens"], c["name"]


def test_vocab_free_rust_unit_vectors():
    """src/lib.rs:685-701: ranks {ab:0, cd:1}; byte_pair_split(abcd) = [ab, cd], (abab) = [ab, ab]."""
    ranks = {b"ab": 0, b"cd": 1}
    parts = po.byte_pair_merge(ranks, b"abcd")
    assert [b"abcd"[a:b] for a, b in zip(parts[:-1], parts[1:])] == [b"ab", b"cd"]
    parts = po.byte_pair_merge(ranks, b"abab")
    assert [b"abab"[a:b] for a, b in zip(parts[:-1], parts[1:])] == [b"ab", b"ab"]
    full = dict(ranks)
    for b in range(256):
        full[bytes([b])] = 2 + b
    C = c_oracle.COracle(0, full, {})
    assert C.encode_piece(b"abcd") == [0, 1]
    assert C.encode_piece(b"abab") == [0, 0]


@pytest.mark.parametrize("pat", [0, 1, 2])
def test_scanners_equal_regex_findall(pat):
    """Both scanner restatements (Python, C) against regex.findall(pat_str) -- the split the reference
    sanctions at core.py:395-404 -- on adversarial strings."""
    rng = random.Random(100 + pat)
    C = c_oracle.COracle(pat, {bytes([b]): b for b in range(256)}, {})
    ps = h.PAT_STR[pat]
    for _ in range(4000):
        s = "".join(rng.choice(h.ADV[:-3]) for _ in range(rng.randint(0, 24)))
        ref = po.split_regex(ps, s)
        assert po.split_scan(pat, s) == ref, repr(s)
        ends, acc = [], 0
        for piece in ref:
            acc += len(piece.encode())
            ends.append(acc)
        assert C.split(s.encode()) == ends, repr(s)


def test_gpt2_original_pattern_is_equivalent():
    """openai_public.py:9-14 declares the possessive rewrite equivalent to the original GPT-2 regex."""
    rng = random.Random(7)
    for _ in range(3000):
        s = "".join(rng.choice(h.ADV[:-3]) for _ in range(rng.randint(0, 16)))
        assert po.split_regex(po.GPT2_ORIG_PAT, s) == po.split_regex(po.R50K_PAT, s)


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_heap_and_linear_merge_agree(name):
    """_byte_pair_merge (lib.rs:140-196) == _byte_pair_merge_large (lib.rs:47-138), on both sides of the
    100-byte dispatch of lib.rs:204."""
    ranks = h.load_vocab(name)
    rng = random.Random(3)
    C = h.c_oracle_for(name)
    for _ in range(150):
        n = rng.randint(2, 260)
        piece = "".join(rng.choice("abcdehilnorst 中") for _ in range(n)).encode()
        lin = po.byte_pair_merge(ranks, piece)
        lin = [ranks[piece[a:b]] for a, b in zip(lin[:-1], lin[1:])]
        assert po.byte_pair_merge_large(ranks, piece) == lin
        if piece not in ranks:
            assert C.encode_piece(piece) == lin


@pytest.mark.parametrize("name", h.ENCODING_NAMES)
def test_every_sampled_token_reencodes_to_itself(name):
    ranks = h.load_vocab(name)
    C = h.c_oracle_for(name)
    items = list(ranks.items())
    for tb, r in items[::211]:
        assert C.encode_piece(tb) == [r]


def test_batch_threads_equal_sequential():
    C = h.c_oracle_for("cl100k_shaped")
    blob, off = h.gen_corpus(99, 0, 1 << 20)
    t1, o1 = C.encode_batch(blob, off, None, 1)
    t8, o8 = C.encode_batch(blob, off, None, 8)
    assert np.array_equal(t1, t8) and np.array_equal(o1, o8)
    d = 5
    a, b = int(off[d]), int(off[d + 1])
    assert np.array_equal(t1[int(o1[d]):int(o1[d + 1])], C.encode_ordinary(blob[a:b].tobytes()))


def test_special_token_slices_are_independent_haystacks():
    """lib.rs:402-405: the text before an allowed special is its own haystack, so trailing-whitespace
    rules see end-of-text there."""
    C = h.c_oracle_for("cl100k_shaped")
    a = C.encode(b"x  <|endoftext|>", {"<|endoftext|>"}).tolist()
    assert a == C.encode_ordinary(b"x  ").tolist() + [100257]
    assert C.encode(b"x  <|endoftext|>", set()).tolist() == C.encode_ordinary(b"x  <|endoftext|>").tolist()


@pytest.mark.parametrize("name,mix", [("gpt2_shaped", 2), ("cl100k_shaped", 0), ("o200k_shaped", 1)])
def test_oracle_equals_hf_tokenizers(name, mix, tmp_path, monkeypatch):
    """A third-party implementation as a witness: HF `tokenizers` (its own Rust BPE, Oniguruma for the split), built from the same
    `.tiktoken` vocabulary by transformers' TikTokenConverter (which reconstructs the merge list from the ranks), must give the ids the C
    oracle gives -- on corpus documents and on the fuzzer's awkward ones.  (SURVEY 8(d) lists HF tokenizers as context for the baseline;
    here it pins the oracle, for o200k too, where the reference's own tests hold no known answers.)  The pattern is handed to Oniguruma in
    a spelling that means the same there: no possessive marker behind a counted repeat ({1,3}+ repeats the repeat in Ruby syntax), \\z for
    the end of the text ($ is end-of-line there)."""
    import base64
    import gzip
    import os
    import random
    import sys
    import types

    tokenizers = pytest.importorskip("tokenizers")
    convert = pytest.importorskip("transformers.convert_slow_tokenizer")
    from test_patterns import CL100K_PLAIN

    def load_tiktoken_bpe(path, expected_hash=None):  # the .tiktoken text format (reference tiktoken/load.py:159-171); the converter asks for it
        return {base64.b64decode(t): int(r) for t, r in (line.split() for line in open(path, "rb").read().splitlines() if line)}

    shim, shim_load = types.ModuleType("tiktoken"), types.ModuleType("tiktoken.load")
    shim_load.load_tiktoken_bpe = load_tiktoken_bpe
    shim.load = shim_load
    monkeypatch.setitem(sys.modules, "tiktoken", shim)
    monkeypatch.setitem(sys.modules, "tiktoken.load", shim_load)
    vocab = tmp_path / (name + ".tiktoken")
    vocab.write_bytes(gzip.open(os.path.join(h.ROOT, "tiktoken_amd", "vocab", name + ".tiktoken.gz")).read())
    pat = CL100K_PLAIN.replace("$", r"\z") if name == "cl100k_shaped" else h.load_golden(name)["pat_str"]
    hf = convert.TikTokenConverter(vocab_file=str(vocab), pattern=pat, add_prefix_space=False).converted()
    C = h.c_oracle_for(name)
    rng = random.Random(1)
    texts = ["hello world", "The quick brown fox's 12345 jumps\n\n  over\tthe lazy dog. 中文 😀", "DON'T stop", "x \n ", "a  \n\n  b  "]
    texts += [h.fuzz_doc(rng)[:3000] for _ in range(40)]
    blob, off = h.gen_corpus(0x5EED0300 + mix, mix, 1 << 20)
    bb = blob.tobytes()
    texts += [bb[int(off[d]):int(off[d + 1])].decode() for d in range(min(200, len(off) - 1))]
    n = 0
    for text in texts:
        want = C.encode_ordinary(text.encode()).tolist()
        assert hf.encode(text, add_special_tokens=False).ids == want, text[:80]
        n += len(want)
    assert n > 150_000
    # special tokens: registered with HF as added tokens (its own ids for them: mapped back), against encode(..., allowed_special="all")
    specials = h.load_golden(name)["special_tokens"]
    hf.add_special_tokens([tokenizers.AddedToken(s, normalized=False, special=True) for s in specials])
    back = {hf.token_to_id(s): i for s, i in specials.items()}
    sp = list(specials)
    for _ in range(300):
        text = "".join(rng.choice(h.ADV + sp + sp + ["<|endoftext", "|>", "<|"]) for _ in range(rng.randint(1, 30)))
        assert [back.get(i, i) for i in hf.encode(text, add_special_tokens=False).ids] == C.encode(text.encode(), "all").tolist(), text
