"""Host-side mirror of the reference interface: names, signatures, the tiktoken_ext plugin surface, the vocabulary wire format.  (Anything that constructs a CoreBPE needs a GPU and lives in test_gpu_api.py.)"""
import base64
import hashlib
import inspect
import os

import numpy as np
import pytest

import helpers as h
import tiktoken_amd
from tiktoken_amd import vocab_io
from tiktoken_amd.core import Encoding

REFERENCE_ENCODING_API = {  # reference tiktoken/core.py:16-428
    "encode_ordinary": ["self", "text"],
    "encode": ["self", "text", "allowed_special", "disallowed_special"],
    "encode_to_numpy": ["self", "text", "allowed_special", "disallowed_special"],
    "encode_ordinary_batch": ["self", "text", "num_threads"],
    "encode_batch": ["self", "text", "num_threads", "allowed_special", "disallowed_special"],
    "encode_with_unstable": ["self", "text", "allowed_special", "disallowed_special"],
    "encode_single_token": ["self", "text_or_bytes"],
    "decode_bytes": ["self", "tokens"],
    "decode": ["self", "tokens", "errors"],
    "decode_single_token_bytes": ["self", "token"],
    "decode_tokens_bytes": ["self", "tokens"],
    "decode_with_offsets": ["self", "tokens"],
    "decode_batch": ["self", "batch", "errors", "num_threads"],
    "decode_bytes_batch": ["self", "batch", "num_threads"],
    "token_byte_values": ["self"],
    "is_special_token": ["self", "token"],
    "_encode_single_piece": ["self", "text_or_bytes"],
    "_encode_only_native_bpe": ["self", "text"],
    "_encode_bytes": ["self", "text"],
}
REFERENCE_COREBPE_METHODS = [  # reference src/py.rs:13-184
    "encode_ordinary", "encode", "encode_to_tiktoken_buffer", "_encode_bytes", "encode_with_unstable",
    "encode_single_token", "encode_single_piece", "decode_bytes", "decode_single_token_bytes", "token_byte_values",
]


def test_encoding_signatures_match_reference():
    for name, params in REFERENCE_ENCODING_API.items():
        sig = inspect.signature(getattr(Encoding, name))
        assert list(sig.parameters) == params, name
    init = inspect.signature(Encoding.__init__)
    assert list(init.parameters) == ["self", "name", "pat_str", "mergeable_ranks", "special_tokens", "explicit_n_vocab"]
    assert inspect.signature(Encoding.encode).parameters["disallowed_special"].default == "all"
    assert inspect.signature(Encoding.encode_ordinary_batch).parameters["num_threads"].default == 8
    for prop in ("eot_token", "n_vocab"):
        assert isinstance(getattr(Encoding, prop), property)
    for m in REFERENCE_COREBPE_METHODS:
        assert callable(getattr(tiktoken_amd.CoreBPE, m)), m


def test_registry_lists_stock_and_shaped_encodings():
    names = tiktoken_amd.list_encoding_names()
    for n in ["gpt2", "r50k_base", "p50k_base", "p50k_edit", "cl100k_base", "o200k_base", "o200k_harmony",
              "gpt2_shaped", "cl100k_shaped", "o200k_shaped", "o200k_custom8"]:
        assert n in names
    with pytest.raises(ValueError, match="Unknown encoding"):
        tiktoken_amd.get_encoding("no_such_encoding")
    with pytest.raises(ValueError):
        tiktoken_amd.get_encoding(123)


def test_stock_constructor_data():
    """Special-token ids and n_vocab of the stock encodings (openai_public.py:29,43,57,66,80-86,100,128-145)."""
    from tiktoken_ext import amd_shaped, openai_public as pub

    h9 = amd_shaped.o200k_custom8()
    assert h9["special_tokens"]["<|custom_0|>"] == 200019 and h9["special_tokens"]["<|custom_7|>"] == 200026
    assert len(amd_shaped.gpt2_shaped()["mergeable_ranks"]) == 50256
    assert len(amd_shaped.cl100k_shaped()["mergeable_ranks"]) == 100256
    assert len(amd_shaped.o200k_shaped()["mergeable_ranks"]) == 199998
    assert set(pub.ENCODING_CONSTRUCTORS) == {"gpt2", "r50k_base", "p50k_base", "p50k_edit", "cl100k_base", "o200k_base",
                                              "o200k_harmony"}


def test_tiktoken_file_roundtrip_and_cache(tmp_path, monkeypatch):
    """Wire format `base64(token) SP rank` (reference load.py:147-171) through the NATIVE parser (tk_parse_tiktoken_bpe), sha256 pinning,
    and the reference's cache layout (file name = sha1 of the URL, load.py:51) for URL locations."""
    ranks = {b"a": 0, b"b": 1, b"ab": 2, b"\xff\x00": 3, b"x" * 100: 4000000000}
    path = tmp_path / "tiny.tiktoken"
    vocab_io.dump_tiktoken_bpe(ranks, str(path))
    assert path.read_bytes().splitlines()[2] == base64.b64encode(b"ab") + b" 2"
    sha = hashlib.sha256(path.read_bytes()).hexdigest()
    got = vocab_io.load_tiktoken_bpe(str(path), expected_hash=sha)
    assert got == ranks and got.packed is not None and got.packed[2].tolist() == list(ranks.values())
    got[b"zz"] = 9
    assert got.packed is None  # a mutated table no longer vouches for its packed form
    with pytest.raises(ValueError, match="Hash mismatch"):
        vocab_io.load_tiktoken_bpe(str(path), expected_hash="0" * 64)
    for bad in (b"!!notbase64 x\n", b"YQ== notanumber\n", b"YQ==\n", b"YQ= 1\n", b"YQ== 4294967296\n"):
        with pytest.raises(ValueError, match="Error parsing line 1"):
            vocab_io.parse_tiktoken_bpe(bad)
    assert vocab_io.parse_tiktoken_bpe(b"") == {} and vocab_io.parse_tiktoken_bpe(b"\nYQ== 7\r\n\n") == {b"a": 7}
    # a URL is served from the cache directory under sha1(url) without touching the network
    url = "https://example.invalid/encodings/tiny.tiktoken"
    cache = tmp_path / "cache"
    cache.mkdir()
    (cache / hashlib.sha1(url.encode()).hexdigest()).write_bytes(path.read_bytes())
    monkeypatch.setenv("TIKTOKEN_CACHE_DIR", str(cache))
    assert vocab_io.load_tiktoken_bpe(url, expected_hash=sha) == ranks


def test_native_parser_is_as_lenient_as_the_references_python_calls():
    """reference load.py:162-171 reads a `.tiktoken` file with contents.splitlines(), line.split(), base64.b64decode(token) (which skips bytes
    outside the alphabet and ignores what follows the padding) and int(rank) (sign, underscores): a file it loads has to load here, with the same
    table, and what it refuses has to be refused here.  Differential, on hand-picked lines and 40 000 random ones; the one difference by
    design: a negative rank is an error at parse time here (the reference fails in CoreBPE's constructor: OverflowError)."""
    import random

    def ref(contents: bytes):
        ret = {}
        for line in contents.splitlines():
            if not line:
                continue
            token, rank = line.split()
            ret[base64.b64decode(token)] = int(rank)
        return ret

    cases = [b"YQ==\t7\n", b"YQ==  7\n", b" YQ== 7\n", b"YQ== 7 \n", b"YQ== +7\n", b"YQ== 007\n", b"YQ== 1_0\n", b"YQ== 1__0\n", b"YQ== _1\n", b"YQ== 1_\n",
             b"YQ== 7\rYg== 8\n", b"YQ== 7\r\nYg== 8\r\n", b"YQ== 7\x0bYg== 8\n", b"YQ== 7\x0c\n", b"Y!Q== 7\n", b"YQ==YQ== 7\n", b"YQ 7\n", b"YQ= 7\n",
             b"YQ=== 7\n", b"= 7\n", b"YWI 7\n", b"YWI= 7\n", b"YWJj 7\n", b"Y 7\n", b"YQ== 7\n\x00", b"YQ== 7\nYQ== 8\n", b"\xef\xbb\xbfYQ== 7\n", b"YQ== 4294967295\n",
             b"YQ== 0x10\n", b"YQ== 7.0\n", b"YQ== 1e3\n", b"YQ==\x1c7\n", b"YQ==\xa07\n", b"YQ== 7\x85", b"Y Q== 7\n", b"YQ==\n7\n", b"-_-_ 7\n", b"+/+/ 7\n",
             b"YQ== -0\n", b"YQ== +\n", b"YQ== 7 8\n", b"\r\r\n\n", b"YQ== 7"]
    rng = random.Random(3)
    alphabet = b"YQWJj=+/ \t\r\n0123456789_!x"
    cases += [bytes(rng.choice(alphabet) for _ in range(rng.randint(0, 14))) for _ in range(40000)]
    both_ok = 0
    for c in cases:
        try:
            want = ref(c)
        except Exception:
            want = None
        try:
            got = dict(vocab_io.parse_tiktoken_bpe(c))
        except ValueError as e:
            assert "Error parsing line" in str(e)
            got = None
        assert got == want, c
        both_ok += want is not None
    assert both_ok > 2000
    for c in (b"YQ== -1\n", b"YQ== 4294967296\n", b"YQ== 99999999999999999999999\n"):
        with pytest.raises(ValueError, match="Error parsing line 1"):
            vocab_io.parse_tiktoken_bpe(c)


def test_native_parser_equals_python_on_the_shipped_vocabularies():
    import gzip

    for name in h.ENCODING_NAMES:
        raw = gzip.open(os.path.join(h.ROOT, "tiktoken_amd", "vocab", name + ".tiktoken.gz")).read()
        want = {base64.b64decode(t): int(r) for t, r in (line.split() for line in raw.splitlines() if line)}
        assert vocab_io.parse_tiktoken_bpe(raw) == want


def test_shaped_vocab_files_parse_like_reference_format():
    ranks = h.load_vocab("gpt2_shaped")
    from tiktoken_ext import amd_shaped

    assert amd_shaped.gpt2_shaped()["mergeable_ranks"] == ranks
    assert sorted(ranks.values()) == list(range(50256))
    order = vocab_io.data_gym_byte_order()
    assert [ranks[bytes([b])] for b in order] == list(range(256))


def test_data_gym_conversion_equals_the_references(tmp_path, monkeypatch):
    """vocab.bpe + encoder.json -> ranks (reference load.py:84-144), on a synthetic vocabulary: against the reference's own function where its
    source tree is at hand (this container), and against the table the files were made from everywhere."""
    import importlib.util
    import json
    import random

    rng = random.Random(7)
    order = vocab_io.data_gym_byte_order()
    to_char = {b: chr(b) for b in order[:188]}
    to_char.update({b: chr(256 + i) for i, b in enumerate(order[188:])})
    assert len([b for b in range(256) if chr(b).isprintable() and chr(b) != " "]) == 188

    def gym(tok: bytes) -> str:
        return "".join(to_char[b] for b in tok)

    toks = [bytes([b]) for b in order]
    ranks = {t: i for i, t in enumerate(toks)}
    lines = ["#version: 0.2"]
    while len(ranks) < 900:
        a, b = rng.choice(toks), rng.choice(toks)
        if a + b in ranks or len(a + b) > 12:
            continue
        ranks[a + b] = len(ranks)
        toks.append(a + b)
        lines.append(gym(a) + " " + gym(b))
    bpe, enc = tmp_path / "vocab.bpe", tmp_path / "encoder.json"
    bpe.write_text("\n".join(lines) + "\n", encoding="utf-8")
    table = {gym(t): r for t, r in ranks.items()}
    table["<|endoftext|>"] = len(ranks)
    enc.write_text(json.dumps(table), encoding="utf-8")
    monkeypatch.setenv("TIKTOKEN_CACHE_DIR", "")  # (no cache: local files)
    got = vocab_io.data_gym_to_mergeable_bpe_ranks(str(bpe), str(enc))
    assert got == ranks
    assert vocab_io.data_gym_to_mergeable_bpe_ranks(str(bpe), str(enc), clobber_one_byte_tokens=True) == ranks
    # a merge listed twice takes a number both times (the later ones move up by one): encoder.json numbered that way is accepted, as in the reference
    dup = lines[:400] + [lines[300]] + lines[400:]
    bpe2, enc2 = tmp_path / "vocab2.bpe", tmp_path / "encoder2.json"
    bpe2.write_text("\n".join(dup) + "\n", encoding="utf-8")
    first, second = lines[300].split()
    back = {v: k for k, v in to_char.items()}
    dup_tok = bytes(back[c] for c in first) + bytes(back[c] for c in second)
    ranks2 = {t: (r if r < 256 + 399 else r + 1) for t, r in ranks.items()}
    ranks2[dup_tok] = 256 + 399
    enc2.write_text(json.dumps({gym(t): r for t, r in ranks2.items()}), encoding="utf-8")
    assert vocab_io.data_gym_to_mergeable_bpe_ranks(str(bpe2), str(enc2)) == ranks2
    with pytest.raises(AssertionError):
        vocab_io.data_gym_to_mergeable_bpe_ranks(str(bpe2), str(enc))  # (files that do not belong together)
    ref_path = "/root/reference/tiktoken/load.py"
    if os.path.exists(ref_path):
        spec = importlib.util.spec_from_file_location("reference_load", ref_path)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        assert ref.data_gym_to_mergeable_bpe_ranks(str(bpe), str(enc)) == got
        assert ref.data_gym_to_mergeable_bpe_ranks(str(bpe2), str(enc2)) == ranks2
        with pytest.raises(AssertionError):
            ref.data_gym_to_mergeable_bpe_ranks(str(bpe2), str(enc))
        # and the .tiktoken wire format through the reference's loader, on a file written here
        path = tmp_path / "t.tiktoken"
        vocab_io.dump_tiktoken_bpe(ranks, str(path))
        assert ref.load_tiktoken_bpe(str(path)) == ranks == vocab_io.load_tiktoken_bpe(str(path))


def test_model_table():
    """reference tiktoken/model.py:88-105 and tests/test_misc.py: exact names, dated versions through prefixes, the longest prefix decides."""
    f = tiktoken_amd.encoding_name_for_model
    assert f("gpt2") == "gpt2" and f("text-davinci-003") == "p50k_base" and f("text-davinci-edit-001") == "p50k_edit"
    assert f("gpt-3.5-turbo-0301") == "cl100k_base" and f("gpt-4") == "cl100k_base" and f("gpt-4-32k") == "cl100k_base"
    assert f("gpt-4o") == "o200k_base" and f("gpt-4o-2024-05-13") == "o200k_base" and f("gpt-oss-120b") == "o200k_harmony"
    assert f("ft:gpt-4o:org") == "o200k_base" and f("ft:gpt-4:org") == "cl100k_base" and f("gpt-5-mini") == "o200k_base"
    with pytest.raises(KeyError, match="Could not automatically map"):
        f("llama-3")
    assert callable(tiktoken_amd.encoding_for_model)


def test_rank_table_fills_itself_on_first_use():
    """vocab_io.RankTable: with lazy=True the parsed file stays in packed arrays (what tk_create takes) until somebody reads the dict;
    the public default is a filled dict (C code that reads a dict's storage sees the real contents)."""
    src = {bytes([i]): i for i in range(256)}
    src.update({b"ab": 256, b"abc": 257, b"\xe4\xb8\xad": 258})
    text = b"".join(base64.b64encode(k) + b" %d\n" % v for k, v in src.items())
    t = vocab_io.parse_tiktoken_bpe(text, lazy=True)
    assert t._pending is not None and t.max_rank() == 258 and t._pending is not None  # nothing walked the dict
    assert t.packed is not None and len(t.packed[2]) == len(src)
    assert len(t) == len(src) and t._pending is None  # the first use filled it
    assert t == src and src == t and dict(t) == src and t[b"abc"] == 257 and b"zz" not in t and t.get(b"zz", 7) == 7
    for fresh_use in (lambda x: x[b"ab"], lambda x: b"ab" in x, lambda x: list(x)[0], lambda x: {**x}, lambda x: x.items(), lambda x: repr(x), lambda x: x.copy()):
        u = vocab_io.parse_tiktoken_bpe(text, lazy=True)
        fresh_use(u)
        assert u._pending is None and dict.__len__(u) == len(src)
    u = vocab_io.parse_tiktoken_bpe(text, lazy=True)
    u[b"new"] = 300
    assert u.packed is None and len(u) == len(src) + 1 and u.max_rank() == 300
    import pickle

    assert pickle.loads(pickle.dumps(vocab_io.parse_tiktoken_bpe(text, lazy=True))) == src
    dup = vocab_io.parse_tiktoken_bpe(b"YQ== 0\nYg== 1\nYQ== 2\n", lazy=True)  # a token listed twice: the dict keeps the later rank
    assert dict(dup) == {b"a": 2, b"b": 1} and dup.packed is None
    # the default: filled at once; two tables that are both still lazy compare by contents
    e = vocab_io.parse_tiktoken_bpe(text)
    assert e._pending is None and dict.__len__(e) == len(src) and e.packed is not None
    assert vocab_io.parse_tiktoken_bpe(text, lazy=True) == vocab_io.parse_tiktoken_bpe(text, lazy=True)
    assert not (vocab_io.parse_tiktoken_bpe(text, lazy=True) != vocab_io.parse_tiktoken_bpe(text, lazy=True))
    assert (vocab_io.parse_tiktoken_bpe(text, lazy=True) | vocab_io.parse_tiktoken_bpe(b"eno= 999\n", lazy=True))[b"zz"] == 999
    d5 = vocab_io.parse_tiktoken_bpe(b"YQ== 0\nYg== 5\nYg== 3\n")  # the largest rank belongs to a token that is listed again
    assert d5.max_rank() == 3 and dict(d5) == {b"a": 0, b"b": 3}



def test_explicit_n_vocab_is_checked_before_the_core_is_built():
    """The reference asserts explicit_n_vocab first (core.py:96-101) and builds CoreBPE afterwards: an inconsistent value is an AssertionError
    whatever else is wrong -- here: no GPU in this process, so building the core would fail.  A lazily parsed vocabulary file is checked on
    its packed arrays, and on the dict when those disagree (a file may list a token twice).  `_mergeable_ranks` can be assigned, as the
    reference's plain attribute can."""
    ranks = {bytes([i]): i for i in range(256)}
    with pytest.raises(AssertionError):
        Encoding("x", pat_str=h.PAT_STR[0], mergeable_ranks=ranks, special_tokens={"<|e|>": 256}, explicit_n_vocab=300)
    with pytest.raises(AssertionError):
        Encoding("x", pat_str=h.PAT_STR[0], mergeable_ranks=ranks, special_tokens={"<|e|>": 300}, explicit_n_vocab=257)
    lines = b"".join(base64.b64encode(bytes([i])) + b" " + str(i).encode() + b"\n" for i in range(256))
    lazy = vocab_io.parse_tiktoken_bpe(lines, lazy=True)
    with pytest.raises(AssertionError):
        Encoding("x", pat_str=h.PAT_STR[0], mergeable_ranks=lazy, special_tokens={}, explicit_n_vocab=255)
    assert getattr(lazy, "_pending", None) is None or True  # (a failed check on the arrays is repeated on the dict: the table may have been filled)
    e = object.__new__(Encoding)
    e._mergeable_ranks = {b"a": 0}
    assert e._mergeable_ranks == {b"a": 0}
