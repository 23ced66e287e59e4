"""The UNMODIFIED reference Python package (`/root/reference/tiktoken`) driven over this repo's shim.

The reference binds its native core in two places only: `from tiktoken import _tiktoken` (tiktoken/core.py:7) and
`_tiktoken.CoreBPE(mergeable_ranks, special_tokens, pat_str)` (core.py:57).  INTEGRATION.md section A says that putting
`tiktoken_amd._tiktoken` in that module's place is the whole integration; the tests below do exactly that -- the module goes into
`sys.modules["tiktoken._tiktoken"]`, the package is imported from `/root/reference`, and the reference's OWN `Encoding`, thread pools,
`np.frombuffer`, pickle and registry code run on top of the HIP library.

Where they can run.  The reference's sources exist only in the build container (`/root/reference`; they may not travel to the GPU box in
any form) and the GPU exists only on the GPU box, so:
  * the CPU tests (this container): the package imports over the shim, every `self._core_bpe.<method>(...)` call site of the reference's
    core.py binds to a method of the shim with that arity, construction without a device fails loudly (no CPU fallback), and the call-site
    list in tests/golden/reference_call_sites.json (made by tools/gen_call_sites.py from the reference's AST: line, method, arity -- data,
    not source) is current;
  * `test_reference_package_over_shim_*` (`-m gpu`, skipped when `/root/reference` is absent): the full check -- needs a box with both;
  * `test_shim_driven_as_core_py_drives_it` (`-m gpu`, travels): every call site of that list driven through the shim in the form
    core.py uses (same positional arguments, `np.frombuffer` on the buffer, eight pool threads hammering `encode`), against the oracle.
"""
import ast
import functools
import importlib
import inspect
import json
import os
import pickle
import sys
from concurrent.futures import ThreadPoolExecutor

import hypothesis
import hypothesis.strategies as st
import numpy as np
import pytest

import helpers as h

REF = "/root/reference"
HAVE_REF = os.path.isdir(os.path.join(REF, "tiktoken"))
CALL_SITES = os.path.join(h.ROOT, "tests", "golden", "reference_call_sites.json")
MAX_EXAMPLES = int(os.environ.get("TIKTOKEN_MAX_EXAMPLES", "40"))
needs_ref = pytest.mark.skipif(not HAVE_REF, reason="/root/reference (the reference's Python package) is not on this box")


def reference_package_over_shim():
    """The reference's `tiktoken` package with `tiktoken_amd._tiktoken` bound as its `_tiktoken` (core.py:7)."""
    from tiktoken_amd import _tiktoken as shim

    mod = sys.modules.get("tiktoken")
    if mod is not None and getattr(sys.modules.get("tiktoken._tiktoken"), "CoreBPE", None) is shim.CoreBPE \
            and (mod.__file__ or "").startswith(REF):
        return mod
    for name in [m for m in sys.modules if m == "tiktoken" or m.startswith("tiktoken.")]:
        del sys.modules[name]
    sys.modules["tiktoken._tiktoken"] = shim  # <- the integration: the three lines of INTEGRATION.md section A amount to this
    if REF not in sys.path:
        sys.path.append(REF)  # (behind the repo root: `tiktoken_ext` resolves to this repo's plugins first, the reference's own second)
    ref = importlib.import_module("tiktoken")
    assert ref.__file__.startswith(REF), ref.__file__
    assert ref.core._tiktoken is shim
    return ref


def core_bpe_call_sites(path=os.path.join(REF, "tiktoken", "core.py")):
    """[(line, method, n_positional, [keywords])] of every `self._core_bpe.<method>(...)` and `_tiktoken.CoreBPE(...)` in core.py."""
    tree = ast.parse(open(path).read())
    out = []
    for node in ast.walk(tree):
        if not isinstance(node, ast.Call) or not isinstance(node.func, ast.Attribute):
            continue
        f = node.func
        if isinstance(f.value, ast.Attribute) and f.value.attr == "_core_bpe":
            out.append([node.lineno, f.attr, len(node.args), sorted(k.arg for k in node.keywords)])
        elif isinstance(f.value, ast.Name) and f.value.id == "_tiktoken" and f.attr == "CoreBPE":
            out.append([node.lineno, "__init__", len(node.args), sorted(k.arg for k in node.keywords)])
    return sorted(out)


# ---------------------------------------------------------------------------------------------------------------- CPU (build container)
@needs_ref
def test_call_site_fixture_is_current():
    with open(CALL_SITES) as f:
        fixture = json.load(f)
    assert fixture["call_sites"] == core_bpe_call_sites(), "re-run tools/gen_call_sites.py"


def test_every_call_site_binds_to_the_shim():
    """Each call form of the reference's core.py is accepted by the shim's method of that name (arity and keywords)."""
    from tiktoken_amd import _tiktoken as shim

    with open(CALL_SITES) as f:
        sites = json.load(f)["call_sites"]
    assert len(sites) >= 15 and {m for _, m, _, _ in sites} >= {"encode", "encode_ordinary", "encode_to_tiktoken_buffer", "_encode_bytes"}
    for line, method, n_pos, kws in sites:
        fn = getattr(shim.CoreBPE, method, None)
        assert fn is not None, f"core.py:{line} calls CoreBPE.{method}, which the shim lacks"
        inspect.signature(fn).bind(None, *([None] * n_pos), **{k: None for k in kws})  # raises TypeError on a mismatch
    # the eleven methods of src/py.rs:13-184 (+ the constructor) are all there, public or private as the reference spells them
    for m in ("encode_ordinary", "encode", "encode_to_tiktoken_buffer", "_encode_bytes", "encode_with_unstable", "encode_single_token",
              "encode_single_piece", "decode_bytes", "decode_single_token_bytes", "token_byte_values"):
        assert callable(getattr(shim.CoreBPE, m))


@needs_ref
def test_reference_package_imports_over_the_shim_and_fails_loudly_without_a_device(have_gpu):
    ref = reference_package_over_shim()
    from tiktoken_amd import _tiktoken as shim

    assert ref.core._tiktoken.CoreBPE is shim.CoreBPE and ref.Encoding.__module__ == "tiktoken.core"
    # the reference's registry finds this repo's plugin modules through the `tiktoken_ext` namespace package (registry.py:33-60)
    names = ref.list_encoding_names()
    assert {"gpt2", "cl100k_base", "o200k_base", "o200k_harmony", "o200k_shaped", "cl100k_shaped", "gpt2_shaped"} <= set(names)
    assert ref.encoding_name_for_model("gpt-4o") == "o200k_base"
    if not have_gpu:
        g = h.load_golden("gpt2_shaped")
        with pytest.raises(RuntimeError, match="HIP device"):  # no CPU path to fall back to
            ref.Encoding("x", pat_str=g["pat_str"], mergeable_ranks=h.golden_vocab("gpt2_shaped"), special_tokens=g["special_tokens"])


# ---------------------------------------------------------------------------------------------------------------- GPU + reference
REF_ENCS = ["gpt2_shaped", "cl100k_shaped", "o200k_shaped"]
TEXTS = ["hello world", "", "hello <|endoftext|>", "The quick brown fox's 12345 jumps\n\n  over\tthe lazy dog. 中文テキスト 😀", "x" * 3000,
         "DON'T STOP  \r\n believing ...", "today\n \n", " \x850", "நடிகர் சூர்யா", " Ġ除", "a" * 70 + " " * 70 + "\n" * 9, "0" * 17]


@functools.lru_cache(maxsize=None)
def ref_encoding(name):
    """The reference's own `tiktoken.Encoding`, built by its own constructor, on the shaped vocabulary (a plain dict, as load.py returns)."""
    ref = reference_package_over_shim()
    g = h.load_golden(name)
    enc = ref.Encoding(name + "_ref", pat_str=g["pat_str"], mergeable_ranks=dict(h.golden_vocab(name)), special_tokens=dict(g["special_tokens"]))
    assert type(enc).__module__ == "tiktoken.core" and type(enc._core_bpe).__module__ == "tiktoken_amd._tiktoken"
    return enc


def oracle_encode(name, text, allowed=None):
    C = h.c_oracle_for(name)
    b = text.encode("utf-8")
    return (C.encode_ordinary(b) if allowed is None else C.encode(b, allowed)).tolist()


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", REF_ENCS)
def test_reference_package_over_shim_encode_paths(name):
    enc = ref_encoding(name)
    for t in TEXTS:
        want = oracle_encode(name, t)
        assert enc.encode_ordinary(t) == want, t[:30]
        assert enc.encode(t, disallowed_special=()) == want
        assert enc.encode(t, allowed_special="all") == oracle_encode(name, t, "all")
        assert enc.encode_to_numpy(t, disallowed_special=()).tolist() == want  # core.py:161: np.frombuffer on the shim's buffer
        assert enc.encode_to_numpy(t, disallowed_special=()).dtype == np.uint32
        assert enc.decode(want) == t and enc.decode_bytes(want) == t.encode()
        assert enc._encode_only_native_bpe(t) == want  # core.py:395-404: Python `regex` split + encode_single_piece per piece
        assert enc._encode_bytes(t.encode()) == want
    with pytest.raises(ValueError, match="disallowed special token"):
        enc.encode("hello <|endoftext|>")
    # surrogates: core.py:77-80,128-136 repair path over the shim's UnicodeEncodeError
    assert enc.encode("👍") == enc.encode("👍") == oracle_encode(name, "👍")
    assert enc.encode("\ud83d") == enc.encode("�") and enc.encode_ordinary("\ud83d") == enc.encode_ordinary("�")
    # batches through the reference's ThreadPoolExecutor (core.py:174-176, 202-206): eight threads in the shim at once
    docs = [TEXTS[i % len(TEXTS)] + str(i) for i in range(256)]
    want = [oracle_encode(name, d) for d in docs]
    assert enc.encode_ordinary_batch(docs, num_threads=8) == want
    assert enc.encode_batch(docs, num_threads=8, disallowed_special=()) == want
    assert enc.encode_batch(docs, num_threads=8, allowed_special="all") == [oracle_encode(name, d, "all") for d in docs]
    assert enc.decode_batch(want, num_threads=8) == docs
    assert enc.decode_bytes_batch(want, num_threads=8) == [d.encode() for d in docs]


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", REF_ENCS)
def test_reference_package_over_shim_single_tokens_offsets_unstable_pickle(name):
    from oracle import py_oracle as po

    enc = ref_encoding(name)
    ranks, specials = h.golden_vocab(name), h.load_golden(name)["special_tokens"]
    # tests/test_encoding.py:25-28,158-167
    for token in list(range(0, min(10_000, enc.max_token_value - 1), 41)) + list(specials.values()):
        assert enc.encode_single_token(enc.decode_single_token_bytes(token)) == token
    with pytest.raises(KeyError):
        enc.encode_single_token(b"\xff\xfe\xfd definitely not a token")
    with pytest.raises(KeyError):
        enc.decode_single_token_bytes(enc.max_token_value + 12345)
    assert enc.token_byte_values() == sorted(ranks)  # core.py:358 over lib.rs:648-650
    assert enc.eot_token == specials["<|endoftext|>"] and enc.n_vocab == enc.max_token_value + 1
    assert enc._encode_single_piece("helloqqqq") == po.encode_single_piece(b"helloqqqq", ranks)
    assert enc._encode_bytes(b" \xec\x8b\xa4\xed") == po.encode_bytes(b" \xec\x8b\xa4\xed", h.PAT_STR[h.PATTERN_OF[name]], ranks)
    for i in range(10):
        assert enc.decode_bytes(enc._encode_bytes(b"\x80" * i)) == b"\x80" * i
    # decode_with_offsets (core.py:303-330) against the definition in tests/test_offsets.py:17-24
    for prompt in ("hello world", "hello world<|endoftext|> green cow", "我非常渴望与人工智能一起工作", "நடிகர் சூர்யா", " Ġ除"):
        tokens = enc.encode(prompt, allowed_special="all")
        text, offsets = enc.decode_with_offsets(tokens)
        assert text == prompt
        want = []
        for i in range(len(tokens)):
            prefix = enc.decode(tokens[:i], errors="ignore")
            k = 0
            while k < len(text) and k < len(prefix) and text[k] == prefix[k]:
                k += 1
            want.append(k)
        assert offsets == want, prompt
    # encode_with_unstable (core.py:243 over lib.rs:483-599) against the oracle's restatement
    pat = h.PAT_STR[h.PATTERN_OF[name]]
    for text in ("hello fanta", "hello wor", "a\n\n", "x  ", "hello <|endoftext|>", "", "naïve caf"):
        got_t, got_c = enc.encode_with_unstable(text, disallowed_special=())
        want_t, want_c = po.encode_unstable_native(text, pat, ranks, specials, set())
        assert got_t == want_t and {tuple(c) for c in got_c} == want_c, text
    # pickle (core.py:409-428): an unregistered encoding travels as its constructor arguments and is rebuilt over the shim
    enc2 = pickle.loads(pickle.dumps(enc))
    assert type(enc2).__module__ == "tiktoken.core" and enc2.encode("hello world") == enc.encode("hello world") == oracle_encode(name, "hello world")
    ref = reference_package_over_shim()
    custom = ref.Encoding(name="custom_enc", pat_str=enc._pat_str, mergeable_ranks=enc._mergeable_ranks, special_tokens={"<|pickle|>": 300_000})
    custom2 = pickle.loads(pickle.dumps(custom))  # tests/test_pickle.py:11-23
    assert custom.encode("<|pickle|>", allowed_special="all") == custom2.encode("<|pickle|>", allowed_special="all") == [300_000]


@needs_ref
@pytest.mark.gpu
def test_reference_package_over_shim_registry_and_plugins():
    """The reference's `get_encoding` (registry.py:63-88) finds this repo's `tiktoken_ext` plugin, builds ITS Encoding from the plugin's
    constructor arguments (a lazily parsed RankTable as `mergeable_ranks`) and pickles it by name."""
    ref = reference_package_over_shim()
    enc = ref.get_encoding("o200k_shaped")
    assert type(enc).__module__ == "tiktoken.core" and enc is ref.get_encoding("o200k_shaped")
    s = "The quick brown fox's 12345 jumps\n\n over the lazy dog. 中文 \U0001F600"
    assert enc.encode(s) == oracle_encode("o200k_shaped", s)
    assert pickle.loads(pickle.dumps(enc)) is enc  # registered: by name (core.py:411-413)
    c8 = ref.get_encoding("o200k_custom8")
    assert c8.encode("a<|custom_3|>b", allowed_special="all") == oracle_encode("o200k_shaped", "a") + [200022] + oracle_encode("o200k_shaped", "b")


@needs_ref
@pytest.mark.gpu
def test_reference_package_over_shim_special_token_matrix():
    """tests/test_encoding.py:175-223, ids of the shaped encoding."""
    enc = ref_encoding("cl100k_shaped")
    eot, fip, fim = (enc.encode_single_token(s) for s in ("<|endoftext|>", "<|fim_prefix|>", "<|fim_middle|>"))
    assert eot == enc.eot_token
    text = "<|endoftext|> hello <|fim_prefix|>"
    assert eot not in enc.encode(text, disallowed_special=())
    for kw in ({}, {"disallowed_special": "all"}, {"disallowed_special": {"<|endoftext|>"}}, {"disallowed_special": {"<|fim_prefix|>"}}):
        with pytest.raises(ValueError):
            enc.encode(text, **kw)
    text = "<|endoftext|> hello <|fim_prefix|> there <|fim_middle|>"
    for allowed, inside in (((), set()), ("all", {eot, fip, fim}), ({"<|fim_prefix|>"}, {fip}), ({"<|endoftext|>"}, {eot}), ({"<|fim_middle|>"}, {fim})):
        tokens = enc.encode(text, allowed_special=allowed if allowed else set(), disallowed_special=())
        assert {t for t in tokens if t in (eot, fip, fim)} == inside
        assert tokens == oracle_encode("cl100k_shaped", text, "all" if allowed == "all" else sorted(allowed))
    assert {eot, fip, fim} <= set(enc.encode(text, allowed_special="all", disallowed_special="all"))


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gpt2_shaped", "cl100k_shaped"])
def test_reference_package_over_shim_vocabulary_free_properties(name):
    """tests/test_encoding.py:94-167,226-264 re-pointed at the reference Encoding over the shim."""
    enc = ref_encoding(name)
    for c in ["^", "0", "a", "'s", " ", "\n"]:  # test_catastrophically_repetitive
        for big in (c * 10_000, " " + c * 10_000, " " + c * 10_000 + "\n"):
            toks = enc.encode(big)
            assert big == enc.decode(toks) and toks == oracle_encode(name, big)
    for value in ("hello", "hello ", "hello  ", " hello", " hello ", " hello  ", "hello world", "请考试我的软件！12345"):
        assert value == enc.decode(enc.encode(value)) == enc.decode(enc.encode_ordinary(value))
    t1, t2 = "hello world", "goodbye world"  # test_batch_encode
    assert enc.encode_batch([t1]) == [enc.encode(t1)] and enc.encode_batch([t1, t2]) == [enc.encode(t1), enc.encode(t2)]
    assert enc.encode_ordinary_batch([t1, t2]) == [enc.encode_ordinary(t1), enc.encode_ordinary(t2)]

    @hypothesis.given(text=st.text())
    @hypothesis.settings(deadline=None, max_examples=MAX_EXAMPLES)
    def roundtrip(text):
        toks = enc.encode(text, disallowed_special=())
        assert text == enc.decode(toks)  # test_hyp_roundtrip
        assert enc.encode_ordinary(text) == toks  # test_hyp_special_ordinary
        assert toks == oracle_encode(name, text.encode("utf-16", "surrogatepass").decode("utf-16", "replace"))

    @hypothesis.given(bytestring=st.binary())
    @hypothesis.settings(deadline=None, max_examples=MAX_EXAMPLES)
    def encode_bytes(bytestring):
        assert enc.decode_bytes(enc._encode_bytes(bytestring)) == bytestring  # test_hyp_encode_bytes

    @hypothesis.given(batch=st.lists(st.text()))
    @hypothesis.settings(deadline=None, max_examples=MAX_EXAMPLES // 2)
    def batch_roundtrip(batch):
        encoded = enc.encode_batch(batch, allowed_special="all")
        assert encoded == [enc.encode(t, allowed_special="all") for t in batch]
        assert enc.decode_batch(encoded) == batch  # test_hyp_batch_roundtrip

    roundtrip()
    encode_bytes()
    batch_roundtrip()


# ---------------------------------------------------------------------------------------------------------------- GPU, travels
@pytest.mark.gpu
@pytest.mark.parametrize("name", REF_ENCS)
def test_shim_driven_as_core_py_drives_it(name):
    """No reference source on the GPU box: the shim is driven in exactly the forms the reference's core.py uses (the committed call-site
    list says which: method, positional arity), and every listed method is exercised."""
    from oracle import py_oracle as po
    from tiktoken_amd import _tiktoken as shim

    with open(CALL_SITES) as f:
        sites = json.load(f)["call_sites"]
    g = h.load_golden(name)
    ranks, specials, pat = h.golden_vocab(name), g["special_tokens"], g["pat_str"]
    core = shim.CoreBPE(dict(ranks), dict(specials), pat)  # core.py:57: three positionals
    allowed_all = set(specials)
    t = TEXTS[3]
    want, want_sp = oracle_encode(name, t), oracle_encode(name, "hi <|endoftext|> there", "all")
    drive = {
        "__init__": lambda: core is not None,
        "encode_ordinary": lambda: core.encode_ordinary(t) == want,  # core.py:76,80
        "encode": lambda: core.encode(t, set()) == want and core.encode("hi <|endoftext|> there", allowed_all) == want_sp,  # core.py:127,136
        "encode_to_tiktoken_buffer": lambda: np.frombuffer(core.encode_to_tiktoken_buffer(t, set()), dtype=np.uint32).tolist() == want,  # :161-162
        "encode_with_unstable": lambda: (lambda r: r[0] == po.encode_unstable_native("hello wor", pat, ranks, specials, set())[0])(
            core.encode_with_unstable("hello wor", set())),  # core.py:243
        "encode_single_token": lambda: core.encode_single_token(b"a") == ranks[b"a"],  # core.py:259
        "decode_bytes": lambda: core.decode_bytes(want) == t.encode(),  # core.py:273,287
        "decode_single_token_bytes": lambda: core.decode_single_token_bytes(ranks[b"a"]) == b"a",  # core.py:301
        "token_byte_values": lambda: core.token_byte_values() == sorted(ranks),  # core.py:358
        "encode_single_piece": lambda: core.encode_single_piece(b"helloqqqq") == po.encode_single_piece(b"helloqqqq", ranks),  # core.py:393,403
        "_encode_bytes": lambda: core._encode_bytes(b" \xec\x8b\xa4\xed") == po.encode_bytes(b" \xec\x8b\xa4\xed", pat, ranks),  # core.py:407
    }
    for line, method, n_pos, kws in sites:
        assert method in drive, f"core.py:{line}: CoreBPE.{method} has no driver here"
        assert drive[method](), f"core.py:{line}: CoreBPE.{method}"
    # the buffer the reference hands to np.frombuffer (core.py:161-162; src/py.rs:186-249): read-only, one-dimensional, 4-byte unsigned items
    mv = memoryview(core.encode_to_tiktoken_buffer(t, set()))
    assert mv.readonly and mv.ndim == 1 and mv.itemsize == 4 and mv.format in ("I", "<I", "L", "<L") and mv.contiguous
    # lone surrogates: the shim raises what PyO3's &str extraction raises, which is what core.py:77-80 catches
    with pytest.raises(UnicodeEncodeError):
        core.encode_ordinary("\ud83d")
    with pytest.raises(UnicodeEncodeError):
        core.encode("\ud83d", set())
    # the reference's batch form: functools.partial(self.encode_ordinary) mapped over ThreadPoolExecutor(num_threads) (core.py:174-176)
    docs = [TEXTS[i % len(TEXTS)] + str(i) for i in range(512)]
    with ThreadPoolExecutor(8) as pool:
        got = list(pool.map(core.encode_ordinary, docs))
    assert got == [oracle_encode(name, d) for d in docs]
    with ThreadPoolExecutor(8) as pool:
        got = list(pool.map(functools.partial(core.encode, allowed_special=allowed_all), docs))
    assert got == [oracle_encode(name, d, "all") for d in docs]
    with ThreadPoolExecutor(8) as pool:  # core.py:347-349: decode_batch = decode over a pool
        assert list(pool.map(core.decode_bytes, got)) == [d.encode() for d in docs]
