"""Stock OpenAI encodings (same names, patterns, special-token ids and pinned vocabulary files as the
reference's tiktoken_ext/openai_public.py).  The vocabulary files are fetched -- or read from
$TIKTOKEN_CACHE_DIR -- by tiktoken_amd.vocab_io exactly as the reference does; only the three distinct
`pat_str`s below have compiled GPU scanners, which covers every stock encoding.
"""
from tiktoken_amd.vocab_io import data_gym_to_mergeable_bpe_ranks, load_tiktoken_bpe

ENDOFTEXT = "<|endoftext|>"
FIM_PREFIX = "<|fim_prefix|>"
FIM_MIDDLE = "<|fim_middle|>"
FIM_SUFFIX = "<|fim_suffix|>"
ENDOFPROMPT = "<|endofprompt|>"

_BLOB = "https://openaipublic.blob.core.windows.net"

# -- the three patterns ------------------------------------------------------------------------
# GPT-2 family (possessive rewrite of the original GPT-2 regex; same matches, faster)
r50k_pat_str = (
    r"""'(?:[sdmt]|ll|ve|re)| ?\p{L}++| ?\p{N}++| ?[^\s\p{L}\p{N}]++|\s++$|\s+(?!\S)|\s"""
)
cl100k_pat_str = r"""'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}++|\p{N}{1,3}+| ?[^\s\p{L}\p{N}]++[\r\n]*+|\s++$|\s*[\r\n]|\s+(?!\S)|\s"""
_CONTRACTION = r"""(?i:'s|'t|'re|'ve|'m|'ll|'d)?"""
_UPPERISH, _LOWERISH = r"""[\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}]""", r"""[\p{Ll}\p{Lm}\p{Lo}\p{M}]"""
o200k_pat_str = "|".join([
    r"""[^\r\n\p{L}\p{N}]?""" + _UPPERISH + "*" + _LOWERISH + "+" + _CONTRACTION,
    r"""[^\r\n\p{L}\p{N}]?""" + _UPPERISH + "+" + _LOWERISH + "*" + _CONTRACTION,
    r"""\p{N}{1,3}""",
    r""" ?[^\s\p{L}\p{N}]+[\r\n/]*""",
    r"""\s*[\r\n]+""",
    r"""\s+(?!\S)""",
    r"""\s+""",
])

# -- pinned vocabulary files ---------------------------------------------------------------------
_TIKTOKEN_FILES = {
    "r50k_base": (f"{_BLOB}/encodings/r50k_base.tiktoken", "306cd27f03c1a714eca7108e03d66b7dc042abe8c258b44c199a7ed9838dd930"),
    "p50k_base": (f"{_BLOB}/encodings/p50k_base.tiktoken", "94b5ca7dff4d00767bc256fdd1b27e5b17361d7b8a5f968547f9f23eb70d2069"),
    "cl100k_base": (f"{_BLOB}/encodings/cl100k_base.tiktoken", "223921b76ee99bde995b7ff738513eef100fb51d18c93597a113bcffe865b2a7"),
    "o200k_base": (f"{_BLOB}/encodings/o200k_base.tiktoken", "446a9538cb6c348e3516120d7c08b09f57c36495e2acfffe59a5bf8b0cfb1a2d"),
}


def _ranks(key):
    url, sha256 = _TIKTOKEN_FILES[key]
    return load_tiktoken_bpe(url, expected_hash=sha256, lazy=True)  # (goes straight into an Encoding)


def gpt2():
    ranks = data_gym_to_mergeable_bpe_ranks(
        vocab_bpe_file=f"{_BLOB}/gpt-2/encodings/main/vocab.bpe",
        encoder_json_file=f"{_BLOB}/gpt-2/encodings/main/encoder.json",
        vocab_bpe_hash="1ce1664773c50f3e0cc8842619a93edc4624525b728b188a9e0be33b7726adc5",
        encoder_json_hash="196139668be63f3b5d6574427317ae82f612a97c5d1cdaf36ed2256dbf636783",
    )
    return {"name": "gpt2", "explicit_n_vocab": 50257, "pat_str": r50k_pat_str, "mergeable_ranks": ranks,
            "special_tokens": {ENDOFTEXT: 50256}}


def r50k_base():
    return {"name": "r50k_base", "explicit_n_vocab": 50257, "pat_str": r50k_pat_str, "mergeable_ranks": _ranks("r50k_base"),
            "special_tokens": {ENDOFTEXT: 50256}}


def p50k_base():
    return {"name": "p50k_base", "explicit_n_vocab": 50281, "pat_str": r50k_pat_str, "mergeable_ranks": _ranks("p50k_base"),
            "special_tokens": {ENDOFTEXT: 50256}}


def p50k_edit():
    return {"name": "p50k_edit", "pat_str": r50k_pat_str, "mergeable_ranks": _ranks("p50k_base"),
            "special_tokens": {ENDOFTEXT: 50256, FIM_PREFIX: 50281, FIM_MIDDLE: 50282, FIM_SUFFIX: 50283}}


def cl100k_base():
    return {"name": "cl100k_base", "pat_str": cl100k_pat_str, "mergeable_ranks": _ranks("cl100k_base"),
            "special_tokens": {ENDOFTEXT: 100257, FIM_PREFIX: 100258, FIM_MIDDLE: 100259, FIM_SUFFIX: 100260,
                               ENDOFPROMPT: 100276}}


def o200k_base():
    return {"name": "o200k_base", "pat_str": o200k_pat_str, "mergeable_ranks": _ranks("o200k_base"),
            "special_tokens": {ENDOFTEXT: 199999, ENDOFPROMPT: 200018}}


def o200k_harmony():
    base = o200k_base()
    named = {
        "<|startoftext|>": 199998, "<|endoftext|>": 199999, "<|return|>": 200002, "<|constrain|>": 200003,
        "<|channel|>": 200005, "<|start|>": 200006, "<|end|>": 200007, "<|message|>": 200008, "<|call|>": 200012,
    }
    reserved = [200000, 200001, 200004, 200009, 200010, 200011] + list(range(200013, 201088))
    specials = {**base["special_tokens"], **named, **{f"<|reserved_{i}|>": i for i in reserved}}
    return {"name": "o200k_harmony", "pat_str": base["pat_str"], "mergeable_ranks": base["mergeable_ranks"],
            "special_tokens": specials}


ENCODING_CONSTRUCTORS = {
    "gpt2": gpt2,
    "r50k_base": r50k_base,
    "p50k_base": p50k_base,
    "p50k_edit": p50k_edit,
    "cl100k_base": cl100k_base,
    "o200k_base": o200k_base,
    "o200k_harmony": o200k_harmony,
}
