"""Offline "-shaped" encodings: the stock pat_str and special-token ids of gpt2 / cl100k_base /
o200k_base paired with ranks trained offline (tools/train_vocab.py), because no stock vocabulary
file is reachable without a network.  Used by the tests and bench.py; registered
through the same plugin mechanism as any third-party encoding (reference README "Extending tiktoken").
"""
import gzip
import os

import tiktoken_amd
from tiktoken_amd.vocab_io import parse_tiktoken_bpe

from . import openai_public as _pub

_VOCAB_DIR = os.path.join(os.path.dirname(os.path.abspath(tiktoken_amd.__file__)), "vocab")


def _ranks(name):
    path = os.path.join(_VOCAB_DIR, name + ".tiktoken.gz")
    with open(path, "rb") as f:
        return parse_tiktoken_bpe(gzip.decompress(f.read()), path, lazy=True)  # (goes straight into an Encoding)


def gpt2_shaped():
    return {"name": "gpt2_shaped", "explicit_n_vocab": 50257, "pat_str": _pub.r50k_pat_str,
            "mergeable_ranks": _ranks("gpt2_shaped"), "special_tokens": {_pub.ENDOFTEXT: 50256}}


def cl100k_shaped():
    return {"name": "cl100k_shaped", "pat_str": _pub.cl100k_pat_str, "mergeable_ranks": _ranks("cl100k_shaped"),
            "special_tokens": {_pub.ENDOFTEXT: 100257, _pub.FIM_PREFIX: 100258, _pub.FIM_MIDDLE: 100259,
                               _pub.FIM_SUFFIX: 100260, _pub.ENDOFPROMPT: 100276}}


def o200k_shaped():
    return {"name": "o200k_shaped", "pat_str": _pub.o200k_pat_str, "mergeable_ranks": _ranks("o200k_shaped"),
            "special_tokens": {_pub.ENDOFTEXT: 199999, _pub.ENDOFPROMPT: 200018}}


def o200k_custom8():
    """BASELINE.json config 5: o200k + 8 custom special tokens (README "Extending tiktoken" pattern)."""
    base = o200k_shaped()
    specials = {**base["special_tokens"], **{f"<|custom_{i}|>": 200019 + i for i in range(8)}}
    return {"name": "o200k_custom8", "pat_str": base["pat_str"], "mergeable_ranks": base["mergeable_ranks"],
            "special_tokens": specials}


ENCODING_CONSTRUCTORS = {
    "gpt2_shaped": gpt2_shaped,
    "cl100k_shaped": cl100k_shaped,
    "o200k_shaped": o200k_shaped,
    "o200k_custom8": o200k_custom8,
}
