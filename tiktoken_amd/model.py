"""Model name -> encoding name lookup (reference tiktoken/model.py).  Pure table; out of the hot path."""
from __future__ import annotations

from .core import Encoding
from .registry import get_encoding

# exact names win over prefixes
_O200K = "o200k_base"
_CL100K = "cl100k_base"
_P50K = "p50k_base"
_R50K = "r50k_base"

MODEL_PREFIX_TO_ENCODING: dict[str, str] = {
    "o1-": _O200K, "o3-": _O200K, "o4-mini-": _O200K,
    "gpt-5": _O200K, "gpt-4.5-": _O200K, "gpt-4.1-": _O200K, "chatgpt-4o-": _O200K, "gpt-4o-": _O200K,
    "gpt-4-": _CL100K, "gpt-3.5-turbo-": _CL100K, "gpt-35-turbo-": _CL100K,
    "gpt-oss-": "o200k_harmony",
    "ft:gpt-4o": _O200K, "ft:gpt-4": _CL100K, "ft:gpt-3.5-turbo": _CL100K,
    "ft:davinci-002": _CL100K, "ft:babbage-002": _CL100K,
}

MODEL_TO_ENCODING: dict[str, str] = {
    "o1": _O200K, "o3": _O200K, "o4-mini": _O200K,
    "gpt-5": _O200K, "gpt-4.1": _O200K, "gpt-4o": _O200K, "gpt-4": _CL100K,
    "gpt-3.5-turbo": _CL100K, "gpt-3.5": _CL100K, "gpt-35-turbo": _CL100K,
    "davinci-002": _CL100K, "babbage-002": _CL100K,
    "text-embedding-ada-002": _CL100K, "text-embedding-3-small": _CL100K, "text-embedding-3-large": _CL100K,
    "text-davinci-003": _P50K, "text-davinci-002": _P50K,
    "text-davinci-001": _R50K, "text-curie-001": _R50K, "text-babbage-001": _R50K, "text-ada-001": _R50K,
    "davinci": _R50K, "curie": _R50K, "babbage": _R50K, "ada": _R50K,
    "code-davinci-002": _P50K, "code-davinci-001": _P50K, "code-cushman-002": _P50K, "code-cushman-001": _P50K,
    "davinci-codex": _P50K, "cushman-codex": _P50K,
    "text-davinci-edit-001": "p50k_edit", "code-davinci-edit-001": "p50k_edit",
    "text-similarity-davinci-001": _R50K, "text-similarity-curie-001": _R50K,
    "text-similarity-babbage-001": _R50K, "text-similarity-ada-001": _R50K,
    "text-search-davinci-doc-001": _R50K, "text-search-curie-doc-001": _R50K,
    "text-search-babbage-doc-001": _R50K, "text-search-ada-doc-001": _R50K,
    "code-search-babbage-code-001": _R50K, "code-search-ada-code-001": _R50K,
    "gpt2": "gpt2", "gpt-2": "gpt2",
}


def encoding_name_for_model(model_name: str) -> str:
    """Name of the encoding a model uses; KeyError if the model is unknown."""
    if model_name in MODEL_TO_ENCODING:
        return MODEL_TO_ENCODING[model_name]
    for prefix, enc_name in MODEL_PREFIX_TO_ENCODING.items():
        if model_name.startswith(prefix):
            return enc_name
    raise KeyError(
        f"Could not automatically map {model_name} to a tokeniser. "
        "Please use `tiktoken.get_encoding` to explicitly get the tokeniser you expect."
    ) from None


def encoding_for_model(model_name: str) -> Encoding:
    return get_encoding(encoding_name_for_model(model_name))
