"""Encodings by name, through the reference's plugin surface: any module inside the namespace package `tiktoken_ext`
that defines `ENCODING_CONSTRUCTORS: dict[str, Callable[[], dict]]` contributes encodings; each constructor returns the
keyword arguments of `Encoding(...)` (reference README "Extending tiktoken").  Existing third-party plugins load unchanged.
"""
from __future__ import annotations

import importlib
import pkgutil
import threading


class _Catalogue:
    """Lazily discovered constructors plus the encodings built from them (one instance per name)."""

    def __init__(self):
        self._guard = threading.RLock()
        self._makers = None
        self._built = {}

    def _plugin_modules(self):
        import tiktoken_ext  # a namespace package: its __path__ spans every installed plugin distribution

        return sorted(m.name for m in pkgutil.iter_modules(tiktoken_ext.__path__, "tiktoken_ext."))

    def makers(self):
        with self._guard:
            if self._makers is None:
                table = {}
                for mod_name in self._plugin_modules():
                    plugin = importlib.import_module(mod_name)
                    if not hasattr(plugin, "ENCODING_CONSTRUCTORS"):
                        raise ValueError(f"tiktoken plugin {mod_name} does not define ENCODING_CONSTRUCTORS")
                    for enc_name, maker in plugin.ENCODING_CONSTRUCTORS.items():
                        if enc_name in table:
                            raise ValueError(f"Duplicate encoding name {enc_name} in tiktoken plugin {mod_name}")
                        table[enc_name] = maker
                self._makers = table
            return self._makers

    def get(self, encoding_name: str):
        from .core import Encoding

        if not isinstance(encoding_name, str):
            raise ValueError(f"Expected a string in get_encoding, got {type(encoding_name)}")
        with self._guard:
            enc = self._built.get(encoding_name)
            if enc is None:
                makers = self.makers()
                if encoding_name not in makers:
                    raise ValueError(f"Unknown encoding {encoding_name}.\nPlugins found: {self._plugin_modules()}")
                enc = self._built[encoding_name] = Encoding(**makers[encoding_name]())
            return enc

    def registered(self, enc) -> bool:
        with self._guard:
            return self._built.get(getattr(enc, "name", None)) is enc


_CATALOGUE = _Catalogue()


def get_encoding(encoding_name: str):
    return _CATALOGUE.get(encoding_name)


def list_encoding_names() -> list[str]:
    return list(_CATALOGUE.makers())


# Model name -> encoding name (reference tiktoken/model.py:7-86; public API `encoding_for_model`).  Kept by encoding: exact names and,
# after "|", name prefixes ("gpt-4o-2024-05-13"); of several matching prefixes the longest one decides ("ft:gpt-4o" over "ft:gpt-4").
_MODELS_BY_ENCODING = {
    "o200k_base": "o1 o3 o4-mini gpt-5 gpt-4.1 gpt-4o | o1- o3- o4-mini- gpt-5 gpt-4.5- gpt-4.1- chatgpt-4o- gpt-4o- ft:gpt-4o",
    "o200k_harmony": "| gpt-oss-",
    "cl100k_base": "gpt-4 gpt-3.5-turbo gpt-3.5 gpt-35-turbo davinci-002 babbage-002 text-embedding-ada-002 text-embedding-3-small "
                   "text-embedding-3-large | gpt-4- gpt-3.5-turbo- gpt-35-turbo- ft:gpt-4 ft:gpt-3.5-turbo ft:davinci-002 ft:babbage-002",
    "p50k_base": "text-davinci-003 text-davinci-002 code-davinci-002 code-davinci-001 code-cushman-002 code-cushman-001 davinci-codex "
                 "cushman-codex |",
    "p50k_edit": "text-davinci-edit-001 code-davinci-edit-001 |",
    "r50k_base": "text-davinci-001 text-curie-001 text-babbage-001 text-ada-001 davinci curie babbage ada text-similarity-davinci-001 "
                 "text-similarity-curie-001 text-similarity-babbage-001 text-similarity-ada-001 text-search-davinci-doc-001 "
                 "text-search-curie-doc-001 text-search-babbage-doc-001 text-search-ada-doc-001 code-search-babbage-code-001 "
                 "code-search-ada-code-001 |",
    "gpt2": "gpt2 gpt-2 |",
}
_EXACT, _PREFIXES = {}, []
for _enc, _spec in _MODELS_BY_ENCODING.items():
    _names, _, _pre = _spec.partition("|")
    _EXACT.update((m, _enc) for m in _names.split())
    _PREFIXES.extend((p, _enc) for p in _pre.split())
_PREFIXES.sort(key=lambda pe: -len(pe[0]))


def encoding_name_for_model(model_name: str) -> str:
    """Name of the encoding a model uses; KeyError for a model nobody has listed."""
    hit = _EXACT.get(model_name)
    if hit is None:
        hit = next((enc for pre, enc in _PREFIXES if model_name.startswith(pre)), None)
    if hit is None:
        raise KeyError(
            f"Could not automatically map {model_name} to a tokeniser. "
            "Please use `tiktoken.get_encoding` to explicitly get the tokeniser you expect."
        ) from None
    return hit


def encoding_for_model(model_name: str):
    return get_encoding(encoding_name_for_model(model_name))
