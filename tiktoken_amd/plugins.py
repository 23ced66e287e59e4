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
