"""Encoding registry and plugin discovery (reference tiktoken/registry.py).

Plugins are modules inside the namespace package `tiktoken_ext` that define
`ENCODING_CONSTRUCTORS: dict[str, Callable[[], dict]]`; each constructor returns the keyword
arguments of `Encoding(...)`.  Identical contract to the reference, so existing plugins load.
"""
from __future__ import annotations

import functools
import importlib
import pkgutil
import threading
from typing import Any, Callable, Sequence

import tiktoken_ext

from .core import Encoding

_lock = threading.RLock()
ENCODINGS: dict[str, Encoding] = {}
ENCODING_CONSTRUCTORS: dict[str, Callable[[], dict[str, Any]]] | None = None


@functools.lru_cache
def _available_plugin_modules() -> Sequence[str]:
    # tiktoken_ext is a namespace package: iterating its __path__ finds every installed plugin
    return [name for _, name, _ in pkgutil.iter_modules(tiktoken_ext.__path__, tiktoken_ext.__name__ + ".")]


def _find_constructors() -> None:
    global ENCODING_CONSTRUCTORS
    with _lock:
        if ENCODING_CONSTRUCTORS is not None:
            return
        found: dict[str, Callable[[], dict[str, Any]]] = {}
        for mod_name in _available_plugin_modules():
            mod = importlib.import_module(mod_name)
            try:
                constructors = mod.ENCODING_CONSTRUCTORS
            except AttributeError as e:
                raise ValueError(f"tiktoken plugin {mod_name} does not define ENCODING_CONSTRUCTORS") from e
            for enc_name, constructor in constructors.items():
                if enc_name in found:
                    raise ValueError(f"Duplicate encoding name {enc_name} in tiktoken plugin {mod_name}")
                found[enc_name] = constructor
        ENCODING_CONSTRUCTORS = found  # only published when discovery succeeded, so errors re-raise next time


def get_encoding(encoding_name: str) -> Encoding:
    if not isinstance(encoding_name, str):
        raise ValueError(f"Expected a string in get_encoding, got {type(encoding_name)}")
    enc = ENCODINGS.get(encoding_name)
    if enc is not None:
        return enc
    with _lock:
        enc = ENCODINGS.get(encoding_name)
        if enc is not None:
            return enc
        if ENCODING_CONSTRUCTORS is None:
            _find_constructors()
            assert ENCODING_CONSTRUCTORS is not None
        if encoding_name not in ENCODING_CONSTRUCTORS:
            from . import __version__

            raise ValueError(
                f"Unknown encoding {encoding_name}.\n"
                f"Plugins found: {_available_plugin_modules()}\n"
                f"tiktoken version: {__version__} (are you on latest?)"
            )
        enc = Encoding(**ENCODING_CONSTRUCTORS[encoding_name]())
        ENCODINGS[encoding_name] = enc
        return enc


def list_encoding_names() -> list[str]:
    with _lock:
        if ENCODING_CONSTRUCTORS is None:
            _find_constructors()
            assert ENCODING_CONSTRUCTORS is not None
        return list(ENCODING_CONSTRUCTORS)
