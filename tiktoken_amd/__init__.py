"""tiktoken_amd -- MI355X-native BPE encode path behind the tiktoken API.

Public surface mirrors the reference's `tiktoken/__init__.py`: `Encoding`, `get_encoding`,
`list_encoding_names`, `encoding_for_model`, `encoding_name_for_model`.  `CoreBPE`
(tiktoken_amd._tiktoken) is the drop-in for the reference's Rust extension class; the encode work
runs in hand-written HIP kernels (tiktoken_amd/csrc).  See DESIGN.md.
"""
__version__ = "0.1.0"

from ._tiktoken import CoreBPE  # noqa: F401
from .core import Encoding  # noqa: F401
from .model import encoding_for_model, encoding_name_for_model  # noqa: F401
from .registry import get_encoding, list_encoding_names  # noqa: F401
