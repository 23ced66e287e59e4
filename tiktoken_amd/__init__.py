"""tiktoken_amd -- MI355X-native BPE encode path behind the tiktoken API.

`CoreBPE` (tiktoken_amd._tiktoken) is the drop-in for the reference's Rust extension class `tiktoken._tiktoken.CoreBPE`; the
encode work runs in hand-written HIP kernels (tiktoken_amd/csrc).  `Encoding` mirrors `tiktoken.Encoding` with the batch
methods handing whole batches to the GPU; `get_encoding` / `list_encoding_names` resolve names through the reference's
`tiktoken_ext` plugin surface.  (The unmodified reference Python package also runs over `CoreBPE`: INTEGRATION.md.)  See DESIGN.md.
"""
__version__ = "0.1.0"

from ._tiktoken import CoreBPE  # noqa: F401
from .core import Encoding  # noqa: F401
from .plugins import encoding_for_model, encoding_name_for_model, get_encoding, list_encoding_names  # noqa: F401
