"""tiktoken_amd -- MI355X-native BPE encode path behind the tiktoken API.

`CoreBPE` (tiktoken_amd._tiktoken) is the drop-in for the reference's Rust extension class; the
encode work runs in hand-written HIP kernels (tiktoken_amd/csrc).  See DESIGN.md.
"""
__version__ = "0.1.0"

from ._tiktoken import CoreBPE  # noqa: F401
