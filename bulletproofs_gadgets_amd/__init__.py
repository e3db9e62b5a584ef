"""MI355X-native Bulletproofs R1CS proving backend (host-side Python mirror).

The compute path is libbpg.so (hand-written HIP for gfx950 behind the C ABI in
include/bpg.h).  Nothing here computes on the CPU in its place.
"""
from . import _native  # noqa: F401
from ._native import BpgError, Context  # noqa: F401

__all__ = ["Context", "BpgError"]
