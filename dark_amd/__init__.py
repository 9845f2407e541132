"""dark_amd: MI355X-native drop-in for the block compression hot path of kvark/dark
(src/saca.rs + src/block + src/model + src/entropy) behind the C ABI of include/dark_amd.h.

The shared library dark_amd/libdark_amd.so (HIP kernels for gfx950 + host entropy stage) is required; there is no
CPU fallback.  Build it with `python dark_amd/build.py`."""
from . import _lib
from ._lib import load as load_library
from .context import Context, DarkError, multi_block_encode, multi_block_decode
from . import saca, block, model, entropy

__all__ = ["Context", "DarkError", "multi_block_encode", "multi_block_decode", "saca", "block", "model", "entropy", "load_library"]
