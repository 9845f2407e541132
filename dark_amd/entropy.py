"""Mirror of the reference's own bitwise coder (src/entropy/mod.rs, src/entropy/ari.rs)."""
import ctypes as C

import numpy as np

from . import _lib
from .context import DarkError, _ptr, as_u8


def encode_bits(bits, flat):
    lib = _lib.load()
    b = np.ascontiguousarray(bits, dtype=np.uint8)
    f = np.ascontiguousarray(flat, dtype=np.uint16)
    cap = len(b) + 16
    out = np.empty(cap, dtype=np.uint8)
    ln = C.c_size_t(0)
    rc = lib.dk_bitcoder_encode(_ptr(b), _ptr(f), len(b), _ptr(out), cap, C.byref(ln))
    if rc:
        raise DarkError(rc)
    return out[:ln.value].tobytes()


def decode_bits(stream, flat):
    lib = _lib.load()
    s = as_u8(stream)
    f = np.ascontiguousarray(flat, dtype=np.uint16)
    bits = np.empty(len(f), dtype=np.uint8)
    rc = lib.dk_bitcoder_decode(_ptr(s), len(s), _ptr(f), len(f), _ptr(bits))
    if rc:
        raise DarkError(rc)
    return bits
