"""Mirror of the reference's `model` module: model names of src/main.rs:73-82 and the stream-level helpers the
reference's model tests use (src/model/mod.rs:59-76)."""
import ctypes as C

import numpy as np

from . import _lib
from .context import DarkError, _ptr, as_u8, model_id

NAMES = ("dark", "exp", "ybs", "simple", "rawdc")


def encode(model, dist, sym):
    """model.reset(); model.encode(d, Context{symbol}) for every pair; finish -> bytes"""
    lib = _lib.load()
    d = np.ascontiguousarray(dist, dtype=np.uint32)
    s = np.ascontiguousarray(sym, dtype=np.uint8)
    cap = 16 * len(d) + 64
    out = np.empty(cap, dtype=np.uint8)
    ln = C.c_size_t(0)
    rc = lib.dk_model_encode(model_id(model), _ptr(d), _ptr(s), len(d), _ptr(out), cap, C.byref(ln))
    if rc:
        raise DarkError(rc)
    return out[:ln.value].tobytes()


def decode(model, stream, sym):
    lib = _lib.load()
    b = as_u8(stream)
    s = np.ascontiguousarray(sym, dtype=np.uint8)
    d = np.empty(len(s), dtype=np.uint32)
    rc = lib.dk_model_decode(model_id(model), _ptr(b), len(b), _ptr(s), len(s), _ptr(d))
    if rc:
        raise DarkError(rc)
    return d


def stream_encode(model, n, init, dist, sym, origin, rank=None, run_end=None):
    """host entropy stage alone (src/block/dc.rs:53-90)"""
    lib = _lib.load()
    init = np.ascontiguousarray(init, dtype=np.uint32)
    d = np.ascontiguousarray(dist, dtype=np.uint32)
    s = np.ascontiguousarray(sym, dtype=np.uint8)
    r = np.ascontiguousarray(rank, dtype=np.uint8) if rank is not None else None
    e = np.ascontiguousarray(run_end, dtype=np.uint32) if run_end is not None else None
    mid = model_id(model)
    cap = 10 * (len(d) + 600) if mid == 4 else 8 * len(d) + 8192
    out = np.empty(cap, dtype=np.uint8)
    ln = C.c_size_t(0)
    rc = lib.dk_stream_encode(mid, n, _ptr(init), _ptr(d), _ptr(s), _ptr(r) if r is not None else None,
                              _ptr(e) if e is not None else None, len(d), int(origin), _ptr(out), cap, C.byref(ln))
    if rc:
        raise DarkError(rc)
    return out[:ln.value].tobytes()


def stream_decode(model, stream, n, with_consumed=False):
    """host entropy stage alone (src/block/dc.rs:121-151) -> (bwt, origin, single_symbol[, bytes consumed])"""
    lib = _lib.load()
    b = as_u8(stream)
    out = np.empty(n, dtype=np.uint8)
    origin = C.c_uint32(0)
    single = C.c_int(0)
    used = C.c_size_t(0)
    rc = lib.dk_stream_decode(model_id(model), _ptr(b), len(b), n, _ptr(out), C.byref(origin), C.byref(single), C.byref(used))
    if rc:
        raise DarkError(rc)
    if with_consumed:
        return out, int(origin.value), bool(single.value), int(used.value)
    return out, int(origin.value), bool(single.value)


def raw_stream_encode(bwt, origin, raw_model=1):
    """host half of block::raw with a coding RawModel (1 = bbb): src/block/raw.rs:45-58 from (L, origin) to the coded stream"""
    lib = _lib.load()
    b = as_u8(bwt)
    out = np.empty(2 * len(b) + 4096, dtype=np.uint8)
    ln = C.c_size_t(0)
    rc = lib.dk_raw_stream_encode(int(raw_model), _ptr(b), len(b), int(origin), _ptr(out), len(out), C.byref(ln))
    if rc:
        raise DarkError(rc)
    return out[:ln.value].tobytes()


def raw_stream_decode(stream, n, raw_model=1):
    """-> (bwt, origin, bytes consumed)"""
    lib = _lib.load()
    s = as_u8(stream)
    out = np.empty(n, dtype=np.uint8)
    origin = C.c_uint32(0)
    used = C.c_size_t(0)
    rc = lib.dk_raw_stream_decode(int(raw_model), _ptr(s), len(s), n, _ptr(out), C.byref(origin), C.byref(used))
    if rc:
        raise DarkError(rc)
    return out, int(origin.value), int(used.value)
