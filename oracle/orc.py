"""TEST INFRASTRUCTURE ONLY: ctypes binding of the CPU oracle (oracle/dark_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (dark_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

MODEL_DARK, MODEL_EXP, MODEL_YBS, MODEL_SIMPLE, MODEL_RAWDC = 0, 1, 2, 3, 4
MODEL_IDS = {"dark": 0, "exp": 1, "ybs": 2, "simple": 3, "rawdc": 4}


def build(force=False):
    """Compile the oracle with gcc (plain C, no GPU)."""
    srcs = [os.path.join(_HERE, f) for f in ("dark_oracle.c", "dark_oracle.h", "sais_body.inc", "bbb_state_table.inc")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u16p, u32p, szp = C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)
        L.orc_saca_storage_words.restype = C.c_size_t
        L.orc_saca_storage_words.argtypes = [C.c_size_t]
        L.orc_sa_naive.argtypes = [u8p, C.c_size_t, u32p]
        L.orc_sa_sais.argtypes = [u8p, C.c_size_t, u32p]
        L.orc_bwt_forward.argtypes = [u8p, C.c_size_t, u32p, u8p, C.POINTER(C.c_uint32)]
        L.orc_bwt_inverse.argtypes = [u8p, C.c_size_t, C.c_uint32, u8p]
        L.orc_dc_encode.argtypes = [u8p, C.c_size_t, u32p, u32p, u32p, u8p, u8p, u32p, szp]
        L.orc_dc_decode.argtypes = [u32p, u32p, C.c_size_t, u8p, C.c_size_t, szp]
        L.orc_block_dc_encode.argtypes = [C.c_int, u8p, C.c_size_t, u8p, C.c_size_t, szp]
        L.orc_block_dc_encode_bwt.argtypes = [C.c_int, u8p, C.c_size_t, C.c_uint32, u8p, C.c_size_t, szp]
        L.orc_block_dc_decode.argtypes = [C.c_int, u8p, C.c_size_t, C.c_size_t, u8p]
        L.orc_model_encode.argtypes = [C.c_int, u32p, u8p, C.c_size_t, u8p, C.c_size_t, szp]
        L.orc_model_decode.argtypes = [C.c_int, u8p, C.c_size_t, u8p, C.c_size_t, u32p]
        L.orc_bitcoder_encode.argtypes = [u8p, u16p, C.c_size_t, u8p, C.c_size_t, szp]
        L.orc_bitcoder_decode.argtypes = [u8p, C.c_size_t, u16p, C.c_size_t, u8p]
        L.orc_raw_bbb_encode_bwt.argtypes = [u8p, C.c_size_t, C.c_uint32, u8p, C.c_size_t, szp]
        L.orc_raw_bbb_decode_bwt.argtypes = [u8p, C.c_size_t, C.c_size_t, u8p, C.POINTER(C.c_uint32)]
        L.orc_bbb_state_table.restype = C.POINTER(C.c_uint8 * 1024)
        L.orc_bbb_state_table.argtypes = []
        L.orc_last_stage_seconds.argtypes = [C.POINTER(C.c_double * 4)]
        L.orc_last_stage_seconds.restype = None
        _lib = L
    return _lib


def _u8(x):
    if isinstance(x, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(x), dtype=np.uint8).copy()
    return np.ascontiguousarray(x, dtype=np.uint8)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleError(RuntimeError):
    pass


def _ck(rc, what):
    if rc != 0:
        raise OracleError("%s failed with code %d" % (what, rc))


def sa_naive(t):
    t = _u8(t)
    sa = np.empty(len(t), dtype=np.uint32)
    _ck(lib().orc_sa_naive(_p(t), len(t), _p(sa)), "orc_sa_naive")
    return sa


def sa_sais(t):
    t = _u8(t)
    sa = np.empty(len(t), dtype=np.uint32)
    _ck(lib().orc_sa_sais(_p(t), len(t), _p(sa)), "orc_sa_sais")
    return sa


def bwt_forward(t, sa=None):
    t = _u8(t)
    if sa is None:
        sa = sa_sais(t)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    out = np.empty(len(t), dtype=np.uint8)
    origin = C.c_uint32(0)
    _ck(lib().orc_bwt_forward(_p(t), len(t), _p(sa), _p(out), C.byref(origin)), "orc_bwt_forward")
    return out, int(origin.value)


def bwt_inverse(bwt, origin):
    bwt = _u8(bwt)
    out = np.empty(len(bwt), dtype=np.uint8)
    _ck(lib().orc_bwt_inverse(_p(bwt), len(bwt), origin, _p(out)), "orc_bwt_inverse")
    return out


def dc_encode(bwt):
    """-> dict(init[256], d[m], sym[m], rank[m], limit[m], sparse[n])"""
    bwt = _u8(bwt)
    n = len(bwt)
    sparse = np.empty(n, dtype=np.uint32)
    init = np.empty(256, dtype=np.uint32)
    d = np.empty(n, dtype=np.uint32)
    sym = np.empty(n, dtype=np.uint8)
    rank = np.empty(n, dtype=np.uint8)
    limit = np.empty(n, dtype=np.uint32)
    m = C.c_size_t(0)
    _ck(lib().orc_dc_encode(_p(bwt), n, _p(sparse), _p(init), _p(d), _p(sym), _p(rank), _p(limit), C.byref(m)),
        "orc_dc_encode")
    m = m.value
    return dict(init=init, d=d[:m].copy(), sym=sym[:m].copy(), rank=rank[:m].copy(), limit=limit[:m].copy(),
                sparse=sparse)


def dc_decode(init, d, n):
    init = np.ascontiguousarray(init, dtype=np.uint32)
    d = np.ascontiguousarray(d, dtype=np.uint32)
    out = np.empty(n, dtype=np.uint8)
    used = C.c_size_t(0)
    _ck(lib().orc_dc_decode(_p(init), _p(d), len(d), _p(out), n, C.byref(used)), "orc_dc_decode")
    return out, used.value


def _model(m):
    return MODEL_IDS[m] if isinstance(m, str) else int(m)


def block_dc_encode(model, data):
    data = _u8(data)
    n = len(data)
    cap = 10 * (n + 600) if _model(model) == MODEL_RAWDC else 2 * n + 4096
    out = np.empty(cap, dtype=np.uint8)
    ln = C.c_size_t(0)
    _ck(lib().orc_block_dc_encode(_model(model), _p(data), n, _p(out), cap, C.byref(ln)), "orc_block_dc_encode")
    return out[:ln.value].tobytes()


def block_dc_encode_bwt(model, bwt, origin):
    bwt = _u8(bwt)
    n = len(bwt)
    cap = 10 * (n + 600) if _model(model) == MODEL_RAWDC else 2 * n + 4096
    out = np.empty(cap, dtype=np.uint8)
    ln = C.c_size_t(0)
    _ck(lib().orc_block_dc_encode_bwt(_model(model), _p(bwt), n, origin, _p(out), cap, C.byref(ln)),
        "orc_block_dc_encode_bwt")
    return out[:ln.value].tobytes()


def block_dc_decode(model, stream, n):
    s = _u8(stream)
    out = np.empty(n, dtype=np.uint8)
    _ck(lib().orc_block_dc_decode(_model(model), _p(s), len(s), n, _p(out)), "orc_block_dc_decode")
    return out.tobytes()


def model_encode(model, d, sym):
    d = np.ascontiguousarray(d, dtype=np.uint32)
    sym = np.ascontiguousarray(sym, dtype=np.uint8)
    cap = 16 * len(d) + 64
    out = np.empty(cap, dtype=np.uint8)
    ln = C.c_size_t(0)
    _ck(lib().orc_model_encode(_model(model), _p(d), _p(sym), len(d), _p(out), cap, C.byref(ln)), "orc_model_encode")
    return out[:ln.value].tobytes()


def model_decode(model, stream, sym):
    s = _u8(stream)
    sym = np.ascontiguousarray(sym, dtype=np.uint8)
    d = np.empty(len(sym), dtype=np.uint32)
    _ck(lib().orc_model_decode(_model(model), _p(s), len(s), _p(sym), len(sym), _p(d)), "orc_model_decode")
    return d


def bitcoder_encode(bits, flat):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    flat = np.ascontiguousarray(flat, dtype=np.uint16)
    cap = len(bits) + 16
    out = np.empty(cap, dtype=np.uint8)
    ln = C.c_size_t(0)
    _ck(lib().orc_bitcoder_encode(_p(bits), _p(flat), len(bits), _p(out), cap, C.byref(ln)), "orc_bitcoder_encode")
    return out[:ln.value].tobytes()


def bitcoder_decode(stream, flat):
    s = _u8(stream)
    flat = np.ascontiguousarray(flat, dtype=np.uint16)
    bits = np.empty(len(flat), dtype=np.uint8)
    _ck(lib().orc_bitcoder_decode(_p(s), len(s), _p(flat), len(flat), _p(bits)), "orc_bitcoder_decode")
    return bits


def raw_bbb_encode(data_or_bwt, origin=None):
    """block::raw::Encoder with the bbb model: from the text (origin None) or from a given BWT + origin"""
    x = _u8(data_or_bwt)
    if origin is None:
        x, origin = bwt_forward(x)
    n = len(x)
    out = np.empty(2 * n + 4096, dtype=np.uint8)
    ln = C.c_size_t(0)
    _ck(lib().orc_raw_bbb_encode_bwt(_p(x), n, origin, _p(out), len(out), C.byref(ln)), "orc_raw_bbb_encode_bwt")
    return out[:ln.value].tobytes()


def raw_bbb_decode(stream, n):
    s = _u8(stream)
    bwt = np.empty(n, dtype=np.uint8)
    origin = C.c_uint32(0)
    _ck(lib().orc_raw_bbb_decode_bwt(_p(s), len(s), n, _p(bwt), C.byref(origin)), "orc_raw_bbb_decode_bwt")
    return bwt_inverse(bwt, int(origin.value)).tobytes()


def bbb_state_table():
    """the 256 x 4 bit-history state table the oracle's bbb model was compiled with"""
    return np.frombuffer(lib().orc_bbb_state_table().contents, dtype=np.uint8).reshape(256, 4).copy()


def last_stage_seconds():
    arr = (C.c_double * 4)()
    lib().orc_last_stage_seconds(C.byref(arr))
    return dict(sa=arr[0], bwt=arr[1], dc=arr[2], entropy=arr[3])
