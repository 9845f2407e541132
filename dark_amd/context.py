"""dk_ctx wrapper: one (host thread, GPU) pair owning the device workspace -- the role saca::Constructor's storage
(src/saca.rs:344-384) and block::dc::{Encoder,Decoder}'s buffers (src/block/dc.rs:21-26,96-102) play in the reference."""
import ctypes as C

import numpy as np

from . import _lib


class DarkError(RuntimeError):
    def __init__(self, code, text=""):
        self.code = code
        super().__init__("%s (%d)%s" % (_lib.ERROR_NAMES.get(code, "error"), code, (": " + text) if text else ""))


def _ptr(a):
    """host numpy array or device tensor / raw int -> void*"""
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


def _inputs_ready(*tensors):
    """Stream contract of the dk_dev_ entry points (include/dark_amd.h): device inputs must be complete before the call -- the library
    runs on its own non-blocking stream and does not wait for the caller's.  For torch tensors: drain torch's current stream."""
    for t in tensors:
        if hasattr(t, "is_cuda") and t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()
            return


def as_u8(x):
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x, dtype=np.uint8)
    return np.frombuffer(bytes(x), dtype=np.uint8)


def model_id(m):
    if isinstance(m, str):
        if m not in _lib.MODEL_IDS:
            raise DarkError(_lib.DK_E_MODEL, "unknown model %r" % m)
        return _lib.MODEL_IDS[m]
    return int(getattr(m, "MODEL_ID", m))


class Context:
    def __init__(self, max_n, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.dk_ctx_create(int(device), int(max_n), C.byref(h))
        if rc != 0:
            raise DarkError(rc, "dk_ctx_create(device=%d, max_n=%d)" % (device, max_n))
        self._h = h
        self._batch = None  # the streaming Batch open on this context, if any
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            b = getattr(self, "_batch", None)
            if b is not None:
                b.close()  # joins its coding threads before the staging memory goes (dk_ctx_destroy would do the same)
            self._lib.dk_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            raise DarkError(rc, (self._lib.dk_last_error(self._h) or b"").decode())

    def capacity(self):
        return int(self._lib.dk_capacity(self._h))

    def last_consumed(self):
        return int(self._lib.dk_last_consumed(self._h))

    def last_block_flags(self):
        """DK_FLAG_HAS_FF (1): the last block encode's input held byte 0xFF -> its stream cannot be decoded (reference format)"""
        return int(self._lib.dk_last_block_flags(self._h))

    # ---- host-pointer stage calls ----
    def suffix_array(self, data):
        t = as_u8(data)
        sa = np.empty(len(t), dtype=np.uint32)
        self._ck(self._lib.dk_suffix_array(self._h, _ptr(t), len(t), _ptr(sa)))
        return sa

    def bwt_forward(self, data):
        t = as_u8(data)
        out = np.empty(len(t), dtype=np.uint8)
        origin = C.c_uint32(0)
        self._ck(self._lib.dk_bwt_forward(self._h, _ptr(t), len(t), _ptr(out), C.byref(origin)))
        return out, int(origin.value)

    def bwt_inverse(self, bwt, origin):
        b = as_u8(bwt)
        out = np.empty(len(b), dtype=np.uint8)
        self._ck(self._lib.dk_bwt_inverse(self._h, _ptr(b), len(b), int(origin), _ptr(out)))
        return out

    def dc_encode(self, bwt):
        b = as_u8(bwt)
        n = len(b)
        init = np.empty(256, dtype=np.uint32)
        d = np.empty(n, dtype=np.uint32)
        sym = np.empty(n, dtype=np.uint8)
        rank = np.empty(n, dtype=np.uint8)
        m = C.c_size_t(0)
        self._ck(self._lib.dk_dc_encode(self._h, _ptr(b), n, _ptr(init), _ptr(d), _ptr(sym), _ptr(rank), C.byref(m)))
        m = m.value
        return dict(init=init, d=d[:m].copy(), sym=sym[:m].copy(), rank=rank[:m].copy())

    def dc_decode(self, init, d, n):
        init = np.ascontiguousarray(init, dtype=np.uint32)
        d = np.ascontiguousarray(d, dtype=np.uint32)
        out = np.empty(n, dtype=np.uint8)
        used = C.c_size_t(0)
        self._ck(self._lib.dk_dc_decode(self._h, _ptr(init), _ptr(d), len(d), _ptr(out), n, C.byref(used)))
        return out, used.value

    def block_encode(self, model, data):
        t = as_u8(data)
        n = len(t)
        mid = model_id(model)
        cap = 10 * (n + 600) if mid == _lib.MODEL_IDS["rawdc"] else 2 * n + 4096
        out = np.empty(cap, dtype=np.uint8)
        ln = C.c_size_t(0)
        self._ck(self._lib.dk_block_encode(self._h, mid, _ptr(t), n, _ptr(out), cap, C.byref(ln)))
        return out[:ln.value].tobytes()

    def block_encode_into(self, model, data, out):
        """dk_block_encode with a caller-owned output array (no copy of the stream): `data` is host memory -- pageable or pinned --
        and the H2D copy is part of the call (stats()["ms_h2d"]); returns a view of `out`"""
        t = as_u8(data)
        ln = C.c_size_t(0)
        self._ck(self._lib.dk_block_encode(self._h, model_id(model), _ptr(t), len(t), _ptr(out), len(out), C.byref(ln)))
        return out[:ln.value]

    def block_decode(self, model, stream, n):
        s = as_u8(stream)
        out = np.empty(n, dtype=np.uint8)
        self._ck(self._lib.dk_block_decode(self._h, model_id(model), _ptr(s), len(s), n, _ptr(out)))
        return out.tobytes()

    def raw_block_encode_dump(self, data, raw_model=0):
        """block::raw::Encoder with the dump model Out (src/block/raw.rs:35-59, src/model/raw.rs:46-76) -> what lands in ./out.raw"""
        t = as_u8(data)
        n = len(t)
        out = np.empty(8, dtype=np.uint8)
        dump = np.empty(n + 4, dtype=np.uint8)
        ln, dl = C.c_size_t(0), C.c_size_t(0)
        self._ck(self._lib.dk_raw_block_encode(self._h, int(raw_model), _ptr(t), n, _ptr(out), len(out), C.byref(ln), _ptr(dump), len(dump), C.byref(dl)))
        assert ln.value == 4 and not out[:4].any()
        return dump[:dl.value].tobytes()

    def raw_block_encode(self, data, raw_model=1):
        """block::raw::Encoder with a coding RawModel (1 = bbb) -> coded stream"""
        t = as_u8(data)
        n = len(t)
        out = np.empty(2 * n + 4096, dtype=np.uint8)
        ln, dl = C.c_size_t(0), C.c_size_t(0)
        self._ck(self._lib.dk_raw_block_encode(self._h, int(raw_model), _ptr(t), n, _ptr(out), len(out), C.byref(ln), None, 0, C.byref(dl)))
        return out[:ln.value].tobytes()

    def raw_block_decode(self, stream, n, raw_model=0):
        s = as_u8(stream)
        out = np.empty(n, dtype=np.uint8)
        self._ck(self._lib.dk_raw_block_decode(self._h, int(raw_model), _ptr(s), len(s), n, _ptr(out)))
        return out.tobytes()

    # ---- device-resident calls (torch tensors or raw device addresses) ----
    def dev_suffix_array(self, d_in, n, d_sa_out):
        _inputs_ready(d_in)
        self._ck(self._lib.dk_dev_suffix_array(self._h, _ptr(d_in), n, _ptr(d_sa_out)))

    def dev_bwt_forward(self, d_in, n, d_bwt_out):
        _inputs_ready(d_in)
        origin = C.c_uint32(0)
        self._ck(self._lib.dk_dev_bwt_forward(self._h, _ptr(d_in), n, _ptr(d_bwt_out), C.byref(origin)))
        return int(origin.value)

    def dev_bwt_inverse(self, d_bwt, n, origin, d_out):
        _inputs_ready(d_bwt)
        self._ck(self._lib.dk_dev_bwt_inverse(self._h, _ptr(d_bwt), n, int(origin), _ptr(d_out)))

    def dev_dc_encode(self, d_bwt, n, d_dist, d_sym, d_rank=None):
        _inputs_ready(d_bwt)
        init = np.empty(256, dtype=np.uint32)
        m = C.c_size_t(0)
        self._ck(self._lib.dk_dev_dc_encode(self._h, _ptr(d_bwt), n, _ptr(init), _ptr(d_dist), _ptr(d_sym),
                                            _ptr(d_rank) if d_rank is not None else None, C.byref(m)))
        return init, m.value

    def dev_block_encode(self, model, d_in, n, out=None):
        """out: optional preallocated host uint8 array; returns a view of the coded stream"""
        _inputs_ready(d_in)
        mid = model_id(model)
        if out is None:
            out = np.empty(2 * n + 4096, dtype=np.uint8)
        ln = C.c_size_t(0)
        self._ck(self._lib.dk_dev_block_encode(self._h, mid, _ptr(d_in), n, _ptr(out), len(out), C.byref(ln)))
        return out[:ln.value]

    def dev_batch_encode(self, model, d_blocks, sizes, host_threads=8, outs=None):
        """d_blocks: device tensors / addresses; returns a list of coded streams (views of `outs` when given)"""
        _inputs_ready(*d_blocks[:1])
        count = len(d_blocks)
        if outs is None:
            outs = [np.empty(2 * int(n) + 4096, dtype=np.uint8) for n in sizes]
        ptrs = (C.c_void_p * count)(*[_ptr(b) for b in d_blocks])
        ns = (C.c_size_t * count)(*[int(n) for n in sizes])
        optrs = (C.c_void_p * count)(*[_ptr(o) for o in outs])
        caps = (C.c_size_t * count)(*[len(o) for o in outs])
        lens = (C.c_size_t * count)()
        self._ck(self._lib.dk_dev_batch_encode(self._h, model_id(model), count, ptrs, ns, optrs, caps, lens, int(host_threads)))
        return [o[:lens[i]] for i, o in enumerate(outs)]

    def batch_begin(self, model, host_threads=8):
        """streaming form of dev_batch_encode: returns a Batch; push(d_in, n) per block as it arrives, then finish() -> list of streams"""
        return Batch(self, model, host_threads)

    def dev_batch_decode(self, model, streams, sizes, d_outs, host_threads=8):
        count = len(streams)
        keep = [as_u8(x) for x in streams]
        ins = (C.c_void_p * count)(*[_ptr(x) for x in keep])
        lens = (C.c_size_t * count)(*[len(x) for x in keep])
        ns = (C.c_size_t * count)(*[int(n) for n in sizes])
        outs = (C.c_void_p * count)(*[_ptr(o) for o in d_outs])
        self._ck(self._lib.dk_dev_batch_decode(self._h, model_id(model), count, ins, lens, ns, outs, int(host_threads)))

    def dev_block_decode(self, model, stream, n, d_out):
        s = as_u8(stream)
        self._ck(self._lib.dk_dev_block_decode(self._h, model_id(model), _ptr(s), len(s), n, _ptr(d_out)))

    # ---- measurement ----
    def set_profiling(self, enabled):
        self._ck(self._lib.dk_set_profiling(self._h, 1 if enabled else 0))

    def stats_reset(self):
        self._ck(self._lib.dk_stats_reset(self._h))

    def stats(self):
        st = _lib.Stats()
        self._ck(self._lib.dk_get_stats(self._h, C.byref(st)))
        out = {k: getattr(st, k) for k in ("ms_h2d", "ms_sa", "ms_bwt", "ms_dc", "ms_d2h", "ms_entropy", "ms_ibwt",
                                           "ms_total", "rounds", "sort_passes", "sorted_elements", "dc_runs",
                                           "entropy_threads", "entropy_l3_group", "entropy_l3_numa", "gpu_numa", "ws_peak_bytes", "ws_size_bytes")}
        kernels = {}
        for i in range(_lib.NUM_KERNEL_SLOTS):
            name = self._lib.dk_kernel_name(i)
            if name and st.kernel_launches[i]:
                kernels[name.decode()] = dict(launches=int(st.kernel_launches[i]), ms=float(st.kernel_ms[i]),
                                              bytes=float(st.kernel_bytes[i]))
        out["kernels"] = kernels
        out["routes"] = {name for name, bit in _lib.ROUTES.items() if st.sa_route & bit}  # which ways the last suffix sort took
        return out

    def dbg_sort_pairs(self, keys, vals, begin_bit=0, end_bit=64):
        keys = np.ascontiguousarray(keys, dtype=np.uint64).copy()
        vals = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        self._ck(self._lib.dk_dbg_sort_pairs(self._h, _ptr(keys), _ptr(vals), len(keys), begin_bit, end_bit))
        return keys, vals


class Batch:
    """dk_batch_begin / _push / _finish: the pipelined encoder fed one block at a time.

    The C coding threads write into this object's `out` buffers and length words until dk_batch_finish has joined them, so finish
    ALWAYS runs before those buffers or the context can go away: on a failed push (the error is raised after the batch is closed), at
    the end of a `with` block, from Context.close(), and as a last resort from __del__."""

    def __init__(self, ctx, model, host_threads):
        self._ctx = ctx
        self._h = C.c_void_p()
        self._outs, self._lens = [], []
        ctx._ck(ctx._lib.dk_batch_begin(ctx._h, model_id(model), int(host_threads), C.byref(self._h)))
        ctx._batch = self

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, d_in, n):
        """device stages of one block now (d_in may be reused when this returns), coding in the background"""
        if not self._h:
            raise DarkError(_lib.DK_E_ARG, "the batch is closed")
        _inputs_ready(d_in)
        out = np.empty(2 * int(n) + 4096, dtype=np.uint8)  # virtual until written: only the coded bytes become resident
        ln = C.c_size_t(0)
        self._outs.append(out)
        self._lens.append(ln)
        rc = self._ctx._lib.dk_batch_push(self._h, _ptr(d_in), int(n), _ptr(out), len(out), C.byref(ln))
        if rc != 0:
            text = (self._ctx._lib.dk_last_error(self._ctx._h) or b"").decode()
            self.close()  # joins the coders of the blocks already queued; only then may the buffers die with this object
            raise DarkError(rc, text)

    def close(self):
        """dk_batch_finish without collecting anything (idempotent); returns its code"""
        h, self._h = self._h, None
        if not h:
            return 0
        if getattr(self._ctx, "_batch", None) is self:
            self._ctx._batch = None
        return self._ctx._lib.dk_batch_finish(h)

    def finish(self):
        if not self._h:
            raise DarkError(_lib.DK_E_ARG, "the batch is closed")
        self._ctx._ck(self.close())
        return [o[:ln.value] for o, ln in zip(self._outs, self._lens)]


def multi_block_encode(model, blocks, devices, host_threads_per_gpu=4):
    """dk_multi_block_encode: block i -> GPU devices[i mod len(devices)], one host thread + one context per listed device inside the call"""
    lib = _lib.load()
    keep = [as_u8(b) for b in blocks]
    count = len(keep)
    outs = [np.empty(2 * len(b) + 4096, dtype=np.uint8) for b in keep]
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    ins = (C.c_void_p * count)(*[_ptr(b) for b in keep])
    ns = (C.c_size_t * count)(*[len(b) for b in keep])
    optrs = (C.c_void_p * count)(*[_ptr(o) for o in outs])
    caps = (C.c_size_t * count)(*[len(o) for o in outs])
    lens = (C.c_size_t * count)()
    err = C.create_string_buffer(512)
    rc = lib.dk_multi_block_encode(devs, len(devices), model_id(model), count, ins, ns, optrs, caps, lens, int(host_threads_per_gpu), err, 512)
    if rc:
        raise DarkError(rc, err.value.decode())
    return [o[:lens[i]].tobytes() for i, o in enumerate(outs)]


def multi_block_decode(model, streams, sizes, devices, host_threads_per_gpu=4):
    lib = _lib.load()
    keep = [as_u8(s) for s in streams]
    count = len(keep)
    outs = [np.empty(int(n), dtype=np.uint8) for n in sizes]
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    ins = (C.c_void_p * count)(*[_ptr(s) for s in keep])
    lens = (C.c_size_t * count)(*[len(s) for s in keep])
    ns = (C.c_size_t * count)(*[int(n) for n in sizes])
    optrs = (C.c_void_p * count)(*[_ptr(o) for o in outs])
    err = C.create_string_buffer(512)
    rc = lib.dk_multi_block_decode(devs, len(devices), model_id(model), count, ins, lens, ns, optrs, int(host_threads_per_gpu), err, 512)
    if rc:
        raise DarkError(rc, err.value.decode())
    return [o.tobytes() for o in outs]
