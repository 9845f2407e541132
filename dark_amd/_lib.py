"""ctypes binding of include/dark_amd.h (dark_amd/libdark_amd.so).  No fallback: if the library is missing the
import of anything that needs it raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("DARK_AMD_LIB") or os.path.join(HERE, "libdark_amd.so")  # DARK_AMD_LIB: another build of the same ABI (A/B runs)

DK_OK = 0
DK_E_ARG, DK_E_NOMEM, DK_E_HIP, DK_E_CAPACITY, DK_E_MODEL, DK_E_STREAM, DK_E_INTERNAL, DK_E_NODEVICE = -1, -2, -3, -4, -5, -6, -7, -8
ERROR_NAMES = {-1: "DK_E_ARG", -2: "DK_E_NOMEM", -3: "DK_E_HIP", -4: "DK_E_CAPACITY", -5: "DK_E_MODEL",
               -6: "DK_E_STREAM", -7: "DK_E_INTERNAL", -8: "DK_E_NODEVICE"}
MODEL_IDS = {"dark": 0, "exp": 1, "ybs": 2, "simple": 3, "rawdc": 4}
NUM_KERNEL_SLOTS = 32
DK_FLAG_HAS_FF, DK_FLAG_SINGLE_SYMBOL = 1, 2


class Stats(C.Structure):
    _fields_ = [("ms_h2d", C.c_double), ("ms_sa", C.c_double), ("ms_bwt", C.c_double), ("ms_dc", C.c_double),
                ("ms_d2h", C.c_double), ("ms_entropy", C.c_double), ("ms_ibwt", C.c_double), ("ms_total", C.c_double),
                ("rounds", C.c_uint32), ("sort_passes", C.c_uint32), ("sorted_elements", C.c_uint64),
                ("dc_runs", C.c_uint64), ("entropy_threads", C.c_uint32), ("entropy_l3_group", C.c_int32),
                ("kernel_launches", C.c_uint32 * NUM_KERNEL_SLOTS), ("kernel_ms", C.c_double * NUM_KERNEL_SLOTS),
                ("kernel_bytes", C.c_double * NUM_KERNEL_SLOTS), ("sa_route", C.c_uint32), ("entropy_l3_numa", C.c_int16), ("gpu_numa", C.c_int16),
                ("ws_peak_bytes", C.c_uint64), ("ws_size_bytes", C.c_uint64)]
ROUTES = {"short_prefix": 0x1, "narrow_keys": 0x2, "text_round": 0x4, "isa_windows": 0x8, "isa_marked": 0x10, "isa_buckets": 0x20,
          "general_round": 0x40, "big_groups": 0x80, "inplace_rounds": 0x100, "pair_chains": 0x200, "lfirst": 0x400,
          "lfirst_big_round": 0x800, "lfirst_deep": 0x1000, "lfirst_fallback": 0x2000, "lfirst_giant": 0x4000, "period_round": 0x8000, "packed_pairs": 0x10000}


# every symbol include/dark_amd.h declares: name -> (restype, argtypes)
_vp, _sz, _szp, _u32p, _i = C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.c_int
SIGNATURES = {
    "dk_version": (C.c_char_p, []),
    "dk_ctx_create": (_i, [_i, _sz, C.POINTER(_vp)]),
    "dk_ctx_destroy": (None, [_vp]),
    "dk_capacity": (_sz, [_vp]),
    "dk_last_consumed": (_sz, [_vp]),
    "dk_last_block_flags": (C.c_uint, [_vp]),
    "dk_last_error": (C.c_char_p, [_vp]),
    "dk_suffix_array": (_i, [_vp, _vp, _sz, _vp]),
    "dk_bwt_forward": (_i, [_vp, _vp, _sz, _vp, _u32p]),
    "dk_bwt_inverse": (_i, [_vp, _vp, _sz, C.c_uint32, _vp]),
    "dk_dc_encode": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _szp]),
    "dk_dc_decode": (_i, [_vp, _vp, _vp, _sz, _vp, _sz, _szp]),
    "dk_block_encode": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _szp]),
    "dk_block_decode": (_i, [_vp, _i, _vp, _sz, _sz, _vp]),
    "dk_raw_block_encode": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _szp, _vp, _sz, _szp]),
    "dk_raw_block_decode": (_i, [_vp, _i, _vp, _sz, _sz, _vp]),
    "dk_dev_suffix_array": (_i, [_vp, _vp, _sz, _vp]),
    "dk_dev_bwt_forward": (_i, [_vp, _vp, _sz, _vp, _u32p]),
    "dk_dev_bwt_inverse": (_i, [_vp, _vp, _sz, C.c_uint32, _vp]),
    "dk_dev_dc_encode": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _szp]),
    "dk_dev_block_encode": (_i, [_vp, _i, _vp, _sz, _vp, _sz, _szp]),
    "dk_dev_block_decode": (_i, [_vp, _i, _vp, _sz, _sz, _vp]),
    "dk_dev_batch_encode": (_i, [_vp, _i, _sz, _vp, _vp, _vp, _vp, _vp, _i]),
    "dk_batch_begin": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "dk_batch_push": (_i, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "dk_batch_finish": (_i, [_vp]),
    "dk_dev_batch_decode": (_i, [_vp, _i, _sz, _vp, _vp, _vp, _vp, _i]),
    "dk_multi_block_encode": (_i, [_vp, _i, _i, _sz, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz]),
    "dk_multi_block_decode": (_i, [_vp, _i, _i, _sz, _vp, _vp, _vp, _vp, _i, _vp, _sz]),
    "dk_model_encode": (_i, [_i, _vp, _vp, _sz, _vp, _sz, _szp]),
    "dk_model_decode": (_i, [_i, _vp, _sz, _vp, _sz, _vp]),
    "dk_bitcoder_encode": (_i, [_vp, _vp, _sz, _vp, _sz, _szp]),
    "dk_bitcoder_decode": (_i, [_vp, _sz, _vp, _sz, _vp]),
    "dk_stream_encode": (_i, [_i, _sz, _vp, _vp, _vp, _vp, _vp, _sz, C.c_uint32, _vp, _sz, _szp]),
    "dk_stream_decode": (_i, [_i, _vp, _sz, _sz, _vp, _u32p, C.POINTER(_i), _szp]),
    "dk_raw_stream_encode": (_i, [_i, _vp, _sz, C.c_uint32, _vp, _sz, _szp]),
    "dk_raw_stream_decode": (_i, [_i, _vp, _sz, _sz, _vp, _u32p, _szp]),
    "dk_set_profiling": (_i, [_vp, _i]),
    "dk_stats_reset": (_i, [_vp]),
    "dk_get_stats": (_i, [_vp, C.POINTER(Stats)]),
    "dk_kernel_name": (C.c_char_p, [_i]),
    "dk_set_entropy_threads": (_i, [_i]),
    "dk_host_l3_groups": (_i, [_i]),
    "dk_last_entropy_info": (None, [C.POINTER(_i), C.POINTER(_i)]),
    "dk_set_entropy_numa_node": (None, [_i]),
    "dk_dbg_l3_claim_order": (_i, [C.POINTER(_i), _i, _i, _i, C.POINTER(_i)]),
    "dk_dbg_stream_encode_gated": (_i, [_i, _sz, _vp, _vp, _vp, _sz, C.c_uint32, _vp, _sz, _szp, _vp, C.c_uint, _i]),
    "dk_dbg_sort_pairs": (_i, [_vp, _vp, _vp, _sz, _i, _i]),
    "dk_dbg_dev_sort_pairs": (_i, [_vp, _vp, _vp, _sz, _i, _i]),
    "dk_dbg_dev_local_sort": (_i, [_vp, _vp, _vp, _sz, _i, _i]),
}

_lib = None


def load():
    """dlopen the in-tree library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("dark_amd: %s is missing -- run `python dark_amd/build.py` (hipcc, gfx950). "
                              "There is no CPU fallback." % SO_PATH)
        # PyTorch-ROCm ships its own libamdhip64 under the same soname as /opt/rocm's.  Whichever is loaded first serves the
        # whole process; if ours came first, torch would later fail to find the GPU.  So when torch is installed, load it first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
