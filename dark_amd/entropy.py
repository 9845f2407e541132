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


# ---- host coding threads (include/dark_amd.h "host coding threads") ----------------------------------------------------------------
def set_threads(mode):
    """thread form of the host coding pass, process-wide: 0 automatic | 1 | 2 | 4 | 5"""
    rc = _lib.load().dk_set_entropy_threads(int(mode))
    if rc:
        raise DarkError(rc, "dk_set_entropy_threads(%r)" % (mode,))


def l3_groups(min_cores):
    """last-level-cache groups in which the calling thread may use at least min_cores cores"""
    return int(_lib.load().dk_host_l3_groups(int(min_cores)))


def last_info():
    """(threads, l3_group) of the calling thread's last host coding pass; l3_group -1 = none claimed"""
    t, g = C.c_int(0), C.c_int(0)
    _lib.load().dk_last_entropy_info(C.byref(t), C.byref(g))
    return int(t.value), int(g.value)


def claim_order(group_numa, own, preferred_numa):
    """the order in which a coding pass tries the L3 groups of a machine described by the caller (group_numa[g] = memory node of group g): the
    groups on preferred_numa first, within each class the caller's own group, then by distance (dk_dbg_l3_claim_order)"""
    k = len(group_numa)
    arr = (C.c_int * k)(*[int(x) for x in group_numa])
    out = (C.c_int * k)()
    got = _lib.load().dk_dbg_l3_claim_order(arr, k, int(own), int(preferred_numa), out)
    if got < 0:
        raise DarkError(got, "dk_dbg_l3_claim_order")
    return [int(out[i]) for i in range(got)]


def plan_threads(groups4, groups2, ranks, share, groups5=0):
    """The thread form every rank of a node should use so that none of them loses the race for an L3 group: groups5 / groups4 / groups2 = the
    SMALLEST number of groups with >= 5 / >= 4 / >= 2 usable cores any rank sees, ranks = ranks on the node, share = host CPUs per rank."""
    if share >= 5 and groups5 >= ranks:
        return 5
    if share >= 4 and groups4 >= ranks:
        return 4
    if share >= 2 and groups2 >= ranks:
        return 2
    return 1
