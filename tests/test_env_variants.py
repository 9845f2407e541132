"""Alternative code paths selected by environment variables (read once per process, hence subprocesses).  Every variant must give
the same bytes.
  DK_ENTROPY_THREADS=1|2|4  host coder: one thread, models | coder on two, the four-stage pipeline of the dark model (product library)
The device-side switches exist in the TUNING build only (dark_amd/libdark_amd_tuning.so, -DDK_TUNING: csrc/context.hpp DK_KNOB); the
product library has them compiled in as constants:
  DK_XCD=0 plain tile order | DK_DIGIT_PLANE=0 / 2 histograms from the keys always / from the digit plane from 2^26 pairs only | DK_PLATEAU=0 general doubling rounds only | DK_PAIR_CHAINS=0 no pair chains in front of the in-place rounds
  DK_BWT_CARRY=0 L gathered from the suffix array instead of riding with the suffixes | DK_PREFIX=0|1|2|3 prefix length of the initial sort
  DK_LF_MEDIUM=0 the L-first path's big list as one region through the global sort (default: groups of up to 8192 members sorted inside LDS)
  DK_CHAIN_GROUP=2|3 pair chains for groups of at most two / three members | DK_LF_SHORT_SLOTS / _STEPS, DK_LF_STEPS / _STUCK the step limits of k_lf_finish
  DK_POISON=<byte> every workspace allocation is filled with that byte first (a read of memory nobody wrote fails every time, not once in a while)
  DK_PACKED_SORT=0 the initial sort of at most 32 key bits moves (key, position) pairs, not packed words
  DK_LF_FORK / _REFORK when the deep groups are ordered beside the rounds (0: only at the end) | DK_LF_WAVE=0 no wave kernel | DK_LF_TAIL groups of a short big list end inside LDS
  DK_PERIOD=0 never a period round | 2 a period round wherever the probe finds one periodic 64-byte window (the product asks for an eighth of the block)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CPU_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from dark_amd import model, datagen
from oracle import orc
t = datagen.wiki_like(400000, 5)
bwt, origin = orc.bwt_forward(t)
dc = orc.dc_encode(bwt)
for m in ("dark", "exp", "ybs", "simple"):
    s = model.stream_encode(m, len(t), dc["init"], dc["d"], dc["sym"], origin)
    assert s == orc.block_dc_encode_bwt(m, bwt, origin), m
print("ok")
"""

# error paths of the two-thread coder: the coding thread runs out of output space / the model thread refuses a distance; both must
# come back as return codes (no hang: the side that fails keeps the ring moving until the other one has ended)
CPU_ERROR_SNIPPET = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r)
from dark_amd import _lib, datagen
from oracle import orc
lib = _lib.load()
t = datagen.wiki_like(300000, 6)
bwt, origin = orc.bwt_forward(t)
dc = orc.dc_encode(bwt)
init = np.ascontiguousarray(dc["init"], np.uint32); d = np.ascontiguousarray(dc["d"], np.uint32); s = np.ascontiguousarray(dc["sym"], np.uint8)
def call(dist, cap):
    out = np.empty(max(cap, 1), np.uint8); ln = C.c_size_t(0)
    return lib.dk_stream_encode(0, len(t), init.ctypes.data, dist.ctypes.data, s.ctypes.data, None, None, len(dist), int(origin),
                                out.ctypes.data, cap, C.byref(ln)), ln.value
rc, ln = call(d, 8 * len(d) + 8192)
assert rc == 0 and ln > 1000
for cap in (0, 3, 100, ln // 2, ln - 1):
    rc2, _ = call(d, cap)
    assert rc2 == _lib.DK_E_CAPACITY, (cap, rc2)
bad = d.copy(); bad[len(bad) // 2] = 0x7FFFFFFF
rc3, _ = call(bad, 8 * len(d) + 8192)
assert rc3 == _lib.DK_E_MODEL, rc3
rc4, ln4 = call(d, ln)   # exactly enough
assert rc4 == 0 and ln4 == ln
print("ok")
"""

GPU_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import dark_amd
from dark_amd import datagen
from oracle import orc
rng = np.random.default_rng(3)
with dark_amd.Context(6 << 20) as ctx:
    keys = rng.integers(0, 1 << 62, size=300001, dtype=np.uint64)
    vals = np.arange(len(keys), dtype=np.uint32)
    k2, v2 = ctx.dbg_sort_pairs(keys, vals, 0, 64)
    assert (v2 == vals[np.argsort(keys, kind="stable")]).all()
    def copies(odd):  # 5000 copies of a passage behind one byte value, one copy behind another: groups across many tiles, one of them live
        seg = rng.integers(97, 123, size=40, dtype=np.uint8)
        parts = []
        for k in range(5000):
            parts += [rng.integers(0, 255, size=int(rng.integers(20, 60)), dtype=np.uint8), np.array([66 if k == odd else 65], np.uint8), seg]
        return np.concatenate(parts + [rng.integers(0, 255, size=50000, dtype=np.uint8)])
    for t in (datagen.wiki_like(4500000, 7), np.frombuffer(b"ab" * 50000, np.uint8), rng.integers(0, 4, size=300000, dtype=np.uint8), copies(0), copies(2500),
              copies(4999)):
        t = np.ascontiguousarray(t)
        want = orc.sa_sais(t)
        assert (ctx.suffix_array(t) == want).all()
        bwt, origin = ctx.bwt_forward(t)      # BwtCarry (the symbol in front rides with the suffix) or the gather, by variant
        wb, wo = orc.bwt_forward(t, want)
        assert origin == wo and (bwt == wb).all()
print("ok")
"""


# the short-prefix path of the suffix sort (DK_PREFIX: 0 = never, 1 = ask the sample, 2 / 3 = always, four / five passes): initial sort on a few leading
# symbols, survivors finished from the text, rank array built late only when long repeats remain
PREFIX_SNIPPET = r"""
import os, sys, numpy as np
sys.path.insert(0, %r)
import dark_amd
from dark_amd import datagen
from oracle import orc
rng = np.random.default_rng(11)
def planted(base, seg_len, copies):
    t = base.copy()
    seg = t[1000:1000 + seg_len].copy()
    for c in range(copies):
        o = int(rng.integers(0, len(t) - seg_len))
        t[o:o + seg_len] = seg
    return t
cases = [
    ("random bytes 5M", rng.integers(0, 256, size=5_000_000, dtype=np.uint8)),
    ("random bytes + 3 copies of 50 KB", planted(rng.integers(0, 256, size=5_000_000, dtype=np.uint8), 50_000, 3)),
    ("acgt 5M", datagen.acgt(5_000_000, 4)),
    ("acgt + repeats", planted(datagen.acgt(4_500_000, 5), 20_000, 4)),
    ("zero-heavy", np.where(rng.random(4_300_000) < 0.9, 0, rng.integers(0, 7, size=4_300_000)).astype(np.uint8)),
    ("ends in zeros", np.concatenate([rng.integers(0, 256, size=4_200_000, dtype=np.uint8), np.zeros(3000, np.uint8)])),
]
if os.environ.get("DK_PREFIX") in ("2", "3"):
    cases += [("text", datagen.wiki_like(3_000_000, 9)), ("ab", np.frombuffer(b"ab" * 40000, np.uint8)),
              ("one", np.array([7], np.uint8)), ("two", np.array([1, 1], np.uint8)), ("tiny", rng.integers(0, 3, size=70, dtype=np.uint8)),
              ("run then one", np.concatenate([np.zeros(100000, np.uint8), np.ones(1, np.uint8)])),
              ("binary", rng.integers(0, 2, size=1_000_000, dtype=np.uint8))]
with dark_amd.Context(6 << 20) as ctx:
    for name, t in cases:
        t = np.ascontiguousarray(t)
        want = orc.sa_sais(t) if len(t) > 1 else np.zeros(1, np.uint32)
        assert (ctx.suffix_array(t) == want).all(), name
        if len(t) > 1:  # the BWT rides along with the short-prefix sort (previous symbol in the keys' low byte)
            bwt, origin = ctx.bwt_forward(t)
            wb, wo = orc.bwt_forward(t)
            assert origin == wo and (np.frombuffer(bwt, np.uint8) == np.frombuffer(wb, np.uint8)).all(), name
print("ok")
"""


TUNING_LIB = os.path.join(ROOT, "dark_amd", "libdark_amd_tuning.so")


def _run(snippet, env, tuning=False):
    e = dict(os.environ)
    e.update(env)
    if tuning:
        assert os.path.exists(TUNING_LIB), "build the tuning library: python dark_amd/build.py --tuning (__graft_entry__.build() does)"
        e["DARK_AMD_LIB"] = TUNING_LIB
    out = subprocess.run([sys.executable, "-c", snippet % ROOT], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_multi_thread_entropy_is_bit_exact():
    _run(CPU_SNIPPET, {"DK_ENTROPY_THREADS": "5"})
    _run(CPU_SNIPPET, {"DK_ENTROPY_THREADS": "4"})
    _run(CPU_SNIPPET, {"DK_ENTROPY_THREADS": "2"})
    _run(CPU_SNIPPET, {"DK_ENTROPY_THREADS": "1"})


@pytest.mark.parametrize("threads", ["1", "2", "4", "5"])
def test_entropy_error_paths_return_codes(threads):
    _run(CPU_ERROR_SNIPPET, {"DK_ENTROPY_THREADS": threads})


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"DK_XCD": "0"}, {"DK_DIGIT_PLANE": "0"}, {"DK_DIGIT_PLANE": "2"}, {"DK_PLATEAU": "0"}, {"DK_BWT_CARRY": "0"},
                                 {"DK_PLATEAU": "0", "DK_BWT_CARRY": "0"}, {"DK_PAIR_CHAINS": "0"}, {"DK_PAIR_CHAINS": "1", "DK_BWT_CARRY": "0"},
                                 {"DK_LFIRST": "0"}, {"DK_LFIRST": "2"}, {"DK_LFIRST": "2", "DK_LF_MAX": "32"}, {"DK_LF_SWITCH": "40"},
                                 {"DK_LF_SWITCH": "100"}, {"DK_LF_MEDIUM": "0"}, {"DK_LFIRST": "2", "DK_LF_MEDIUM": "0"},
                                 {"DK_CHAIN_GROUP": "2"}, {"DK_CHAIN_GROUP": "3"}, {"DK_LF_SHORT_SLOTS": "0"}, {"DK_LF_SHORT_SLOTS": "100000000", "DK_LF_SHORT_STEPS": "2"},
                                 {"DK_LF_FORK": "0"}, {"DK_LF_FORK": "100000000", "DK_LF_REFORK": "64"}, {"DK_LF_WAVE": "0"}, {"DK_LF_TAIL": "1000000"},
                                 {"DK_LF_STEPS": "2", "DK_LF_STUCK": "1"}, {"DK_PACKED_SORT": "0"}, {"DK_PACKED_SORT": "1", "DK_NARROW_KEYS": "0"},
                                 {"DK_POISON": "165"}, {"DK_POISON": "255", "DK_LFIRST": "2"}])
def test_gpu_variants_match_oracle(env):
    _run(GPU_SNIPPET, env, tuning=True)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["0", "1", "2", "3"])
def test_gpu_prefix_paths_match_oracle(mode):
    _run(PREFIX_SNIPPET, {"DK_PREFIX": mode, "DK_TRACE": "1"}, tuning=True)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["2", "3"])
def test_gpu_prefix_paths_with_lfirst_forced(mode):
    """the L-first BWT path behind shortened (and narrow) keys, which the product never takes (csrc/suffix_array.hip: nothing survives such a sort)"""
    _run(PREFIX_SNIPPET, {"DK_PREFIX": mode, "DK_LFIRST": "2"}, tuning=True)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"DK_PERIOD": "0"}, {"DK_PERIOD": "1"}, {"DK_PERIOD": "2"}, {"DK_PERIOD": "2", "DK_PREFIX": "2"},
                                 {"DK_PERIOD": "2", "DK_BWT_CARRY": "0"}, {"DK_PERIOD": "2", "DK_LFIRST": "0"}])
def test_gpu_period_round_variants(env):
    """runs, stretches of period 1..8 with ties between them, whole-block periods with one odd byte, blocks ending inside a run
    (tools/period_check.py): suffix array, BWT and origin against the oracle with the period round off, as the product decides, and forced
    -- also behind a four-symbol initial sort (periods above four are then out of reach) and with L gathered from the suffix array"""
    e = dict(os.environ)
    e.update(env)
    e["DARK_AMD_LIB"] = TUNING_LIB
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "period_check.py"), "3", "15"], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "wrong results: 0" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    if env == {"DK_PERIOD": "0"}:
        assert "period_round" not in out.stdout
    else:
        assert "period_round" in out.stdout
