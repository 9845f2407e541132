"""CPU tests of the product library: it loads, exports every symbol include/dark_amd.h declares, refuses to run without a
GPU, and its host entropy stage (range coder, models, stream layout, dc::decode, bitwise coder) is bit-exact with the oracle.
No device compute is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import dark_amd
from dark_amd import _lib, entropy, model
from conftest import ROOT, seeded_inputs

MODELS = ("dark", "exp", "ybs", "simple")


def header_functions():
    text = open(os.path.join(ROOT, "include", "dark_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = dark_amd.load_library()
    names = header_functions()
    assert len(names) >= 28
    for name in names:
        assert hasattr(lib, name), "libdark_amd.so lacks %s" % name
        assert name in _lib.SIGNATURES, "python binding lacks %s" % name
    assert sorted(_lib.SIGNATURES) == names
    assert b"gfx950" in lib.dk_version()


def test_no_cpu_backend():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dark_amd.DarkError) as e:
        dark_amd.Context(1000)
    assert e.value.code == _lib.DK_E_NODEVICE
    lib = dark_amd.load_library()
    h = C.c_void_p()
    assert lib.dk_ctx_create(-1, 1000, C.byref(h)) == _lib.DK_E_NODEVICE  # -1 ("CPU") is not a backend
    assert lib.dk_ctx_create(0, 0, C.byref(h)) == _lib.DK_E_ARG


@pytest.mark.parametrize("m", MODELS)
def test_model_streams_match_oracle(orc, vectors, m):
    # src/model/mod.rs:112-120 roundtrips_dc: the fixed 4-tuple vector, then 1000 random (dist < 200, sym) pairs
    g = vectors["oracle"]["model_mod_rs_113"]
    assert model.encode(m, g["d"], g["sym"]).hex() == g["streams_hex"][m]
    rng = np.random.default_rng(13)
    for size, hi in ((1000, 200), (5000, 1 << 16), (3000, 1 << 23)):
        d = rng.integers(0, hi, size=size, dtype=np.uint32)
        sym = rng.integers(0, 256, size=size, dtype=np.uint8)
        s = model.encode(m, d, sym)
        assert s == orc.model_encode(m, d, sym)
        assert (model.decode(m, s, sym) == d).all()


def test_dark_model_wide_distances(orc):
    rng = np.random.default_rng(17)
    d = (rng.integers(0, 2**31 - 2, size=4000, dtype=np.int64) >> rng.integers(0, 31, size=4000)).astype(np.uint32)
    d[:4] = [0, 1, 2**31 - 2, 2**30]
    sym = rng.integers(0, 256, size=4000, dtype=np.uint8)
    s = model.encode("dark", d, sym)
    assert s == orc.model_encode("dark", d, sym)
    assert (model.decode("dark", s, sym) == d).all()
    with pytest.raises(dark_amd.DarkError):  # dist+1 would need 32 significant bits: dark.rs:132 has 32 mantissa rows
        model.encode("dark", [2**31 - 1], [0])


@pytest.mark.parametrize("name", ["abracababra", "LICENSE"])
def test_entropy_stage_against_golden(orc, vectors, license_bytes, name):
    data = b"abracababra" if name == "abracababra" else license_bytes
    g = vectors["oracle"][name]
    bwt = bytes.fromhex(g["bwt_hex"])
    dc = orc.dc_encode(bwt)
    ends = np.flatnonzero(dc["sparse"] != len(data)).astype(np.uint32)
    for m in MODELS:
        s = model.stream_encode(m, len(data), dc["init"], dc["d"], dc["sym"], g["origin"])
        assert s.hex() == g["streams_hex"][m]
        b2, o2, single = model.stream_decode(m, s, len(data))
        assert b2.tobytes() == bwt and o2 == g["origin"] and not single
    rec = model.stream_encode("rawdc", len(data), dc["init"], dc["d"], dc["sym"], g["origin"], rank=dc["rank"], run_end=ends)
    assert rec.hex() == g["rawdc_records_hex"]


def test_entropy_stage_seeded(orc):
    for t in seeded_inputs(seed=23, count=40):
        n = len(t)
        if n == 1:
            continue
        bwt, origin = orc.bwt_forward(t)
        dc = orc.dc_encode(bwt)
        for m in MODELS:
            s = model.stream_encode(m, n, dc["init"], dc["d"], dc["sym"], origin)
            assert s == orc.block_dc_encode_bwt(m, bwt, origin)
            if (bwt == 255).any():
                continue  # header cannot carry symbol 0xFF (block/dc.rs:57-73): undecodable by design
            b2, o2, single = model.stream_decode(m, s, n)
            assert (b2 == bwt).all()
            assert single == (len(set(bwt.tolist())) == 1)
            if not single:
                assert o2 == origin


def test_consumed_equals_written_also_for_one_symbol_blocks(orc):
    # records of a multi-block file are walked by "bytes consumed"; for a one-symbol block the reference's dc::decode returns
    # before the sweep distance the encoder wrote, so the decoder must read and drop it or the next record header is missed by 2-4 bytes
    cases = [np.zeros(1000, np.uint8), np.full(7, 65, np.uint8), np.frombuffer(b"abracadabra", np.uint8), np.array([3, 3], np.uint8),
             np.frombuffer(b"ab" * 50, np.uint8)]
    for t in cases:
        n = len(t)
        for m in MODELS:
            s = orc.block_dc_encode(m, t)
            bwt, origin = orc.bwt_forward(t)
            b2, o2, single, used = model.stream_decode(m, s + b"\x5a" * 9, n, with_consumed=True)  # trailing bytes = the next record
            assert used == len(s), (m, n, used, len(s))
            assert (b2 == bwt).all() and o2 == origin
            assert single == (len(set(t.tolist())) == 1)
    with pytest.raises(dark_amd.DarkError):  # nothing but 0xFF: no symbol reaches the decoder -> error, not a block of zeros
        t = np.full(50, 255, np.uint8)
        model.stream_decode("dark", orc.block_dc_encode("dark", t), 50)


def test_dc_decode_host(orc):
    lib = dark_amd.load_library()
    for t in seeded_inputs(seed=29, count=30):
        if (t == 255).any():
            continue
        dc = orc.dc_encode(t)  # any byte string is a valid "BWT" for DC
        init = np.ascontiguousarray(dc["init"], dtype=np.uint32)
        d = np.ascontiguousarray(dc["d"], dtype=np.uint32)
        out = np.empty(len(t), dtype=np.uint8)
        used = C.c_size_t(0)
        rc = lib.dk_dc_decode(None, init.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), len(d),
                              out.ctypes.data_as(C.c_void_p), len(t), C.byref(used))
        assert rc == 0 and (out == t).all()
        ref, ref_used = orc.dc_decode(dc["init"], dc["d"], len(t))
        assert used.value == ref_used


def test_stream_errors():
    with pytest.raises(dark_amd.DarkError) as e:
        model.stream_decode("dark", b"\x00\x01", 100)
    assert e.value.code == _lib.DK_E_STREAM
    with pytest.raises(dark_amd.DarkError) as e:
        model.encode("nope", [1], [1])
    assert e.value.code == _lib.DK_E_MODEL
    init = np.full(256, 1 << 25, dtype=np.uint32)
    init[65] = 0
    with pytest.raises(dark_amd.DarkError) as e:  # exp codes 24 bits only (exp.rs:22,67): refuse n > 2^24
        model.stream_encode("exp", 1 << 25, init, [(1 << 25) - 1], [65], 3)
    assert e.value.code == _lib.DK_E_MODEL


def test_bitcoder_matches_oracle(orc, vectors):
    g = vectors["oracle"]["entropy_ari_range"]
    bits = [(b >> i) & 1 for b in g["bytes"] for i in range(8)]
    assert entropy.encode_bits(bits, [g["flat"]] * 32).hex() == g["stream_hex"]
    rng = np.random.default_rng(31)
    flat = rng.integers(1, 4095, size=20000, dtype=np.uint16)
    bits = (rng.integers(0, 4096, size=20000) >= flat).astype(np.uint8)
    s = entropy.encode_bits(bits, flat)
    assert s == orc.bitcoder_encode(bits, flat)
    assert (entropy.decode_bits(s, flat) == bits).all()


def test_property_model_streams(orc):
    # hypothesis: arbitrary (distance, symbol) streams -- the product's host coder and the oracle must produce the same bytes,
    # and decoding must give the stream back, for every model within its documented distance range
    from hypothesis import given, settings, strategies as st

    limits = {"dark": 2**31 - 2, "exp": 2**24 - 1, "ybs": 2**29 - 1, "simple": 2**24 + 254}

    @settings(max_examples=60, deadline=None)
    @given(st.sampled_from(MODELS), st.lists(st.tuples(st.integers(0, 31), st.integers(0, 2**31), st.integers(0, 255)), min_size=0, max_size=300))
    def check(m, items):
        d = np.array([min((1 << b) - 1 + (r % (1 << b) if b else 0), limits[m]) for b, r, _ in items], dtype=np.uint32)
        sym = np.array([s for _, _, s in items], dtype=np.uint8)
        got = model.encode(m, d, sym)
        assert got == orc.model_encode(m, d, sym)
        assert (model.decode(m, got, sym) == d).all()

    check()


def test_reciprocal_division_is_exact():
    """The four-stage encoder replaces r = span / total by the high half of span * ceil(2^64 / total) (entropy.cpp, UEvent).  The claim:
    exact for every span < 2^32 and total in [2, 2^15).  Checked here for every total against edge spans and random ones."""
    import random
    rng = random.Random(5)
    for total in range(2, 1 << 15):
        inv = (((1 << 64) - 1) // total + 1) if total & (total - 1) else ((1 << 63) // total * 2)
        assert inv == -(-(1 << 64) // total)  # ceil(2^64 / total), as the C code computes it
        spans = [0, 1, total - 1, total, total + 1, (1 << 32) - 1, (1 << 32) - total, ((1 << 32) // total) * total - 1,
                 ((1 << 32) // total) * total] + [rng.randrange(1 << 32) for _ in range(6)]
        for span in spans:
            if 0 <= span < (1 << 32):
                assert (span * inv) >> 64 == span // total, (span, total)


def test_cli_names_follow_set_extension():
    # std::path::PathBuf::set_extension as used at src/main.rs:64-66,96-98
    from dark_amd import cli
    for name, want in (("book1", "book1.dark"), ("book.txt", "book.dark"), ("a.tar.gz", "a.tar.dark"), (".hidden", ".hidden.dark"),
                       ("dir/sub/x.bin", "x.dark"), ("trailing.", "trailing.dark")):
        assert cli.output_name(name, "dark") == want, name
    assert cli.output_name("book.dark", "orig") == "book.orig"
    assert cli.has_extension("x/book.dark", "dark") and not cli.has_extension(".dark", "dark") and not cli.has_extension("dark", "dark")


def test_bbb_raw_stream_matches_oracle(orc, license_bytes):
    """block::raw with the bbb model (src/block/raw.rs:45-58,85-97, src/model/bbb.rs): product coder == oracle restatement, byte for
    byte, and both round-trip.  PARITY UNPINNED: both build the gates after etc/bbb/main.cpp (assumptions G1-G5 in oracle/dark_oracle.c)."""
    rng = np.random.default_rng(37)
    cases = [np.frombuffer(license_bytes, np.uint8), np.zeros(300, np.uint8), np.frombuffer(b"abracadabra" * 30, np.uint8),
             rng.integers(0, 256, size=5000, dtype=np.uint8), rng.integers(0, 4, size=20000, dtype=np.uint8),
             np.repeat(rng.integers(0, 50, size=400, dtype=np.uint8), rng.integers(1, 40, size=400)), np.array([7, 7], np.uint8)]
    for t in cases:
        t = np.ascontiguousarray(t)
        bwt, origin = orc.bwt_forward(t)
        s = model.raw_stream_encode(bwt, origin)
        assert s == orc.raw_bbb_encode(bwt, origin), len(t)
        b2, o2, used = model.raw_stream_decode(s + b"\x11" * 5, len(t))
        assert (b2 == bwt).all() and o2 == origin and used == len(s)
        assert orc.raw_bbb_decode(s, len(t)) == t.tobytes()
    with pytest.raises(dark_amd.DarkError):
        model.raw_stream_decode(b"\x00\x01", 50)
    with pytest.raises(dark_amd.DarkError) as e:
        model.raw_stream_encode(b"abc", 0, raw_model=0)  # the dump model codes nothing: not a stream model
    assert e.value.code == _lib.DK_E_MODEL


def _gated_encode(n, init, dist, sym, origin, ready, stall_ms, threads):
    lib = _lib.load()
    out = np.empty(8 * len(dist) + 8192, dtype=np.uint8)
    ln = C.c_size_t(0)
    rc = lib.dk_dbg_stream_encode_gated(0, n, init.ctypes.data, dist.ctypes.data, sym.ctypes.data, len(dist), origin, out.ctypes.data, len(out),
                                        C.byref(ln), C.addressof(ready), stall_ms, threads)
    return rc, out[:ln.value].tobytes()


@pytest.mark.parametrize("threads", [1, 2, 4, 5])
def test_stream_that_is_still_arriving(threads):
    """ADVICE r3 (abi.cpp:113): the host coder of a large block starts while the distance stream is still on its way from the GPU and waits
    at a frontier the HIP stream's host functions move.  Here another thread moves the frontier: in steps (same bytes as the finished
    stream, whatever the thread form), not at all (the coder gives up with DK_E_HIP after stall_ms), or to the poison value."""
    import threading
    import time
    rng = np.random.default_rng(12)
    m = 300_000
    n = 1 << 22
    dist = (rng.integers(0, 1 << 20, size=m, dtype=np.uint32) >> rng.integers(0, 20, size=m).astype(np.uint32)).astype(np.uint32)
    sym = rng.integers(0, 40, size=m, dtype=np.uint8)
    init = np.full(256, n, dtype=np.uint32)
    init[:40] = np.arange(40)
    want = model.stream_encode("dark", n, init, dist, sym, 77)
    # the frontier moves in 16 steps
    ready = C.c_size_t(0)

    def feeder():
        for c in range(16):
            time.sleep(0.002)
            ready.value = m * (c + 1) // 16
    t = threading.Thread(target=feeder)
    t.start()
    rc, got = _gated_encode(n, init, dist, sym, 77, ready, 5000, threads)
    t.join()
    assert rc == 0 and got == want
    # the frontier stalls half way: every thread of the pipeline must come back
    ready = C.c_size_t(m // 2)
    t0 = time.perf_counter()
    rc, _ = _gated_encode(n, init, dist, sym, 77, ready, 150, threads)
    assert rc == _lib.DK_E_HIP and time.perf_counter() - t0 < 20
    # the producer gives up
    ready = C.c_size_t(m // 4)

    def poisoner():
        time.sleep(0.01)
        ready.value = (1 << 64) - 1
    t = threading.Thread(target=poisoner)
    t.start()
    rc, _ = _gated_encode(n, init, dist, sym, 77, ready, 5000, threads)
    t.join()
    assert rc == _lib.DK_E_HIP
