"""CPU tests of the oracle (oracle/dark_oracle.c): pinned against the reference's known answers and its
test inputs, and against the committed golden vectors.  No GPU, no product code."""
import numpy as np
import pytest

from conftest import seeded_inputs


def test_known_answers_saca_rs_411(orc, vectors):
    # /root/reference/src/saca.rs:409-413 `detailed`
    for v in vectors["reference"]["saca_rs_411_412"]:
        t = v["input"].encode()
        sa = orc.sa_sais(t)
        assert list(sa) == v["sa"]
        assert list(orc.sa_naive(t)) == v["sa"]
        bwt, origin = orc.bwt_forward(t, sa)
        assert bwt.tobytes() == v["bwt"].encode() and origin == v["origin"]
        assert orc.bwt_inverse(bwt, origin).tobytes() == t


def test_roundtrips_license(orc, license_bytes):
    # /root/reference/src/saca.rs:429-433 `roundtrips`
    sa = orc.sa_sais(license_bytes)
    assert (sa == orc.sa_naive(license_bytes)).all()
    bwt, origin = orc.bwt_forward(license_bytes, sa)
    assert orc.bwt_inverse(bwt, origin).tobytes() == license_bytes


@pytest.mark.parametrize("name", ["abracababra", "LICENSE"])
def test_golden_stage_vectors(orc, vectors, license_bytes, name):
    data = b"abracababra" if name == "abracababra" else license_bytes
    g = vectors["oracle"][name]
    sa = orc.sa_sais(data)
    assert list(sa) == g["sa"]
    bwt, origin = orc.bwt_forward(data, sa)
    assert bwt.tobytes().hex() == g["bwt_hex"] and origin == g["origin"]
    dc = orc.dc_encode(bwt)
    assert list(dc["init"]) == g["dc_init"] and list(dc["d"]) == g["dc_d"]
    assert dc["sym"].tobytes().hex() == g["dc_sym_hex"] and dc["rank"].tobytes().hex() == g["dc_rank_hex"]
    assert orc.block_dc_encode("rawdc", data).hex() == g["rawdc_records_hex"]
    for m, hx in g["streams_hex"].items():
        assert orc.block_dc_encode(m, data).hex() == hx
        assert orc.block_dc_decode(m, bytes.fromhex(hx), len(data)) == data


def test_block_dc_roundtrips_like_reference(orc, license_bytes):
    # /root/reference/src/block/dc.rs:187-192 `roundtrips` (exp on abracababra + LICENSE, ybs on LICENSE)
    for model, data in [("exp", b"abracababra"), ("exp", license_bytes), ("ybs", license_bytes),
                        ("dark", license_bytes), ("simple", license_bytes)]:
        s = orc.block_dc_encode(model, data)
        assert orc.block_dc_decode(model, s, len(data)) == data


@pytest.mark.parametrize("model", ["dark", "exp", "simple", "ybs"])
def test_model_roundtrips_like_reference(orc, vectors, model):
    # /root/reference/src/model/mod.rs:112-120 `roundtrips_dc`: fixed 4-tuple vector, then 1000 random
    # (dist < 200, sym) pairs (the reference seeds from the OS; here the seed is fixed)
    g = vectors["oracle"]["model_mod_rs_113"]
    s = orc.model_encode(model, g["d"], g["sym"])
    assert s.hex() == g["streams_hex"][model]
    assert list(orc.model_decode(model, s, g["sym"])) == g["d"]
    rng = np.random.default_rng(7)
    d = rng.integers(0, 200, size=1000, dtype=np.uint32)
    sym = rng.integers(0, 256, size=1000, dtype=np.uint8)
    s = orc.model_encode(model, d, sym)
    assert (orc.model_decode(model, s, sym) == d).all()


def test_dark_model_large_distances(orc):
    rng = np.random.default_rng(3)
    d = (rng.integers(0, 2**31 - 2, size=500, dtype=np.int64) >> rng.integers(0, 31, size=500)).astype(np.uint32)
    sym = rng.integers(0, 256, size=500, dtype=np.uint8)
    s = orc.model_encode("dark", d, sym)
    assert (orc.model_decode("dark", s, sym) == d).all()


def test_entropy_ari_range_roundtrip(orc, vectors):
    # /root/reference/src/entropy/ari.rs:76-107 `roundtrip`
    g = vectors["oracle"]["entropy_ari_range"]
    bits = [(b >> i) & 1 for b in g["bytes"] for i in range(8)]
    flat = [g["flat"]] * len(bits)
    s = orc.bitcoder_encode(bits, flat)
    assert s.hex() == g["stream_hex"]
    assert list(orc.bitcoder_decode(s, flat)) == bits
    rng = np.random.default_rng(5)
    flat = rng.integers(1, 4095, size=5000, dtype=np.uint16)
    bits = (rng.integers(0, 4096, size=5000) >= flat).astype(np.uint8)
    s = orc.bitcoder_encode(bits, flat)
    assert (orc.bitcoder_decode(s, flat) == bits).all()
    assert len(s) < 5000 / 8 * 1.2 + 8  # adaptive-free sanity: coded size near entropy bound


def test_seeded_inputs_all_stages(orc):
    for t in seeded_inputs(seed=11, count=60):
        n = len(t)
        if n == 1:
            # the reference panics on a 1-byte block: `assert!(n1+n1 <= input.len())` saca.rs:300 with n1 == 1
            with pytest.raises(orc.OracleError, match="code -4"):
                orc.sa_sais(t)
            continue
        sa = orc.sa_sais(t)
        assert (sa == orc.sa_naive(t)).all()
        bwt, origin = orc.bwt_forward(t, sa)
        assert (orc.bwt_inverse(bwt, origin) == t).all()
        dc = orc.dc_encode(bwt)
        runs = 1 + int((bwt[1:] != bwt[:-1]).sum())
        assert len(dc["d"]) == runs  # one distance per run of the BWT
        if not (bwt == 255).any():
            back, used = orc.dc_decode(dc["init"], dc["d"], n)
            assert (back == bwt).all()
            assert used == (0 if len(set(bwt.tolist())) == 1 else runs)


def test_header_quirks(orc):
    # A7: symbol 0xFF is never transmitted (block/dc.rs:57,60,73,127) -> such blocks encode but do not decode
    t = np.frombuffer(b"hello\xffworld\xff\xffagain", np.uint8)
    s = orc.block_dc_encode("dark", t)
    with pytest.raises(orc.OracleError):
        orc.block_dc_decode("dark", s, len(t))
    # one-symbol block: dc::decode skips the sweep distance, origin is mis-read, output truncated (code +1)
    s = orc.block_dc_encode("dark", b"aaaaaaa")
    with pytest.raises(orc.OracleError, match="code 1"):
        orc.block_dc_decode("dark", s, 7)


def test_saca_storage_words(orc):
    # /root/reference/src/saca.rs:353-354
    L = orc.lib()
    assert L.orc_saca_storage_words(768771) == 768771 + 256 + 768771 // 4
    assert L.orc_saca_storage_words(1000) == 1000 + 256 + 500
    assert L.orc_saca_storage_words(10**6) == 10**6 + 256 + 250000


def test_dc_distances_agree_with_dark_c_analogue(orc):
    # /root/reference/etc/dark-c/src/ptax.cpp:61-111 (Ptax::Perform) is the same author's earlier DC: at every run start `cp` of a
    # symbol whose previous run ended at `lp` it stores r[lp] = cp - lp - arm - 1, `arm` being the symbol's move-to-front rank.
    # It keeps different side information (per-symbol run counts instead of a final sweep), but wherever it defines a distance
    # the restated compress::bwt::dc::encode must agree.  The analogue is restated here in plain Python (explicit MTF list).
    def analogue(data):
        n = len(data)
        r = {}
        last, mtf, cp = {}, [], 0
        while cp < n:
            cs = data[cp]
            if cs in last:
                arm = mtf.index(cs)
                r[last[cs]] = cp - last[cs] - arm - 1
                mtf.remove(cs)
            mtf.insert(0, cs)
            while cp < n and data[cp] == cs:
                cp += 1
            last[cs] = cp - 1
        return r, last, mtf

    for t in seeded_inputs(seed=71, count=40, max_n=1500):
        data = t.tolist()
        r, last, mtf = analogue(data)
        dc = orc.dc_encode(t)
        sparse = dc["sparse"]
        n = len(t)
        for pos, d in r.items():
            assert sparse[pos] == d, (pos, d, sparse[pos])
        # every other entry the oracle holds is the final sweep: one per symbol, at its last position, n - last - rank - 1
        extra = {int(p): int(sparse[p]) for p in np.flatnonzero(sparse != n) if int(p) not in r}
        assert extra == {last[s]: n - last[s] - rank - 1 for rank, s in enumerate(mtf)}


def test_fullsize_helpers_on_small_blocks(orc):
    """tests/oracle_jobs.py (background oracle runs) and the checkers of tests/test_gpu_fullsize.py, exercised on the CPU at toy sizes:
    a correct suffix array passes, a swapped pair and a duplicated index are caught."""
    torch = pytest.importorskip("torch")
    import oracle_jobs
    import test_gpu_fullsize as fs
    oracle_jobs.start("small_random")
    job = oracle_jobs.get("small_text")
    assert job["stream"] == orc.block_dc_encode("dark", job["block"]) and job["origin"] < len(job["block"])
    assert oracle_jobs.get("small_random")["stream"] is None
    for t in (job["block"], np.frombuffer(b"ab" * 9000 + b"a", np.uint8), np.zeros(3000, np.uint8)):
        n = len(t)
        sa = orc.sa_sais(t)
        d_t, d_sa = torch.from_numpy(t.copy()), torch.from_numpy(sa.view(np.int32).copy())
        fs.assert_permutation(d_sa, n)
        fs.assert_all_neighbours_in_order(d_t, d_sa, n, t, chunk=4096, max_steps=50)  # few steps: the leftovers go to the host compare
        bad = d_sa.clone()
        bad[[n // 2, n // 2 + 1]] = bad[[n // 2 + 1, n // 2]]
        with pytest.raises(AssertionError):
            fs.assert_all_neighbours_in_order(d_t, bad, n, t, chunk=4096, max_steps=50)
        dup = d_sa.clone()
        dup[5] = dup[6]
        with pytest.raises(AssertionError):
            fs.assert_permutation(dup, n)


def test_bbb_state_tables_agree(orc):
    """The bit-history table of the bbb model exists three times: the reference's literal (src/model/bbb.rs:34-99), the oracle's copy
    (oracle/bbb_state_table.inc, generated from that literal) and the product's (dark_amd/csrc/bbb_states.inc, typed in separately).
    Oracle and product must not share source, so their agreement is checked here -- against each other always, against the
    reference's literal wherever the reference tree is present (not on the GPU box)."""
    import os
    import re
    from conftest import ROOT
    theirs = orc.bbb_state_table()
    text = open(os.path.join(ROOT, "dark_amd", "csrc", "bbb_states.inc")).read()
    nums = [int(x) for x in re.findall(r"\d+", re.sub(r"//[^\n]*", "", text))]
    assert len(nums) == 1024
    assert np.array_equal(np.array(nums, dtype=np.uint8).reshape(256, 4), theirs)
    if os.path.exists("/root/reference/src/model/bbb.rs"):
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_bbb_table", os.path.join(ROOT, "oracle", "make_bbb_table.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        assert np.array_equal(np.array(mod.parse_state_table("/root/reference"), dtype=np.uint8), theirs)
