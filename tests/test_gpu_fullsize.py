"""Full-size GPU checks at the block sizes BASELINE.json names, through size-independent properties (the oracle would take
minutes to hours at these sizes): encode -> decode round trips, forward -> inverse BWT round trips, BWT = permutation of the
text, suffix-array spot checks (sampled neighbours are in order, via direct comparison of the text), DC -> rebuild round trip.
Data stays on the GPU; only small samples come back."""
import numpy as np
import pytest

import dark_amd
from dark_amd import datagen

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _sa_spot_check(text_np, d_sa, n, rng, samples=2000):
    # neighbouring suffixes must be in order; compare them on the host from the text (bounded prefix, long enough for these inputs)
    pos = rng.integers(0, n - 1, size=samples)
    pairs = torch.stack([d_sa[torch.from_numpy(pos).cuda()], d_sa[torch.from_numpy(pos + 1).cuda()]], dim=1).cpu().numpy().astype(np.int64)
    for a, b in pairs:
        assert a != b
        la, lb = text_np[a:a + 20000].tobytes(), text_np[b:b + 20000].tobytes()
        assert la < lb or (la == lb and (a > b or len(la) == 20000)), (a, b)


def _histogram(t):
    return torch.bincount(t.to(torch.int64), minlength=256)


@pytest.mark.parametrize("workload", ["enwik8_like_1e8", "acgt_2p28", "enwik9_block_125e6", "random_2p30"])
def test_fullsize_properties(workload):
    rng = np.random.default_rng(7)
    if workload == "enwik8_like_1e8":
        block = datagen.wiki_like(100_000_000, 2)
    elif workload == "acgt_2p28":
        block = datagen.acgt(1 << 28, 3)
    elif workload == "enwik9_block_125e6":
        block = datagen.wiki_like(125_000_000, 41)  # BASELINE configs[3]: one of the 8 blocks (seed 40 is in test_gpu_configs.py)
    else:
        block = datagen.random_bytes(1 << 30, 50)
    n = len(block)
    d_in = torch.from_numpy(block).cuda()
    with dark_amd.Context(n) as ctx:
        # suffix array: a permutation whose sampled neighbours are in order
        d_sa = torch.empty(n, dtype=torch.int32, device="cuda")
        ctx.dev_suffix_array(d_in, n, d_sa)
        st = ctx.stats()
        sa64 = d_sa.to(torch.int64) & 0xFFFFFFFF
        assert int(sa64.sum().item()) == n * (n - 1) // 2  # checksum of a permutation of 0..n-1
        assert int(sa64.min().item()) == 0 and int(sa64.max().item()) == n - 1
        _sa_spot_check(block, sa64, n, rng)
        del sa64, d_sa
        # BWT forward / inverse round trip; L is a permutation of T
        d_bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
        origin = ctx.dev_bwt_forward(d_in, n, d_bwt)
        assert torch.equal(_histogram(d_bwt), _histogram(d_in))
        d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
        ctx.dev_bwt_inverse(d_bwt, n, origin, d_back)
        assert torch.equal(d_back, d_in)
        del d_back
        # DC: one entry per run; the host rebuild (dc::decode) gives the BWT back (skipped for data with byte 0xFF: the
        # reference's header cannot carry that symbol, src/block/dc.rs:57-73)
        d_dist = torch.empty(n, dtype=torch.int32, device="cuda")
        d_sym = torch.empty(n, dtype=torch.uint8, device="cuda")
        init, m = ctx.dev_dc_encode(d_bwt, n, d_dist, d_sym)
        runs = 1 + int((d_bwt[1:] != d_bwt[:-1]).sum().item())
        assert m == runs
        if workload != "random_2p30":
            back, used = ctx.dc_decode(init, d_dist[:m].cpu().numpy().view(np.uint32), n)
            assert used == m and torch.equal(torch.from_numpy(back).cuda(), d_bwt)
        del d_dist, d_sym, d_bwt
        # whole block: encode -> decode
        if workload == "enwik8_like_1e8":
            stream = ctx.dev_block_encode("dark", d_in, n).copy()
            d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
            ctx.dev_block_decode("dark", stream, n, d_out)
            assert torch.equal(d_out, d_in)
        print(workload, "rounds", st["rounds"], "sort passes", st["sort_passes"], "sa ms", round(st["ms_sa"], 2))


@pytest.mark.skipif(not __import__("os").environ.get("DK_TEST_MAXSIZE"), reason="set DK_TEST_MAXSIZE=1: needs ~230 GB of HBM and a few minutes")
def test_largest_block_the_index_type_allows():
    """n = 2^31 - 2 (the largest block u32 ranks + h can carry): ACGT with planted long repeats, so that the short-prefix path, the late
    rank array and doubling rounds all run at full size."""
    rng = np.random.default_rng(9)
    n = (1 << 31) - 2
    block = datagen.acgt(n, 11)
    seg = block[12345:12345 + 300_000].copy()
    for off in (1 << 30, (1 << 31) - 400_000, 777_777_777):
        block[off:off + len(seg)] = seg
    d_in = torch.from_numpy(block).cuda()
    with dark_amd.Context(n) as ctx:
        d_sa = torch.empty(n, dtype=torch.int32, device="cuda")
        ctx.dev_suffix_array(d_in, n, d_sa)
        st = ctx.stats()
        total = 0
        for lo in range(0, n, 1 << 28):  # checksum of a permutation of 0..n-1, in slices (keeps the int64 copy small)
            total += int((d_sa[lo:lo + (1 << 28)].to(torch.int64) & 0xFFFFFFFF).sum().item())
        assert total == n * (n - 1) // 2
        pos = rng.integers(0, n - 1, size=1500)
        a = (d_sa[torch.from_numpy(pos).cuda()].to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
        b = (d_sa[torch.from_numpy(pos + 1).cuda()].to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
        for x, y in zip(a, b):
            lx, ly = block[x:x + 400_000].tobytes(), block[y:y + 400_000].tobytes()
            assert lx < ly or (lx == ly and x > y), (x, y)
        del d_sa
        d_bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
        origin = ctx.dev_bwt_forward(d_in, n, d_bwt)
        d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
        ctx.dev_bwt_inverse(d_bwt, n, origin, d_back)
        assert torch.equal(d_back, d_in)
        print("n = 2^31 - 2: rounds", st["rounds"], "sort passes", st["sort_passes"], "sa ms", round(st["ms_sa"], 2))
