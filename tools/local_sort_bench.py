"""Round-4 experiment (the MSD-first hybrid): what would its LOCAL pass cost?  Every 8192-pair tile of a device array
of (u64 key, u32 value) pairs is sorted by itself inside one workgroup's LDS on `bits` low bits (k_radix_sort_small on a grid), timed with
HIP events, beside full LSD passes of the global sort over the same pairs.   python tools/local_sort_bench.py [N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import dark_amd
from dark_amd import _lib

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
keys = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
vals = torch.arange(n, dtype=torch.int32, device="cuda")
lib = _lib.load()
with dark_amd.Context(n) as ctx:
    def run(fn, bits):
        k, v = keys.clone(), vals.clone()
        torch.cuda.synchronize()
        ctx.set_profiling(True); ctx.stats_reset()
        rc = fn(ctx._h, k.data_ptr(), v.data_ptr(), n, 0, bits)
        assert rc == 0, rc
        st = ctx.stats(); ctx.set_profiling(False)
        return {kk[2:]: round(vv["ms"], 3) for kk, vv in st["kernels"].items()}, k, v
    for bits in (8, 16, 24, 32):
        loc, k, v = run(lib.dk_dbg_dev_local_sort, bits)
        # every tile must be sorted on those bits
        kk = (k & ((1 << bits) - 1)).view(-1)
        tiles = kk[: (n // 8192) * 8192].view(-1, 8192)
        ok = bool((tiles[:, 1:] >= tiles[:, :-1]).all().item())
        glob, _, _ = run(lib.dk_dbg_dev_sort_pairs, bits)
        print("n=%d bits=%d  local tiles: %s (sorted: %s)   global LSD: %s  sum %.3f ms" % (n, bits, loc, ok, glob, sum(glob.values())), flush=True)
