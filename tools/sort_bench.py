"""Micro-benchmark of the radix sort engine on random (u64, u32) pairs; prints per-kernel HIP-event times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dark_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(1)
keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2)
vals = np.arange(n, dtype=np.uint32)
ctx = dark_amd.Context(n)
ctx.dbg_sort_pairs(keys[:1 << 20], vals[:1 << 20], 0, bits)
ctx.stats_reset()
ctx.set_profiling(True)
for _ in range(2):
    k2, v2 = ctx.dbg_sort_pairs(keys, vals, 0, bits)
st = ctx.stats()
for name, k in sorted(st["kernels"].items(), key=lambda x: -x[1]["ms"]):
    print("%-18s launches %4d  total %8.3f ms  avg %8.1f us  %7.1f GB/s" % (name, k["launches"], k["ms"], 1e3 * k["ms"] / k["launches"], k["bytes"] / k["ms"] / 1e6))
chk = np.argsort(keys[:2_000_000] & np.uint64((1 << bits) - 1 if bits < 64 else 0xFFFFFFFFFFFFFFFF), kind="stable")
k3, v3 = ctx.dbg_sort_pairs(keys[:2_000_000], vals[:2_000_000], 0, bits)
print("check:", bool((v3 == vals[:2_000_000][chk]).all()))
