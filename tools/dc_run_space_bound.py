"""k_dc_main on the BWT of the text block, and on the same BWT with every run cut to one symbol (what a run-space kernel would see per lane)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, dark_amd
from dark_amd import datagen
which = sys.argv[1] if len(sys.argv) > 1 else "enwik8_like_1e8"
t = datagen.WORKLOADS[which]()
n = len(t)
d = torch.from_numpy(t).cuda(); L = torch.empty(n, dtype=torch.uint8, device="cuda")
with dark_amd.Context(n) as ctx:
    ctx.dev_bwt_forward(d, n, L)
    keep = torch.ones(n, dtype=torch.bool, device="cuda"); keep[1:] = L[1:] != L[:-1]
    R = L[keep].contiguous(); m = R.numel()
    print(which, "n", n, "runs", m, "positions per run %.2f" % (n / m))
    for name, x in (("L", L), ("runs only", R)):
        k = x.numel()
        dist = torch.empty(k, dtype=torch.int32, device="cuda"); sym = torch.empty(k, dtype=torch.uint8, device="cuda")
        ctx.dev_dc_encode(x, k, dist, sym)
        ctx.set_profiling(True); ctx.stats_reset()
        for _ in range(3): ctx.dev_dc_encode(x, k, dist, sym)
        st = ctx.stats(); ctx.set_profiling(False)
        print("  %-10s" % name, {kk: round(v["ms"] / 3, 3) for kk, v in st["kernels"].items() if kk.startswith("k_dc")})
