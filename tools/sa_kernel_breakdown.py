"""Per-kernel-slot times of one dk_suffix_array call (the suffix-array path, no L-first): python tools/sa_kernel_breakdown.py halves|thirds|text|tile
(two identical 50 MB halves, three thirds, the 1e8 text block; `tile`: the BWT route of the tile-crossing-groups test shape, with DK_TRACE on the tuning build)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, dark_amd
from dark_amd import datagen
which = sys.argv[1]
if which == "halves":
    half = datagen.wiki_like(50_000_000, 2); t = np.concatenate([half, half])
elif which == "thirds":
    th = datagen.wiki_like(33_333_333, 3); t = np.concatenate([th, th, th])
elif which == "text":
    t = datagen.WORKLOADS["enwik8_like_1e8"]()
elif which == "tile":
    rng = np.random.default_rng(77)
    seg = rng.integers(97, 123, size=40, dtype=np.uint8)
    parts = []
    for k in range(9000):
        parts += [rng.integers(0, 256, size=int(rng.integers(20, 60)), dtype=np.uint8), np.array([66 if k == 0 else 65], np.uint8), seg]
    t = np.concatenate(parts + [rng.integers(0, 256, size=100_000, dtype=np.uint8)])
    t[t == 255] = 0
n = len(t)
d = torch.from_numpy(np.ascontiguousarray(t)).cuda()
with dark_amd.Context(n) as ctx:
    if which == "tile":
        out = torch.empty(n, dtype=torch.uint8, device="cuda")
        ctx.dev_bwt_forward(d, n, out); print(sorted(ctx.stats()["routes"]))
        sys.exit(0)
    sa = torch.empty(n, dtype=torch.int32, device="cuda")
    ms = []
    for _ in range(3):
        ctx.dev_suffix_array(d, n, sa); st = ctx.stats(); ms.append(st["ms_sa"])
    print(which, "suffix array, no event pairs: %.2f ms" % sorted(ms)[1], "rounds", st["rounds"], "passes", st["sort_passes"], sorted(st["routes"]))
    ctx.set_profiling(True); ctx.stats_reset()
    ctx.dev_suffix_array(d, n, sa)
    st = ctx.stats()
    for k, v in sorted(st["kernels"].items(), key=lambda kv: -kv[1]["ms"]):
        print("   %-22s %4d launches %8.3f ms" % (k, v["launches"], v["ms"]))
    print("with event pairs %.2f ms, sum of kernel slots %.2f ms" % (st["ms_sa"], sum(v["ms"] for v in st["kernels"].values())))
