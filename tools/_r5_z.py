import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, dark_amd
from dark_amd import datagen
half = datagen.wiki_like(50_000_000, 2)
t = np.ascontiguousarray(np.concatenate([half, np.zeros(3000, np.uint8), half[:50_000_000 - 3000][::-1]]))
n = len(t)
d = torch.from_numpy(t).cuda(); out = torch.empty(n, dtype=torch.uint8, device="cuda")
with dark_amd.Context(n) as ctx:
    for _ in range(2):
        ctx.dev_bwt_forward(d, n, out); st = ctx.stats()
        print("bwt %.2f ms" % (st["ms_sa"] + st["ms_bwt"]), st["rounds"], sorted(st["routes"]))
    ctx.set_profiling(True); ctx.stats_reset()
    ctx.dev_bwt_forward(d, n, out)
    st = ctx.stats()
    for k, v in sorted(st["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:12]:
        print("   %-22s %4d launches %8.3f ms" % (k, v["launches"], v["ms"]))
