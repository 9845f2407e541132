"""Per-kernel-slot times of one suffix sort + BWT (dk_dev_bwt_forward) of a bench workload, from the library's own HIP-event pairs:
    python tools/kernel_breakdown.py [workload, default wordlike_1e8]      (DARK_AMD_LIB / DK_* select a build and its knobs)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import dark_amd
from dark_amd import datagen

which = sys.argv[1] if len(sys.argv) > 1 else "wordlike_1e8"
t = datagen.WORKLOADS[which]()
d = torch.from_numpy(t).cuda(); out = torch.empty(len(t), dtype=torch.uint8, device="cuda")
with dark_amd.Context(len(t)) as ctx:
    ms = []
    for _ in range(3):
        ctx.dev_bwt_forward(d, len(t), out); st = ctx.stats(); ms.append(st["ms_sa"] + st["ms_bwt"])
    print(which, "suffix sort + BWT, no event pairs: %.2f ms" % sorted(ms)[1], "rounds", st["rounds"], "passes", st["sort_passes"], sorted(st["routes"]))
    ctx.set_profiling(True); ctx.stats_reset()
    ctx.dev_bwt_forward(d, len(t), out)
    st = ctx.stats()
    for k, v in sorted(st["kernels"].items(), key=lambda kv: -kv[1]["ms"]):
        print("   %-22s %4d launches %8.3f ms" % (k, v["launches"], v["ms"]))
    print("with event pairs %.2f ms, sum of kernel slots %.2f ms" % (st["ms_sa"] + st["ms_bwt"], sum(v["ms"] for v in st["kernels"].values())))
