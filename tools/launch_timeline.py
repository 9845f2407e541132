"""Warm-up for a launch timeline of one workload's suffix sort + BWT: run under DK_TRACE=1 DK_TRACE_LAUNCHES=1 on the tuning build (DARK_AMD_LIB), every
bracketed launch is printed with its slot, stream and time.   python tools/launch_timeline.py [workload, default realtext_5e7]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, dark_amd
from dark_amd import datagen
t = datagen.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "realtext_5e7"]()
d = torch.from_numpy(t).cuda(); out = torch.empty(len(t), dtype=torch.uint8, device="cuda")
with dark_amd.Context(len(t)) as ctx:
    for _ in range(3):
        ctx.dev_bwt_forward(d, len(t), out)
    torch.cuda.synchronize()
    st = ctx.stats()
    print("ms", st["ms_sa"] + st["ms_bwt"])
