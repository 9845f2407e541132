"""Suffix sort + BWT of a Zipf-word text: words of a 200 000-word vocabulary drawn with Zipf(1.15) frequencies, spaces, line ends, 3 % copied
segments -- a more natural distribution of group sizes than the order-3 Markov stand-in of the bench (half of all suffixes sit in groups of
more than 1024 members after the initial sort).  Prints per-kernel times of one profiled step and two plain timings (DK_TRACE=1 with the
tuning build shows the rounds).  usage: python tools/zipf_text.py [N [LETTERS]]   (LETTERS = 26, or e.g. 180 for an alphabet that needs 8-bit codes)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import dark_amd

from dark_amd.datagen import word_like as zipf_text  # the bench workload wordlike_1e8 is zipf_text(100_000_000, alpha=180)

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
alpha = int(sys.argv[2]) if len(sys.argv) > 2 else 26
t0 = time.time(); t = zipf_text(n, alpha=alpha); print("gen %.1f s, n=%d, sigma=%d" % (time.time() - t0, len(t), len(np.unique(t))), flush=True)
n = len(t)
d = torch.from_numpy(t).cuda(); out = torch.empty(n, dtype=torch.uint8, device="cuda")
with dark_amd.Context(n) as ctx:
    ctx.dev_bwt_forward(d, n, out)  # warm-up: first launches load the code
    for prof in (True, False, False):
        ctx.set_profiling(prof); ctx.stats_reset()
        ctx.dev_bwt_forward(d, n, out)
        st = ctx.stats()
        print("profiled" if prof else "plain", "ms_sa %.2f rounds %d passes %d" % (st["ms_sa"] + st["ms_bwt"], st["rounds"], st["sort_passes"]),
              {k[2:]: round(v["ms"], 2) for k, v in sorted(st["kernels"].items(), key=lambda x: -x[1]["ms"])[:12]} if prof else "", flush=True)
