"""Suffix sort + BWT of a Zipf-word text: words of a 200 000-word vocabulary drawn with Zipf(1.15) frequencies, spaces, line ends, 3 % copied
segments -- a more natural distribution of group sizes than the order-3 Markov stand-in of the bench (half of all suffixes sit in groups of
more than 1024 members after the initial sort).  Prints per-kernel times of one profiled step and two plain timings (DK_TRACE=1 with the
tuning build shows the rounds).  usage: python tools/zipf_text.py [N [LETTERS]]   (LETTERS = 26, or e.g. 180 for an alphabet that needs 8-bit codes)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import dark_amd

def zipf_text(n, seed=5, vocab=200_000, alpha=26):
    rng = np.random.default_rng(seed)
    lens = rng.integers(2, 11, size=vocab)
    letters = (rng.integers(0, alpha, size=int(lens.sum())) + (97 if alpha <= 26 else 48)).astype(np.uint8)  # alpha > 128: codes of 8 bits, like real enwik8
    offs = np.concatenate([[0], np.cumsum(lens)])
    # Zipf ranks
    nwords = n // 5
    ranks = np.minimum(rng.zipf(1.15, size=nwords) - 1, vocab - 1)
    wl = lens[ranks] + 1
    total = int(wl.sum())
    out = np.full(total, 32, dtype=np.uint8)
    starts = np.concatenate([[0], np.cumsum(wl)[:-1]])
    # fill word letters (vectorised by word length)
    for L in range(2, 11):
        sel = np.nonzero(lens[ranks] == L)[0]
        if len(sel) == 0: continue
        src = offs[ranks[sel]][:, None] + np.arange(L)[None, :]
        dst = starts[sel][:, None] + np.arange(L)[None, :]
        out[dst.reshape(-1)] = letters[src.reshape(-1)]
    out = out[:n]
    # sentence structure + 3 % copied segments
    out[rng.integers(0, len(out), size=len(out) // 80)] = 10
    budget = len(out) // 33
    while budget > 0:
        ln = int(rng.integers(64, 4097)); s = int(rng.integers(0, len(out) - ln)); d = int(rng.integers(0, len(out) - ln))
        out[d:d + ln] = out[s:s + ln].copy(); budget -= ln
    return out

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
