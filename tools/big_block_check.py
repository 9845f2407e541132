"""A 300 MB text block (above 2^27 suffixes: the L-first path, the LDS-window inverse permutation's third split level, the five-stage host coder on
300 M distances) against the oracle: BWT + origin, then encode -> decode.   python tools/big_block_check.py   (TEST INFRASTRUCTURE use of the oracle)"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch, dark_amd
from dark_amd import datagen
from oracle import orc
n = 300_000_000
t = datagen.wiki_like(n, 23)
t0 = time.time(); wb, wo = orc.bwt_forward(t); print("oracle %.1f s" % (time.time() - t0), flush=True)
d = torch.from_numpy(t).cuda(); out = torch.empty(n, dtype=torch.uint8, device="cuda")
with dark_amd.Context(n) as ctx:
    for _ in range(2):
        origin = ctx.dev_bwt_forward(d, n, out)
    st = ctx.stats()
    got = out.cpu().numpy()
    print("n=%d origin %d/%d diffs %d  sa+bwt %.2f ms  routes %s  ws peak %.1f n" % (n, origin, wo, int((got != np.frombuffer(wb, np.uint8)).sum()), st["ms_sa"] + st["ms_bwt"], sorted(st["routes"]), st["ws_peak_bytes"] / n))
    s = ctx.dev_block_encode("dark", d, n).copy()
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    ctx.dev_block_decode("dark", s, n, back)
    print("round trip", bool(torch.equal(back, d)), "stream", len(s))
