"""Suffix array and BWT of many small and medium blocks against the oracle with the workspace POISONED (tuning build, DK_POISON=<byte>: every workspace allocation is
filled with that byte first): a read of memory that nobody has written gives a wrong answer every time instead of once in a while.
    DARK_AMD_LIB=dark_amd/libdark_amd_tuning.so DK_POISON=165 python tools/poison_check.py [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import dark_amd
from dark_amd import datagen
from oracle import orc

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
cases = []
for n in (2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 65536 + 1, 100_003, 300_000):
    for sigma in (2, 3, 5, 26, 256):
        cases.append(("iid n=%d sigma=%d" % (n, sigma), rng.integers(0, sigma, size=n, dtype=np.uint8)))
base = datagen.wiki_like(400_000, 31)
cases.append(("two halves", np.concatenate([base[:150_000]] * 2)))
cases.append(("three copies", np.concatenate([base[:100_000]] * 3)))
cases.append(("five copies", np.concatenate([base[:60_000]] * 5)))
t = base.copy(); t[200_000:230_000] = t[10_000:40_000]; cases.append(("text with a repeat", t))
t = base[:200_000].copy(); t[50_000:53_000] = 65; cases.append(("text with a run", t))
cases.append(("period 7", np.tile(rng.integers(0, 200, size=7, dtype=np.uint8), 30_000)))
bad = 0
with dark_amd.Context(1 << 20) as ctx:
    for rep in range(2):
        for name, t in cases:
            t = np.ascontiguousarray(t)
            want = orc.sa_sais(t)
            got = ctx.suffix_array(t)
            wb, wo = orc.bwt_forward(t, want)
            bwt, origin = ctx.bwt_forward(t)
            ok_sa = bool((got == want).all()); ok_bwt = origin == wo and bool((np.frombuffer(bwt, np.uint8) == np.frombuffer(wb, np.uint8)).all())
            if ok_bwt and rep == 0:  # the stages behind the BWT as well: inverse, distance coding, the coded block both ways
                back = ctx.bwt_inverse(bwt, origin)
                dc_w, dc_g = orc.dc_encode(wb), ctx.dc_encode(bwt)
                ok_bwt = bytes(back) == t.tobytes() and all((np.asarray(dc_g[k]) == np.asarray(dc_w[k])).all() for k in ("init", "d", "sym", "rank"))
                if len(np.unique(t)) > 1 and not (t == 255).any():
                    stream = ctx.block_encode("dark", t)
                    ok_bwt = ok_bwt and stream == orc.block_dc_encode("dark", t) and bytes(ctx.block_decode("dark", stream, len(t))) == t.tobytes()
            if not (ok_sa and ok_bwt):
                bad += 1
                d = np.flatnonzero(got != want)
                print("BAD", name, "rep", rep, "sa", ok_sa, "bwt", ok_bwt, "first sa diff", d[:3], "of", len(d), sorted(ctx.stats()["routes"]), flush=True)
print("poison", os.environ.get("DK_POISON"), "cases", 2 * len(cases), "bad", bad)
sys.exit(1 if bad else 0)
