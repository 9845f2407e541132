"""Quick parity pass of the L-first BWT path on the GPU box: BWT + origin against the oracle on inputs of every flavour, with the route
taken and the time.  python tools/lf_check.py [N]   (DK_TRACE=1 + DARK_AMD_LIB=.../libdark_amd_tuning.so shows the rounds)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import dark_amd
from dark_amd import datagen
from oracle import orc

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
rng = np.random.default_rng(7)
def planted(t, seg_len, copies):
    t = t.copy(); seg = t[1000:1000 + seg_len].copy()
    for _ in range(copies):
        o = int(rng.integers(0, len(t) - seg_len)); t[o:o + seg_len] = seg
    return t
cases = [
    ("wiki_like", datagen.wiki_like(n, 7)),
    ("word_like 180", datagen.word_like(n, 5)),
    ("word_like 26", datagen.word_like(n, 6, alpha=26)),
    ("acgt", datagen.acgt(n, 8)),
    ("acgt + repeats", planted(datagen.acgt(n, 5), 20_000, 4)),
    ("random", datagen.random_bytes(n, 9)),
    ("random + 3 copies of 50 KB", planted(datagen.random_bytes(n, 9), 50_000, 3)),
    ("two identical halves", np.concatenate([datagen.wiki_like(n // 2, 10)] * 2)),
    ("three copies", np.concatenate([datagen.wiki_like(n // 3, 11)] * 3)),
    ("ab period", np.frombuffer(b"ab" * (n // 2) + b"a", np.uint8)),
    ("a^n b", np.concatenate([np.full(n - 1, 97, np.uint8), np.array([98], np.uint8)])),
    ("zero-heavy", np.where(rng.random(n) < 0.9, 0, rng.integers(0, 7, size=n)).astype(np.uint8)),
    ("ends in zeros", np.concatenate([rng.integers(0, 256, size=n - 3000, dtype=np.uint8), np.zeros(3000, np.uint8)])),
    ("zeros then text then zeros", np.concatenate([np.zeros(5000, np.uint8), datagen.wiki_like(n - 10000, 3), np.zeros(5000, np.uint8)])),
    ("period 1000", np.tile(rng.integers(0, 256, size=1000, dtype=np.uint8), n // 1000 + 1)[:n].copy()),
    ("text with a 400-byte run", np.concatenate([datagen.wiki_like(n // 2, 13), np.full(400, 65, np.uint8), datagen.wiki_like(n // 2, 14)])),
    ("text with 300 x abc", np.concatenate([datagen.wiki_like(n // 2, 15), np.frombuffer(b"abc" * 300, np.uint8), datagen.wiki_like(n // 2, 16)])),
    ("text with runs of 100..500", (lambda t: [t.__setitem__(slice(o, o + int(l)), 32) for o, l in zip(rng.integers(0, n - 600, 200), rng.integers(100, 500, 200))] and t)(datagen.wiki_like(n, 17).copy())),
    ("small text 70000", datagen.wiki_like(70_000, 4)),
    ("tiny", rng.integers(0, 3, size=70, dtype=np.uint8)),
]
bad = 0
with dark_amd.Context(max(len(t) for _, t in cases)) as ctx:
    for name, t in cases:
        t = np.ascontiguousarray(t)
        wb, wo = orc.bwt_forward(t)
        t0 = time.perf_counter()
        bwt, origin = ctx.bwt_forward(t)
        dt = time.perf_counter() - t0
        st = ctx.stats()
        ok = origin == wo and (np.frombuffer(bwt, np.uint8) == np.frombuffer(wb, np.uint8)).all()
        nd = int((np.frombuffer(bwt, np.uint8) != np.frombuffer(wb, np.uint8)).sum())
        print("%-28s n=%-9d %s origin %d/%d diffs %d  sa+bwt %.2f ms rounds %d routes %s" % (name, len(t), "ok " if ok else "BAD", origin, wo, nd, st["ms_sa"] + st["ms_bwt"],
              st["rounds"], sorted(st["routes"])), flush=True)
        bad += 0 if ok else 1
print("ok" if not bad else "%d BAD" % bad)
sys.exit(1 if bad else 0)
