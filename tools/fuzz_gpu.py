"""One-off randomized campaign on the GPU box: suffix array, BWT + origin, DC arrays and whole-block streams against the oracle, on
inputs built to cross the thresholds of the suffix sort (prefix probe, text rounds, late rank array, big groups) and of the distance
coder (narrow / wide tiles, chunks with many first occurrences).  python tools/fuzz_gpu.py [cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import dark_amd
from oracle import orc


def make(rng):
    n = int(rng.integers(1 << 22, 6_000_000)) if rng.random() < 0.7 else int(rng.integers(1000, 300_000))
    sigma = int(rng.choice([2, 3, 4, 5, 8, 16, 30, 64, 100, 200, 255]))
    kind = rng.choice(["iid", "skewed", "repeats", "runs", "mixed_tiles", "periodic"])
    alphabet = rng.choice(255, size=sigma, replace=False).astype(np.uint8)  # never 0xFF: the container cannot carry it
    if kind == "iid":
        t = alphabet[rng.integers(0, sigma, size=n)]
    elif kind == "skewed":
        p = rng.dirichlet(np.full(sigma, 0.3))
        t = alphabet[rng.choice(sigma, size=n, p=p)]
    elif kind == "repeats":
        t = alphabet[rng.integers(0, sigma, size=n)]
        for _ in range(int(rng.integers(1, 6))):
            ln = int(rng.integers(100, max(200, n // 20)))
            src = int(rng.integers(0, n - ln))
            dst = int(rng.integers(0, n - ln))
            t[dst:dst + ln] = t[src:src + ln].copy()
    elif kind == "runs":
        lens = rng.geometric(0.2, size=n // 3 + 10)
        syms = alphabet[rng.integers(0, sigma, size=len(lens))]
        t = np.repeat(syms, lens)[:n]
    elif kind == "mixed_tiles":  # stretches of small and large alphabets: DC tiles switch between the narrow and the wide route
        parts = []
        while sum(map(len, parts)) < n:
            k = int(rng.choice([2, 4, sigma]))
            parts.append(alphabet[rng.integers(0, min(k, sigma), size=int(rng.integers(1000, 50_000)))])
        t = np.concatenate(parts)[:n]
    else:
        period = alphabet[rng.integers(0, sigma, size=int(rng.integers(1, 5000)))]
        t = np.tile(period, n // len(period) + 1)[:n].copy()
        for _ in range(int(rng.integers(0, 4))):
            t[int(rng.integers(0, n))] = alphabet[0]
    return kind, sigma, np.ascontiguousarray(t)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    with dark_amd.Context(6 << 20) as ctx:
        for c in range(cases):
            kind, sigma, t = make(rng)
            n = len(t)
            tag = "case %d: %s sigma=%d n=%d" % (c, kind, sigma, n)
            sa = orc.sa_sais(t) if n > 1 else np.zeros(1, np.uint32)
            assert (ctx.suffix_array(t) == sa).all(), tag + " SA"
            wb, wo = orc.bwt_forward(t, sa)
            bwt, origin = ctx.bwt_forward(t)
            assert origin == wo and (bwt == np.frombuffer(wb, np.uint8)).all(), tag + " BWT"
            want = orc.dc_encode(wb)
            got = ctx.dc_encode(bwt)
            for key in ("init", "d", "sym", "rank"):
                assert (np.asarray(got[key]) == np.asarray(want[key])).all(), tag + " DC " + key
            if len(np.unique(t)) > 1:
                stream = ctx.block_encode("dark", t)
                assert stream == orc.block_dc_encode("dark", t), tag + " stream"
                assert bytes(ctx.block_decode("dark", stream, n)) == t.tobytes(), tag + " roundtrip"
            print("%s ok (%.0f s)" % (tag, time.time() - t0), flush=True)
    print("ok")


if __name__ == "__main__":
    main()
