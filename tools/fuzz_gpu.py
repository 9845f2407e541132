"""One-off randomized campaign on the GPU box: suffix array, BWT + origin, DC arrays and whole-block streams against the oracle, on
inputs built to cross the thresholds of the suffix sort (prefix probe, text rounds, late rank array, big groups) and of the distance
coder (narrow / wide tiles, chunks with many first occurrences).  python tools/fuzz_gpu.py [cases] [seed]
python tools/fuzz_gpu.py dc [cases] [seed]: the DC stage alone on byte arrays of mixed segments (hundreds of cases per minute)."""
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
    kind = rng.choice(["iid", "skewed", "repeats", "runs", "mixed_tiles", "periodic", "copies"])
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
    elif kind == "copies":  # whole passages two to seven times, copies of parts of copies: groups of 2 .. 7 suffixes with long common prefixes (pair chains, the walks through bigger groups)
        p = rng.dirichlet(np.full(sigma, 0.5))
        parts = [alphabet[rng.choice(sigma, size=max(64, n // int(rng.integers(3, 12))), p=p)]]
        total = len(parts[0])
        while total < n:
            cur = np.concatenate(parts)
            ln = int(rng.integers(1, max(2, min(len(cur), n // 2))))
            at = int(rng.integers(0, len(cur) - ln + 1))
            parts.append(cur[at:at + ln].copy())
            if rng.random() < 0.5:
                parts.append(alphabet[rng.choice(sigma, size=int(rng.integers(1, 300)), p=p)])
            total = sum(map(len, parts))
        t = np.concatenate(parts)[:n]
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


def make_dc_input(rng):
    """A byte array for the DC stage alone (any byte array is a valid input): segments of different alphabets and run lengths, so
    that tiles and 64-position chunks of every kind follow each other -- wide / narrow matching, with / without the bitmap, chunks
    without a run start, previous occurrences inside / before the window."""
    n = int(rng.integers(1, 40)) if rng.random() < 0.05 else int(rng.integers(100, 1_500_000))
    parts, total = [], 0
    while total < n:
        seg = int(rng.choice([1, 17, 64, 100, 1000, 4096, 5000, 20000, 100000]))
        sigma = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 16, 21, 22, 40, 64, 100, 128, 129, 200, 256]))
        lo = int(rng.integers(0, 257 - sigma))
        mean_run = float(rng.choice([1, 1, 1, 2, 5, 50, 3000]))
        if mean_run == 1:
            part = rng.integers(lo, lo + sigma, size=seg)
        else:
            lens = rng.geometric(1.0 / mean_run, size=seg // max(1, int(mean_run)) + 2)
            part = np.repeat(rng.integers(lo, lo + sigma, size=len(lens)), lens)[:seg]
        if rng.random() < 0.2:  # periodic
            per = rng.integers(lo, lo + sigma, size=int(rng.integers(1, 70)))
            part = np.tile(per, seg // len(per) + 1)[:seg]
        parts.append(part)
        total += len(part)
    return np.ascontiguousarray(np.concatenate(parts)[:n], dtype=np.uint8)


def main_dc(cases, seed):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    with dark_amd.Context(2 << 20) as ctx:
        for c in range(cases):
            L = make_dc_input(rng)
            want = orc.dc_encode(L)
            got = ctx.dc_encode(L)
            for key in ("init", "d", "sym", "rank"):
                a, b = np.asarray(got[key]), np.asarray(want[key])
                if a.shape != b.shape or not (a == b).all():
                    np.save("gpurun_out/dc_fuzz_fail_%d_%d.npy" % (seed, c), L) if os.path.isdir("gpurun_out") else None
                    raise AssertionError("dc case %d seed %d n=%d: %s differs" % (c, seed, len(L), key))
            if c % 50 == 0:
                print("dc case %d n=%d ok (%.0f s)" % (c, len(L), time.time() - t0), flush=True)
    print("ok")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "dc":  # python tools/fuzz_gpu.py dc [cases] [seed]: the DC stage alone, many more shapes
        return main_dc(int(sys.argv[2]) if len(sys.argv) > 2 else 300, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
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
