"""The period round (suffix_array.hip, k_period_*): inputs made of runs and short-period stretches, SA + BWT + origin against the oracle (TEST
INFRASTRUCTURE use, like tests/), with the route printed.   python tools/period_check.py [seed] [cases]
With DARK_AMD_LIB=<tuning build> DK_PERIOD=2 the round is forced wherever the probe finds a single periodic window."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def make_case(rng, kind, n):
    if kind == 0:    # runs of one value of many lengths between random bytes of a small alphabet
        sigma = int(rng.integers(2, 6))
        out = []
        total = 0
        while total < n:
            if rng.random() < 0.5:
                ln = int(rng.choice([1, 2, 7, 8, 9, 63, 64, 65, 200, 1000, 5000]))
                out.append(np.full(ln, rng.integers(0, sigma), np.uint8))
            else:
                ln = int(rng.integers(1, 40))
                out.append(rng.integers(0, sigma, ln, dtype=np.uint8))
            total += ln
        return np.concatenate(out)[:n]
    if kind == 1:    # stretches of a random period 1..8 (the same period strings come back: ties between stretches), text in between
        units = [rng.integers(97, 100, int(rng.integers(1, 9)), dtype=np.uint8) for _ in range(4)]
        out, total = [], 0
        while total < n:
            if rng.random() < 0.6:
                u = units[int(rng.integers(0, 4))]
                ln = int(rng.choice([10, 64, 100, 640, 3000, 20000]))
                out.append(np.tile(u, ln // len(u) + 2)[int(rng.integers(0, len(u))):][:ln])
            else:
                ln = int(rng.integers(1, 100))
                out.append(rng.integers(97, 123, ln, dtype=np.uint8))
            total += ln
        return np.concatenate(out)[:n]
    if kind == 2:    # one period over the whole block, one odd byte somewhere (or none)
        u = rng.integers(0, 3, int(rng.integers(1, 9)), dtype=np.uint8)
        t = np.tile(u, n // len(u) + 1)[:n].copy()
        if rng.random() < 0.7:
            t[int(rng.integers(0, n))] = rng.integers(0, 4)
        return t
    if kind == 3:    # zero-padded records: 90 % zeros
        t = np.zeros(n, np.uint8)
        k = n // 10
        t[rng.integers(0, n, k)] = rng.integers(1, 256, k, dtype=np.uint8)
        return t
    if kind == 5:    # one long period (9 .. 5000) over the whole block, a few odd bytes (long-period search, token round once the depth covers it)
        u = rng.integers(0, 256, int(rng.choice([9, 12, 100, 1000, 4999])), dtype=np.uint8)
        t = np.tile(u, n // len(u) + 1)[:n].copy()
        for _ in range(int(rng.integers(0, 4))):
            t[int(rng.integers(0, n))] = rng.integers(0, 256)
        return t
    # kind 4: bytes 0 and 255 (the token's direction at both ends of the alphabet), runs ending at the block's end
    t = rng.choice(np.array([0, 255], np.uint8), n, p=[0.9, 0.1])
    t[-int(rng.integers(1, 300)):] = rng.choice(np.array([0, 255], np.uint8))
    return t


def main():
    import torch, dark_amd
    from oracle import orc
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rng = np.random.default_rng(seed)
    bad = 0
    with dark_amd.Context(1 << 23) as ctx:
        for c in range(ncases):
            kind = c % 6
            n = int(rng.choice([5000, 70000, 300000, 1 << 20, 3_000_000, (1 << 22) + 77]))
            t = np.ascontiguousarray(make_case(rng, kind, n))
            n = len(t)
            d = torch.from_numpy(t).cuda()
            d_sa = torch.empty(n, dtype=torch.int32, device="cuda"); d_bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
            ctx.dev_suffix_array(d, n, d_sa); r_sa = sorted(ctx.stats()["routes"]); rounds_sa = ctx.stats()["rounds"]
            origin = ctx.dev_bwt_forward(d, n, d_bwt); st = ctx.stats()
            want = orc.sa_sais(t); wb, wo = orc.bwt_forward(t, want)
            ok_sa = bool((d_sa.cpu().numpy().view(np.uint32) == want).all())
            ok_bwt = origin == wo and bool((d_bwt.cpu().numpy() == np.frombuffer(wb, np.uint8)).all())
            print("case %2d kind %d n=%8d  SA %s (rounds %d, %s)  BWT %s (%.2f ms, rounds %d, %s)" % (c, kind, n, "ok" if ok_sa else "WRONG", rounds_sa, ",".join(r_sa),
                  "ok" if ok_bwt else "WRONG", st["ms_sa"] + st["ms_bwt"], st["rounds"], ",".join(sorted(st["routes"]))), flush=True)
            bad += (not ok_sa) + (not ok_bwt)
    print("wrong results:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
