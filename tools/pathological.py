"""Suffix sort + BWT on pathological inputs (long runs, periodic text, Fibonacci words, two identical halves ...): correctness against
the oracle and timings -- the worst cases of prefix doubling, which the reference's SA-IS (linear time: README.md:12, src/saca.rs:5-6)
does not have.

    python tools/pathological.py [small|full] [out.json]

small: the 4-16 MB cases rounds 1 and 2 measured (gpurun_out/patho.log, r2_patho.log).  full: the same plus 1e8-byte cases -- period-2
text and two identical 50 MB halves of wiki-like text.  One JSON object per case on stdout (and collected into out.json): MB/s of the
device suffix sort + BWT (input resident in HBM; `suffix_array_ms`: the suffix array itself, dk_dev_suffix_array), rounds, radix passes, and whether SA equals the oracle's (TEST INFRASTRUCTURE use of
the oracle: this is a checker tool, like tests/)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dark_amd  # noqa: E402
from dark_amd import datagen  # noqa: E402
from oracle import orc  # noqa: E402


def fib_word(n):
    a, b = b"a", b"ab"
    while len(b) < n:
        a, b = b, b + a
    return np.frombuffer(b[:n], np.uint8)


def cases(mode):
    out = {
        "ab*2^23": lambda: np.frombuffer(b"ab" * (1 << 23), np.uint8),
        "fib-like 2^22": lambda: fib_word(1 << 22),
        "a^n b 2^22": lambda: np.concatenate([np.zeros((1 << 22) - 1, np.uint8), np.ones(1, np.uint8)]),
        "period 1000 x 8000": lambda: np.tile(np.random.default_rng(1).integers(0, 256, 1000, dtype=np.uint8), 8000),
    }
    if mode == "full":
        half = datagen.wiki_like(50_000_000, 2)
        out.update({
            "ab*5e7 (1e8 bytes, period 2)": lambda: np.frombuffer(b"ab" * 50_000_000, np.uint8),
            "two identical 50 MB halves (1e8 bytes)": lambda: np.concatenate([half, half]),
            "a^n b 1e8": lambda: np.concatenate([np.zeros(100_000_000 - 1, np.uint8), np.ones(1, np.uint8)]),
            "period 1000 x 100000 (1e8 bytes)": lambda: np.tile(np.random.default_rng(1).integers(0, 256, 1000, dtype=np.uint8), 100_000),
            "text with a 3000-byte run of zeros (1e8 bytes)": lambda: np.concatenate([half, np.zeros(3000, np.uint8), half[:50_000_000 - 3000][::-1]]),
        })
    return out


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "small"
    results = []
    with dark_amd.Context(100_000_000 if mode == "full" else 1 << 24) as ctx:
        for name, make in cases(mode).items():
            t = np.ascontiguousarray(make())
            n = len(t)
            d_in = torch.from_numpy(t).cuda()
            d_sa = torch.empty(n, dtype=torch.int32, device="cuda")
            d_bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
            ctx.dev_suffix_array(d_in, n, d_sa)  # warm
            best, st = None, None
            for _ in range(3):
                t0 = time.perf_counter()
                origin = ctx.dev_bwt_forward(d_in, n, d_bwt)
                dt = time.perf_counter() - t0
                if best is None or dt < best:
                    best, st = dt, ctx.stats()
            t_sa0 = time.perf_counter()
            ctx.dev_suffix_array(d_in, n, d_sa)
            sa_ms = 1e3 * (time.perf_counter() - t_sa0)
            sa_st = ctx.stats()
            t1 = time.perf_counter()
            want = orc.sa_sais(t)
            dto = time.perf_counter() - t1
            want_bwt, want_origin = orc.bwt_forward(t, want)
            ok = bool((d_sa.cpu().numpy().view(np.uint32) == want).all()) and origin == want_origin and bool((d_bwt.cpu().numpy() == want_bwt).all())
            res = {"case": name, "bytes": n, "gpu_ms": round(1e3 * best, 2), "gpu_MBps": round(n / best / 1e6, 1), "rounds": st["rounds"],
                   "bwt_routes": sorted(st["routes"]), "suffix_array_ms": round(sa_ms, 2), "suffix_array_rounds": sa_st["rounds"],
                   "sort_passes": st["sort_passes"], "oracle_sais_s": round(dto, 2), "oracle_MBps": round(n / dto / 1e6, 1), "equal_to_oracle": ok}
            print(json.dumps(res), flush=True)
            results.append(res)
            del d_in, d_sa, d_bwt
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump({"tool": "tools/pathological.py " + mode, "results": results}, f, indent=1)
    if not all(r["equal_to_oracle"] for r in results):
        sys.exit(1)


if __name__ == "__main__":
    main()
