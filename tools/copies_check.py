"""dk_suffix_array on two identical halves, three identical thirds and a block put together from copies of copies: time, rounds, route; with the
oracle's suffix array beside it unless `nocheck` is given.   python tools/copies_check.py [bytes per half, default 5e7] [nocheck]"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, dark_amd
from dark_amd import datagen
from oracle import orc

def cases(half_n):
    half = datagen.wiki_like(half_n, 2)
    third = datagen.wiki_like(2 * half_n // 3, 3)
    rng = np.random.default_rng(5)
    # copies of copies: a base text, then twelve chunks copied from anywhere before (so copies of copies occur), some edits in between
    parts = [datagen.wiki_like(half_n // 2, 7)]
    total = len(parts[0])
    while total < 2 * half_n:
        cur = np.concatenate(parts)
        ln = int(rng.integers(half_n // 16, half_n // 4))
        at = int(rng.integers(0, max(1, len(cur) - ln)))
        parts.append(cur[at:at + ln].copy())
        parts.append(datagen.wiki_like(int(rng.integers(10, 2000)), int(rng.integers(1, 1 << 30))))
        total += len(parts[-1]) + len(parts[-2])
    coc = np.concatenate(parts)[:2 * half_n]
    return {"two halves": np.concatenate([half, half]), "three thirds": np.concatenate([third, third, third]), "copies of copies": coc}

def main():
    half_n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
    check = (sys.argv[2] != "nocheck") if len(sys.argv) > 2 else True
    cs = cases(half_n)
    cap = max(len(t) for t in cs.values())
    with dark_amd.Context(cap) as ctx:
        for name, t in cs.items():
            t = np.ascontiguousarray(t)
            n = len(t)
            d_in = torch.from_numpy(t).cuda()
            d_sa = torch.empty(n, dtype=torch.int32, device="cuda")
            ctx.dev_suffix_array(d_in, n, d_sa)
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.dev_suffix_array(d_in, n, d_sa)
                dt = 1e3 * (time.perf_counter() - t0)
                best = dt if best is None else min(best, dt)
            st = ctx.stats()
            res = {"case": name, "n": n, "suffix_array_ms": round(best, 2), "rounds": st["rounds"], "routes": sorted(st["routes"]), "sort_passes": st["sort_passes"]}
            if check:
                want = orc.sa_sais(t)
                res["equal_to_oracle"] = bool((d_sa.cpu().numpy().view(np.uint32) == want).all())
            print(json.dumps(res), flush=True)
            del d_in, d_sa

main()
