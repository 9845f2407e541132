"""A/B of whole-block encodes between builds of the library (where did the host coding stage lose 5-12 %?).

usage: python tools/ab_entropy.py [--steps K] [--rounds R] [--workload W] NAME=PATH[:threads=MODE] ...

Every NAME=PATH is a libdark_amd.so of some commit (built in a worktree, copied under tools/_ab/).  The libraries are loaded side by side
(RTLD_LOCAL) and take turns, R rounds of K steps each of dk_dev_block_encode on the same device-resident block, inside ONE process on ONE
box: wall-clock per step and the library's own ms_entropy (the first eight doubles of dk_stats have not moved since round 1).
`:threads=MODE` calls dk_set_entropy_threads(MODE) before that library's turn where the symbol exists (0 = automatic)."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from dark_amd import datagen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--workload", default="enwik8_like_1e8")
    ap.add_argument("--decode", action="store_true", help="time dk_dev_block_decode of the stream as well (ms_entropy of the decode = header + range decoder + model + rebuild of L)")
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    block = datagen.WORKLOADS[args.workload]()
    n = len(block)
    d_in = torch.from_numpy(block).cuda()
    torch.cuda.synchronize()
    out = np.empty(n + n // 2 + 4096, dtype=np.uint8)
    libs = []
    for spec in args.libs:
        name, rest = spec.split("=", 1)
        path, _, opt = rest.partition(":")
        lib = ctypes.CDLL(os.path.abspath(path), mode=ctypes.RTLD_LOCAL)
        lib.dk_ctx_create.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        lib.dk_dev_block_encode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.POINTER(ctypes.c_size_t)]
        lib.dk_get_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.dk_ctx_destroy.argtypes = [ctypes.c_void_p]
        lib.dk_last_error.restype = ctypes.c_char_p
        lib.dk_last_error.argtypes = [ctypes.c_void_p]
        threads = int(opt.split("=")[1]) if opt.startswith("threads=") else None
        lib.dk_dev_block_decode.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
        libs.append(dict(name=name, lib=lib, threads=threads, wall=[], ent=[], dev=[], stream=None, dwall=[], dent=[]))
    stats = (ctypes.c_double * 1024)()
    for rnd in range(args.rounds):
        for L in libs:
            lib = L["lib"]
            if L["threads"] is not None and hasattr(lib, "dk_set_entropy_threads"):
                lib.dk_set_entropy_threads(L["threads"])
            ctx = ctypes.c_void_p()
            rc = lib.dk_ctx_create(0, n, ctypes.byref(ctx))
            assert rc == 0, (L["name"], rc)
            ln = ctypes.c_size_t(0)
            for step in range(args.steps + 1):  # the first step of a turn is the warm-up
                t0 = time.perf_counter()
                rc = lib.dk_dev_block_encode(ctx, 0, d_in.data_ptr(), n, out.ctypes.data, out.size, ctypes.byref(ln))
                dt = 1e3 * (time.perf_counter() - t0)
                assert rc == 0, (L["name"], rc, lib.dk_last_error(ctx))
                lib.dk_get_stats(ctx, stats)
                if step:
                    L["wall"].append(dt)
                    L["ent"].append(stats[5])
                    L["dev"].append(stats[1] + stats[2] + stats[3] + stats[4])
            s = out[:ln.value].tobytes()
            assert L["stream"] in (None, s), "stream changed between turns"
            L["stream"] = s
            if args.decode:
                d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
                for step in range(3):
                    t0 = time.perf_counter()
                    rc = lib.dk_dev_block_decode(ctx, 0, out.ctypes.data, ln.value, n, d_out.data_ptr())
                    dt = 1e3 * (time.perf_counter() - t0)
                    assert rc == 0, (L["name"], rc, lib.dk_last_error(ctx))
                    lib.dk_get_stats(ctx, stats)
                    if step:
                        L["dwall"].append(dt)
                        L["dent"].append(stats[5])
                assert torch.equal(d_out, d_in), "decode does not give the block back"
                del d_out
            lib.dk_ctx_destroy(ctx)
            print("round %d %-10s wall %s  ms_entropy %s" % (rnd, L["name"], " ".join("%.1f" % x for x in L["wall"][-args.steps:]),
                                                              " ".join("%.1f" % x for x in L["ent"][-args.steps:])), flush=True)
    res = {"workload": args.workload, "n": n, "steps_per_turn": args.steps, "turns": args.rounds, "host_cpus": os.cpu_count(), "libs": {}}
    for L in libs:
        res["libs"][L["name"]] = {"threads_mode": L["threads"], "wall_ms_median": round(float(np.median(L["wall"])), 2),
                                  "ms_entropy_median": round(float(np.median(L["ent"])), 2), "ms_entropy_min": round(float(np.min(L["ent"])), 2),
                                  "ms_device_median": round(float(np.median(L["dev"])), 2),
                                  "stream_bytes": len(L["stream"]), "stream_equals_first": L["stream"] == libs[0]["stream"]}
        if L["dwall"]:
            res["libs"][L["name"]].update({"decode_wall_ms_median": round(float(np.median(L["dwall"])), 2), "decode_ms_entropy_median": round(float(np.median(L["dent"])), 2),
                                           "decode_MBps": round(n / float(np.median(L["dwall"])) / 1e3, 2)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
