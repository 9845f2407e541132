"""Times one device stage alone on the GPU box, per kernel slot, with and without HIP-event pairs around the launches:

    python tools/stage_time.py sa  [N] [text|wordlike|acgt|random]     suffix sort + BWT (dk_dev_bwt_forward)
    python tools/stage_time.py dc  [N] [text|acgt|random]     distance coding (dk_dev_dc_encode); "text" = the BWT of the text block
    python tools/stage_time.py ibwt [N] [text|acgt|random]    inverse BWT (dk_dev_bwt_inverse) of the block's BWT

REPS=k repeats each measurement k times (default 4).  DARK_AMD_LIB=path/to/other/libdark_amd.so times another build of the library
(A/B runs inside one gpurun call; box-to-box variation is a few per cent).  Under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ...`
this is the small program to put after `--` when a kernel's instruction counts are wanted."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dark_amd  # noqa: E402
from dark_amd import datagen  # noqa: E402


def make_input(kind, n):
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    if kind == "text":
        return torch.from_numpy(datagen.wiki_like(n, 2)).cuda()
    if kind == "wordlike":  # the bench workload wordlike_1e8 at n = 1e8
        return torch.from_numpy(datagen.word_like(n, 5)).cuda()
    if kind == "wordlike26":  # the 28-symbol variant (5-bit codes: eleven symbols in the initial key)
        return torch.from_numpy(datagen.word_like(n, 6, alpha=26)).cuda()
    if kind == "wordlike_big_vocab":  # 2 M words, Zipf 1.15: fewer repetitions of the frequent words' neighbourhoods
        return torch.from_numpy(datagen.word_like(n, 7, vocab=2_000_000)).cuda()
    if kind == "mixed":  # half Markov text, half word-like
        import numpy as _np
        return torch.from_numpy(_np.concatenate([datagen.wiki_like(n // 2, 2), datagen.word_like(n - n // 2, 5)])).cuda()
    if kind == "acgt":
        return torch.from_numpy(np.frombuffer(b"ACGT", np.uint8)).cuda()[torch.randint(0, 4, (n,), device="cuda", generator=g)]
    return torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda", generator=g)


def main():
    stage = sys.argv[1] if len(sys.argv) > 1 else "sa"
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
    kind = sys.argv[3] if len(sys.argv) > 3 else "text"
    reps = int(os.environ.get("REPS", "4"))
    lib = os.path.basename(os.environ.get("DARK_AMD_LIB", "libdark_amd.so"))
    t = make_input(kind, n)
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    with dark_amd.Context(n) as ctx:
        if stage == "dc":
            if kind == "text":  # the stage's real input on this workload
                ctx.dev_bwt_forward(t, n, out)
                t, out = out, t
            dist = torch.empty(n, dtype=torch.int32, device="cuda")
            rank = torch.empty(n, dtype=torch.uint8, device="cuda")
        if stage == "ibwt":
            origin = ctx.dev_bwt_forward(t, n, out)
            back = torch.empty(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        plain, kernels, passes = [], {}, 0
        for prof in (True, False):
            ctx.set_profiling(prof)
            for rep in range(reps):
                ctx.stats_reset()
                t0 = time.perf_counter()
                if stage == "dc":
                    ctx.dev_dc_encode(t, n, dist, out, rank)
                elif stage == "ibwt":
                    ctx.dev_bwt_inverse(out, n, origin, back)
                else:
                    ctx.dev_bwt_forward(t, n, out)
                wall = (time.perf_counter() - t0) * 1e3
                st = ctx.stats()
                kern = {k: round(v["ms"], 3) for k, v in st["kernels"].items()} if prof and rep == reps - 1 else ""
                if kern:
                    kernels = kern
                if not prof:
                    plain.append(st["ms_dc"] if stage == "dc" else st["ms_ibwt"] if stage == "ibwt" else st["ms_sa"] + st["ms_bwt"])
                passes = st["sort_passes"]
                print(lib, stage, kind, n, "events" if prof else "plain", "wall %.2f ms  stage %.2f ms  rounds %d" %
                      (wall, st["ms_dc"] if stage == "dc" else st["ms_ibwt"] if stage == "ibwt" else st["ms_sa"] + st["ms_bwt"], st["rounds"]), kern, flush=True)
        print("SUMMARY", lib, stage, kind, n, "median plain %.3f ms" % sorted(plain)[len(plain) // 2], "passes", passes, "rounds", st["rounds"],
              "| per kernel (one profiled step):", " ".join("%s=%.3f" % (k[2:], v) for k, v in sorted(kernels.items(), key=lambda x: -x[1])))
        if stage == "ibwt":
            assert torch.equal(back, t), "inverse BWT does not give the block back"
        print("checksum", int(out[: 1 << 22].to(torch.int64).sum()))


if __name__ == "__main__":
    main()
