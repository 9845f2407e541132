#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X BWT compression path (BASELINE.json metric).

A step = one pass of the hot path over one block that is already resident in HBM:
    suffix sort -> BWT -> DC (GPU)  ->  distance stream D2H  ->  `dark` model + range coder (host)  -> coded stream.
`value` is whole-job encode throughput in MB/s (MB = 10^6 input bytes) over all ranks.  Extra keys on the same JSON line:
device-only rates (BWT forward, forward incl. DC), decode rate, the roofline of the dominant kernel measured with HIP
events on the library's own stream, and the CPU oracle timed on a bounded sample (rank 0, N=1 only).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), independent blocks, no data-path collective;
RCCL only gathers the final bitstreams on rank 0 (inside the timed region).  Scaling is weak: one block per GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="enwik8_like_1e8",
                    help="enwik8_like_1e8 (BASELINE configs[1], default) | book1_like_768771 | acgt_2p28 | enwik9_block_125e6 | random_2p30")
    ap.add_argument("--n", type=int, default=0, help="override the block size (debug)")
    ap.add_argument("--model", default="dark")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=125_000_000,
                    help="bytes of the block the CPU oracle is timed on (default: the whole block up to 125 MB, about 10 s of one core)")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--pipeline-blocks", type=int, default=15,
                    help="extra leg: this many blocks in flight on one GPU, device stages pipelined against host coding (0 = skip)")
    ap.add_argument("--pipeline-threads", type=int, default=15)
    return ap.parse_args()


def make_block(workload, seed_offset, n_override):
    from dark_amd import datagen
    if n_override:
        return datagen.wiki_like(n_override, 2 + seed_offset)
    gens = {
        "book1_like_768771": (datagen.english_like, 768771, 1),
        "enwik8_like_1e8": (datagen.wiki_like, 100_000_000, 2),
        "acgt_2p28": (datagen.acgt, 1 << 28, 3),
        "enwik9_block_125e6": (datagen.wiki_like, 125_000_000, 40),
        "random_2p30": (datagen.random_bytes, 1 << 30, 50),
    }
    fn, n, seed = gens[workload]
    return fn(n, seed + seed_offset)


def cpu_budget():
    """CPUs this container may use (cgroup quota; os.cpu_count() when unlimited)."""
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(q) // int(period))
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


def main():
    args = parse()
    # Host threads: the ranks of one node share the container's CPU quota.  The single-block encoder busy-waits on up to four cores
    # per rank; tell the library to stay within this rank's share (it reads DK_ENTROPY_THREADS once, when it is loaded).
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    share = max(1, cpu_budget() // max(1, local_world))
    if "DK_ENTROPY_THREADS" not in os.environ and share < 4:
        os.environ["DK_ENTROPY_THREADS"] = "2" if share >= 2 else "1"
    args.pipeline_threads = max(1, min(args.pipeline_threads, share - 1))
    import torch
    import torch.distributed as dist
    import dark_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    t_gen = time.time()
    block = make_block(args.workload, rank, args.n)
    n = len(block)
    t_gen = time.time() - t_gen
    d_in = torch.from_numpy(block).to(dev)
    ctx = dark_amd.Context(n, device=dev.index)
    out_buf = np.empty(n + n // 2 + 4096, dtype=np.uint8)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def gather_streams(stream_np):
        """RCCL gathers the final bitstreams on rank 0: lengths first, then the padded payloads."""
        if world == 1:
            return [stream_np]
        ln = torch.tensor([len(stream_np)], dtype=torch.int64, device=dev)
        lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(lens, ln)
        mx = int(max(int(x.item()) for x in lens))
        pad = torch.zeros(mx, dtype=torch.uint8, device=dev)
        pad[:len(stream_np)] = torch.from_numpy(stream_np).to(dev)
        if rank == 0:
            bufs = [torch.empty(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
            dist.gather(pad, bufs, dst=0)
            return [b[:int(l.item())] for b, l in zip(bufs, lens)]
        dist.gather(pad, None, dst=0)
        return None

    def encode_step():
        s = ctx.dev_block_encode(args.model, d_in, n, out_buf)
        gather_streams(s)
        return s

    for _ in range(args.warmup):
        encode_step()
    ctx.stats_reset()
    ctx.set_profiling(True)
    stage_acc = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stream = encode_step()
        st = ctx.stats()
        for k in ("ms_sa", "ms_bwt", "ms_dc", "ms_d2h", "ms_entropy", "ms_total"):
            stage_acc[k] = stage_acc.get(k, 0.0) + st[k]
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.set_profiling(False)
    stats = ctx.stats()
    stream = stream.copy()

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed_max = float(tmax.item())

    # decode leg (same protocol, reported beside the headline)
    decode_mbps = None
    roundtrip_ok = None
    dstats = None
    if not args.no_decode:
        d_out = torch.empty(n, dtype=torch.uint8, device=dev)
        try:
            ctx.dev_block_decode(args.model, stream, n, d_out)
            ctx.stats_reset()
            ctx.set_profiling(True)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                ctx.dev_block_decode(args.model, stream, n, d_out)
            barrier()
            ctx.set_profiling(False)
            dt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            decode_mbps = world * n * args.steps / float(dt.item()) / 1e6
            roundtrip_ok = bool(torch.equal(d_out, d_in))
            dstats = ctx.stats()
        except dark_amd.DarkError as e:  # blocks with byte 0xFF encode but cannot be decoded (reference format)
            roundtrip_ok = "undecodable by reference format: %s" % e
            dstats = None

    # pipelined leg: B independent blocks per GPU, the device path of block i+1 overlapping the host coding of earlier blocks
    # ("one block per core, pipelined against GPU work of the next block").  Reported beside the headline, never as it.
    pipelined = None
    if args.pipeline_blocks > 0:
        B = max(2, min(args.pipeline_blocks, int(2.4e9 // n)))  # bound pinned staging (about 3 bytes per input byte per block)
        # one serial coding pass per block and worker: a block count that is a multiple of the workers keeps every worker busy to the
        # end (16 blocks on 15 workers would spend half of the time on the sixteenth)
        if B > args.pipeline_threads:
            B -= B % args.pipeline_threads
        outs = [np.empty(len(stream) + len(stream) // 8 + 65536, dtype=np.uint8) for _ in range(B)]
        try:
            ctx.dev_batch_encode(args.model, [d_in] * 2, [n] * 2, args.pipeline_threads, outs[:2])  # warm the staging slots
            warm = ctx.dev_batch_encode(args.model, [d_in] * B, [n] * B, args.pipeline_threads, outs)
            barrier()
            tp = time.perf_counter()
            res = ctx.dev_batch_encode(args.model, [d_in] * B, [n] * B, args.pipeline_threads, outs)
            barrier()
            dtp = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(dtp, op=dist.ReduceOp.MAX)
            same = all(r.tobytes() == stream.tobytes() for r in (res[0], res[-1]))
            pipelined = {"blocks_per_gpu": B, "host_threads": args.pipeline_threads, "MBps": round(world * B * n / float(dtp.item()) / 1e6, 1),
                         "seconds": round(float(dtp.item()), 3), "streams_identical_to_single_block_path": bool(same)}
            del warm
            if not args.no_decode and roundtrip_ok is True:
                d_outs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(B)]
                barrier()
                tp = time.perf_counter()
                ctx.dev_batch_decode(args.model, res, [n] * B, d_outs, args.pipeline_threads)
                barrier()
                dtd = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device=dev)
                if world > 1:
                    dist.all_reduce(dtd, op=dist.ReduceOp.MAX)
                pipelined["decode_MBps"] = round(world * B * n / float(dtd.item()) / 1e6, 1)
                pipelined["decode_roundtrip_ok"] = bool(all(torch.equal(o, d_in) for o in (d_outs[0], d_outs[-1])))
                del d_outs
        except dark_amd.DarkError as e:
            pipelined = {"error": str(e)}
        del outs

    # what the box's HBM delivers to a plain device-to-device copy (report the measured peak beside the 8 TB/s figure)
    copy_gbs = None
    if rank == 0:
        try:
            a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            b = torch.empty_like(a)
            b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize(dev)
            copy_gbs = 10 * 2 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del a, b
        except RuntimeError:
            copy_gbs = None

    if rank == 0:
        k = args.steps
        per = {kk: v / k for kk, v in stage_acc.items()}
        kern = stats["kernels"]
        dom_name = max(kern, key=lambda kk: kern[kk]["ms"]) if kern else None
        roofline = None
        pmc = None
        try:  # HBM traffic per launch from the rocprofv3 PMC passes committed under profiles/ (same command, same workload)
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_enwik8like_1e8.json")) as f:
                pmc = json.load(f)
        except OSError:
            pass
        if dom_name:
            dk_ = kern[dom_name]
            achieved = dk_["bytes"] / (dk_["ms"] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "launches": dk_["launches"], "avg_launch_us": round(1e3 * dk_["ms"] / dk_["launches"], 2),
                        "traffic_source": None,
                        "algorithmic_bytes_per_launch": round(dk_["bytes"] / dk_["launches"], 1),
                        "measured_copy_GBs": None if copy_gbs is None else round(copy_gbs, 1),
                        "frac_of_measured_copy": None if not copy_gbs else round(achieved / copy_gbs, 4)}
            if pmc and pmc.get("workload") == args.workload and not args.n and dom_name in pmc["kernels"]:
                roofline["traffic"] = pmc["kernels"][dom_name]["hbm_bytes_per_launch"]
                roofline["traffic_source"] = "profiles/r01_pmc_traffic_enwik8like_1e8.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        fwd_ms = per["ms_sa"] + per["ms_bwt"]
        # BWT-forward roofline the way SURVEY 8(d) defines it: min(B_fwd formula, PMC-measured HBM bytes) / t_fwd against the 8 TB/s peak
        fwd_roofline = None
        if pmc and pmc.get("workload") == args.workload and not args.n:
            P = 2 * -(-max(1, (n - 1).bit_length()) // 8)
            R = stats["rounds"]
            b_formula = n * (85 + R * (44 + 24 * P) + 6)
            measured = sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for kk, v in pmc["kernels"].items() if not kk.startswith("k_dc_"))
            ach = min(b_formula, measured) / (fwd_ms * 1e-3) / 1e9
            fwd_roofline = {"B_fwd_formula_bytes": b_formula, "rounds": R, "passes_P": P, "measured_hbm_bytes": round(measured),
                            "t_fwd_ms": round(fwd_ms, 3), "achieved_GBs": round(ach, 1), "frac_of_8TBs": round(ach / HBM_PEAK_GBS, 4)}
        result = {
            "metric": "bwt_encode_MBps", "value": round(world * n * k / elapsed_max / 1e6, 3), "unit": "MB/s",
            "n_gpus": world, "steps": k, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed_max / k, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u64 integer", "data": "synthetic",
            "config": {"workload": args.workload if not args.n else "wiki_like_%d" % n, "block_bytes": n, "model": args.model,
                       "blocks_per_gpu": 1, "parallelism": "block-per-gpu x%d" % world},
            "bwt_forward_MBps_per_gpu": round(n / (fwd_ms * 1e-3) / 1e6, 1),
            "device_forward_MBps_per_gpu": round(n / ((fwd_ms + per["ms_dc"]) * 1e-3) / 1e6, 1),
            "decode_MBps": None if decode_mbps is None else round(decode_mbps, 3),
            "roundtrip_ok": roundtrip_ok,
            "compressed_bytes": int(len(stream)), "ratio": round(len(stream) / n, 4),
            "stage_ms": {kk: round(v, 3) for kk, v in per.items()},
            "sa_rounds": stats["rounds"], "sort_passes": stats["sort_passes"], "dc_runs": stats["dc_runs"],
            "host_entropy_threads": stats["entropy_threads"], "host_cpu_share_per_rank": share,
            "kernel_ms_per_step": {kk: round(v["ms"] / k, 3) for kk, v in sorted(kern.items(), key=lambda x: -x[1]["ms"])},
            "roofline": roofline,
            "bwt_forward_roofline": fwd_roofline,
            "pipelined": pipelined,
            "datagen_s": round(t_gen, 2),
        }
        if dstats:
            result["decode_stage_ms"] = {kk: round(dstats[kk], 3) for kk in ("ms_entropy", "ms_h2d", "ms_ibwt", "ms_total")}
            result["decode_kernel_ms_per_step"] = {kk: round(v["ms"] / k, 3) for kk, v in sorted(dstats["kernels"].items(), key=lambda x: -x[1]["ms"])}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import orc  # the checker, timed as the CPU baseline (never the thing measured above)
            sample = block[:min(n, args.cpu_sample)]
            tc = time.perf_counter()
            ref_stream = orc.block_dc_encode(args.model, sample)
            tc = time.perf_counter() - tc
            stages = orc.last_stage_seconds()
            result["cpu_baseline"] = {"value": round(len(sample) / tc / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
                                      "sample": "%s of the same block (%d bytes), one block, C restatement of the reference CPU path "
                                                "(SA-IS + BWT + DC + dark model/range coder)"
                                                % ("the whole" if len(sample) == n else "a prefix", len(sample)),
                                      "seconds": round(tc, 2), "stage_s": {kk: round(v, 3) for kk, v in stages.items()}}
            # parity at sample scale: the same prefix through the GPU path must give the identical coded stream
            gpu_sample = stream.tobytes() if len(sample) == n else ctx.dev_block_encode(args.model, d_in[:len(sample)], len(sample)).tobytes()
            result["cpu_baseline"]["gpu_stream_identical_on_sample"] = bool(gpu_sample == ref_stream)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
