#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X BWT compression path (BASELINE.json metric).

A step = one pass of the hot path over one block that is already resident in HBM:
    suffix sort -> BWT -> DC (GPU)  ->  distance stream D2H  ->  `dark` model + range coder (host)  -> coded stream.
`value` is whole-job encode throughput in MB/s (MB = 10^6 input bytes) over all ranks.  Extra keys on the same JSON line:
device-only rates (BWT forward, forward incl. DC), decode rate, the roofline of the dominant kernel measured with HIP
events on the library's own stream, and the CPU oracle timed on a bounded sample (rank 0, N=1 only).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), independent blocks, no data-path collective;
RCCL only gathers the final bitstreams on rank 0 (inside the timed region).  Scaling is weak: one block per GPU.
`python bench.py --gpus N` without a launcher starts the N rank processes itself (the parent never touches the GPU and only
relays the children's output); under torchrun (WORLD_SIZE set) it is one of the ranks.  Every rank gets the same host-CPU share
at every N: container CPU quota / GPUs of the node, so that the N=1 line is the per-rank baseline of the N=8 line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of these boxes supports dmabuf IPC only: without this RCCL's first collective fails with hipIpcGetMemHandle (the image exports it;
# a launcher that scrubs the environment would not)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="enwik8_like_1e8",
                    help="enwik8_like_1e8 (BASELINE configs[1], default) | book1_like_768771 | acgt_2p28 | enwik9_block_125e6 | random_2p30 | "
                         "wordlike_1e8 (Zipf-word text with an 8-bit alphabet: what real text does to the suffix sort, tracked beside the headline) | "
                         "realtext_5e7 (50 MB of the image's own text files -- Python sources, ROCm headers: the only real bytes on the box)")
    ap.add_argument("--n", type=int, default=0, help="override the block size (debug)")
    ap.add_argument("--model", default="dark")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="bytes of the block (a prefix) the CPU oracle is timed on, --cpu-runs times, median reported; 0 (default) = the whole block, "
                         "SURVEY 8(d) 'same inputs': about 12 s of one core per run at 1e8 bytes")
    ap.add_argument("--cpu-runs", type=int, default=3)
    ap.add_argument("--corpus", default="",
                    help="a real file instead of the synthetic stand-in (book1, enwik8, enwik9 ...): rank r codes block r of the file cut into blocks of the "
                         "workload's size (--n overrides); \"data\": \"real\".  Without it, $DARK_CORPUS_DIR/{book1,enwik8,enwik9} is used when the workload "
                         "is that corpus's stand-in and the file exists")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-host-input", action="store_true",
                    help="skip the host_input leg (dk_block_encode from host memory: the H2D copy inside the timed region, beside the headline)")
    ap.add_argument("--pipeline-blocks", type=int, default=15,
                    help="extra leg: this many blocks in flight on one GPU, device stages pipelined against host coding (0 = skip)")
    ap.add_argument("--pipeline-threads", type=int, default=15)
    ap.add_argument("--pipeline-reps", type=int, default=3, help="timed repetitions of the pipelined leg (distinct blocks per repetition)")
    ap.add_argument("--gather", default="rccl", choices=("rccl", "host"),
                    help="how the final bitstreams reach rank 0 inside the timed region: RCCL over xGMI (north_star) or a host-side gloo gather; "
                         "the other one is timed beside it")
    ap.add_argument("--no-rccl-selftest", action="store_true",
                    help="one-rank runs without a launcher: skip the pass of the finished stream through init_process_group('nccl', world_size=1) + "
                         "the RCCL gather code after the timed region")
    ap.add_argument("--node-gpus", type=int, default=0, help="GPUs of the node the per-rank CPU share is computed for (default: visible devices)")
    ap.add_argument("--stub-exchange", action="store_true",
                    help="CPU rehearsal of the multi-rank plumbing (launcher, process group, exchange, max-over-ranks timing) with backend gloo: "
                         "the per-rank streams come from the product's host coder on seeded distances; no GPU is touched")
    return ap.parse_args()


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """--gpus N without a launcher: start N fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment, exactly what torchrun would set) and wait.  This parent has not imported torch or touched HIP and never does."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    if rc:  # a rank failed: make sure none is left waiting in a collective
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


CORPUS_OF = {"book1_like_768771": ("book1", 768771), "enwik8_like_1e8": ("enwik8", 100_000_000), "enwik9_block_125e6": ("enwik9", 125_000_000)}
BOOK1_DARK_FILE_BYTES = 214445  # README.md:20 of the reference: book1 (768 771 B) -> 214 445 B with the `dark` model (4-byte header included)


def find_corpus(args):
    """-> (path, block size) of the real file to use, or (None, 0): --corpus, else $DARK_CORPUS_DIR/<corpus of this workload>"""
    if args.corpus:
        if not os.path.isfile(args.corpus):
            sys.exit("bench.py: --corpus %s: no such file" % args.corpus)
        return args.corpus, args.n or CORPUS_OF.get(args.workload, (None, 0))[1] or os.path.getsize(args.corpus)
    d = os.environ.get("DARK_CORPUS_DIR")
    if d and args.workload in CORPUS_OF and not args.n:
        name, size = CORPUS_OF[args.workload]
        path = os.path.join(d, name)
        if os.path.isfile(path):
            return path, size
    return None, 0


def read_corpus_block(path, block_bytes, index):
    """block `index` (modulo the number of blocks) of the file cut into blocks of block_bytes; a memory-mapped slice, copied once"""
    size = os.path.getsize(path)
    block_bytes = min(block_bytes, size)
    nblocks = max(1, size // block_bytes)  # a short tail is left out: every rank codes a block of the same size
    off = (index % nblocks) * block_bytes
    return np.fromfile(path, dtype=np.uint8, count=block_bytes, offset=off)


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
        "wordlike_1e8": (datagen.word_like, 100_000_000, 5),
        "realtext_5e7": (datagen.real_text, 50_000_000, 0),
    }
    fn, n, seed = gens[workload]
    return fn(n, seed + seed_offset)


def exchange_streams(torch, dist, stream_np, world, rank, dev, group):
    """lengths all_gather + padded gather to rank 0.  dev = a cuda device: default (nccl = RCCL) group, device buffers;
    dev = None: `group` (gloo), host buffers."""
    kw = {} if dev is not None else {"group": group}
    where = dev if dev is not None else "cpu"
    ln = torch.tensor([len(stream_np)], dtype=torch.int64, device=where)
    lens = [torch.zeros(1, dtype=torch.int64, device=where) for _ in range(world)]
    dist.all_gather(lens, ln, **kw)
    lens = [int(x.item()) for x in lens]
    mx = max(lens)
    pad = torch.zeros(mx, dtype=torch.uint8, device=where)
    pad[:len(stream_np)] = torch.from_numpy(stream_np).to(where)
    if rank == 0:
        bufs = [torch.empty(mx, dtype=torch.uint8, device=where) for _ in range(world)]
        dist.gather(pad, bufs, dst=0, **kw)
        return [b[:l] for b, l in zip(bufs, lens)]
    dist.gather(pad, None, dst=0, **kw)
    return None


def rccl_selftest(torch, dist, stream_np, dev):
    """A plain one-GPU run (no launcher) never needs a process group; so that the RCCL leg of the N > 1 runs has executed on the box
    that produced the line, the finished stream goes once more through exactly that code -- init_process_group("nccl") with the world
    size the box has (1) and the device-buffer form of exchange_streams -- after the timed region, and the line says what happened."""
    out = {"world_size": 1, "when": "after the timed region (the headline of a one-rank run has no exchange in it)"}
    try:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        t0 = time.perf_counter()
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        got = exchange_streams(torch, dist, stream_np, 1, 0, dev, None)
        torch.cuda.synchronize(dev)
        out["rccl_init_and_first_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
        t0 = time.perf_counter()
        for _ in range(3):
            got = exchange_streams(torch, dist, stream_np, 1, 0, dev, None)
        torch.cuda.synchronize(dev)
        out["rccl"] = round(1e3 * (time.perf_counter() - t0) / 3, 3)
        out["rccl_ok"] = bool(len(got) == 1 and got[0].cpu().numpy().tobytes() == stream_np.tobytes())
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 -- reported in the line, never fatal for the measurement above
        out["rccl_ok"] = False
        out["error"] = "%s: %s" % (type(e).__name__, e)
    return out


def stub_exchange(args, world, rank):
    """CPU rehearsal of the N-rank plumbing (tests/test_dist_gloo.py): same launcher, same exchange code, backend gloo."""
    import torch
    import torch.distributed as dist
    from dark_amd import model
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def rank_input(r):
        rng = np.random.default_rng(1000 + r)
        m = 20000 + 777 * r
        return (rng.integers(0, 1 << 16, size=m, dtype=np.uint32) >> rng.integers(0, 16, size=m).astype(np.uint32)), rng.integers(0, 256, size=m, dtype=np.uint8)

    d, sym = rank_input(rank)
    # the same planning of the host coding threads as the real run: the smallest number of claimable L3 groups over the ranks decides for all
    from dark_amd import entropy as dk_entropy
    groups = [dk_entropy.l3_groups(4), dk_entropy.l3_groups(2), dk_entropy.l3_groups(5)]
    gt = torch.tensor(groups, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(gt, op=dist.ReduceOp.MIN)
    share = max(1, cpu_budget() // max(world, args.node_gpus, 1))
    if "DK_ENTROPY_THREADS" in os.environ:
        thread_plan = int(os.environ["DK_ENTROPY_THREADS"])
    else:
        thread_plan = dk_entropy.plan_threads(int(gt[0]), int(gt[1]), world, share, int(gt[2]))
        dk_entropy.set_threads(thread_plan)

    def step():
        s = np.frombuffer(model.encode(args.model, d, sym), dtype=np.uint8).copy()  # product host coder (dk_model_encode), no oracle
        return s, (exchange_streams(torch, dist, s, world, rank, None, None) if world > 1 else [torch.from_numpy(s.copy())])

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stream, got = step()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    # one pass of the block-level host coder (dk_stream_encode: the entry point whose thread form the plan sets) over this rank's
    # distances, to see which form it really got; then every rank's report goes to rank 0 like in the real run
    init = np.full(256, 1 << 20, dtype=np.uint32)
    init[:16] = np.arange(16)
    te = time.perf_counter()
    model.stream_encode(args.model, 1 << 20, init, d, sym & 15, 0)
    te = 1e3 * (time.perf_counter() - te)
    threads, group = dk_entropy.last_info()
    mine = rank_report(rank, args.steps, float(time.perf_counter() - t0), {"ms_entropy": te * args.steps}, [threads], [group], groups, len(d), len(stream))
    reports = [None] * world
    if world > 1:
        dist.all_gather_object(reports, mine)
    else:
        reports = [mine]
    if rank == 0:
        ok = len(got) == world
        for r, g in enumerate(got):
            dr, sr = rank_input(r)
            ok = ok and bool((model.decode(args.model, g.numpy().tobytes(), sr) == dr).all())
        print(json.dumps({"metric": "stub_exchange", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * float(el.item()) / args.steps, 3), "gathered_streams_decode": bool(ok),
                          "stream_bytes": [int(len(g)) for g in got], "backend": "gloo", "data": "synthetic",
                          "host_cpu_share_per_rank": share, **thread_summary(reports, thread_plan)}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def rank_report(rank, steps, elapsed, stage_acc, threads_seen, groups_seen, groups_visible, n, stream_bytes, numa=(None, None)):
    """what one rank knows about its own timed steps (gathered on rank 0 and printed under "ranks")"""
    return {"rank": rank, "block_bytes": int(n), "stream_bytes": int(stream_bytes), "ms_per_step": round(1e3 * elapsed / steps, 3),
            "ms_entropy": round(stage_acc.get("ms_entropy", 0.0) / steps, 3),
            "ms_device": round((stage_acc.get("ms_sa", 0.0) + stage_acc.get("ms_bwt", 0.0) + stage_acc.get("ms_dc", 0.0) + stage_acc.get("ms_d2h", 0.0)) / steps, 3),
            "host_entropy_threads": min(threads_seen) if threads_seen else None,
            "host_entropy_threads_max": max(threads_seen) if threads_seen else None,
            "l3_group": groups_seen[-1] if groups_seen else None,
            # memory node of the L3 group the coder claimed, and of the rank's GPU (the coder looks there first: with eight ranks on two sockets a
            # group across the socket link reads its pinned staging remotely); -1 / None: unknown
            "l3_group_numa": numa[0], "gpu_numa": numa[1],
            "l3_groups_with_4_cores": int(groups_visible[0]), "l3_groups_with_2_cores": int(groups_visible[1]),
            "l3_groups_with_5_cores": int(groups_visible[2]) if len(groups_visible) > 2 else 0}


def thread_summary(reports, plan):
    """-> keys for the JSON line: the plan, every rank's report, and which ranks coded on fewer threads than planned"""
    want = plan if plan in (1, 2, 4, 5) else None
    short = [r["rank"] for r in reports if want and r["host_entropy_threads"] is not None and r["host_entropy_threads"] < want]
    return {"entropy_thread_plan": plan, "ranks": reports, "entropy_fallback_ranks": short,
            "entropy_fallback": ("ranks %s coded on fewer host threads than planned (%d): max-over-ranks is theirs" % (short, want)) if short else None}


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
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; pass the same N to both" % (args.gpus, world))
    if args.stub_exchange:
        return stub_exchange(args, world, rank)
    import torch  # device_count() does not initialise the GPU on this image
    # Host threads: the ranks of one node share the container's CPU quota.  Every rank gets quota / (GPUs of the node), at every N, so
    # that an N=1 run uses exactly the per-rank resources of an N=8 run on the same node.  The single-block encoder busy-waits on up
    # to four cores per rank: the thread form is planned below from this share and from the L3 groups every rank can claim.
    node_gpus = max(args.node_gpus or torch.cuda.device_count(), world, 1)
    share = max(1, cpu_budget() // node_gpus)
    args.pipeline_threads = max(1, min(args.pipeline_threads, share - 1))
    import torch.distributed as dist
    import dark_amd

    host_group = None
    # under a launcher (torchrun sets RANK / WORLD_SIZE, also for one rank) the run is a distributed job at every N: process groups,
    # barriers and the RCCL gather are the N > 1 code, executed with whatever world size the launcher gave
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
        host_group = dist.new_group(backend="gloo")  # the host-side alternative for the final gather
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if use_dist else 0)

    # One coding pipeline per rank needs one last-level-cache group per rank.  Every rank reports how many groups it could claim; the
    # smallest answer decides the thread form of ALL ranks, so that max-over-ranks is not decided by who loses the race for a group.
    from dark_amd import entropy as dk_entropy
    groups = [dk_entropy.l3_groups(4), dk_entropy.l3_groups(2), dk_entropy.l3_groups(5)]
    if use_dist:
        gt = torch.tensor(groups, dtype=torch.int64)
        dist.all_reduce(gt, op=dist.ReduceOp.MIN, group=host_group)
        groups_min = [int(x) for x in gt]
    else:
        groups_min = groups
    if "DK_ENTROPY_THREADS" in os.environ:
        thread_plan = int(os.environ["DK_ENTROPY_THREADS"])
    else:
        thread_plan = dk_entropy.plan_threads(groups_min[0], groups_min[1], world, share, groups_min[2])
        dk_entropy.set_threads(thread_plan)

    t_gen = time.time()
    corpus_path, corpus_block = find_corpus(args)
    if corpus_path:
        block = read_corpus_block(corpus_path, corpus_block, rank)
        workload_name = os.path.basename(corpus_path) if len(block) == os.path.getsize(corpus_path) else "%s_block_%d" % (os.path.basename(corpus_path), len(block))
    else:
        block = make_block(args.workload, rank, args.n)
        workload_name = args.workload if not args.n else "wiki_like_%d" % len(block)
    n = len(block)
    t_gen = time.time() - t_gen
    d_in = torch.from_numpy(block).to(dev)
    ctx = dark_amd.Context(n, device=dev.index)
    out_buf = np.empty(n + n // 2 + 4096, dtype=np.uint8)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def gather_streams(stream_np, how=None):
        """The final bitstreams reach rank 0: lengths first (all_gather), then the padded payloads (gather).
        rccl: through device buffers over xGMI (north_star); host: the stream never leaves host memory (gloo)."""
        if not use_dist:
            return [stream_np]
        return exchange_streams(torch, dist, stream_np, world, rank, dev if (how or args.gather) == "rccl" else None, host_group)

    def encode_step():
        s = ctx.dev_block_encode(args.model, d_in, n, out_buf)
        gather_streams(s)
        return s

    for _ in range(args.warmup):
        encode_step()
    ctx.stats_reset()
    ctx.set_profiling(True)
    stage_acc = {}
    threads_seen, groups_seen = [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stream = encode_step()
        st = ctx.stats()
        for k in ("ms_sa", "ms_bwt", "ms_dc", "ms_d2h", "ms_entropy", "ms_total"):
            stage_acc[k] = stage_acc.get(k, 0.0) + st[k]
        threads_seen.append(int(st["entropy_threads"]))
        groups_seen.append(int(st["entropy_l3_group"]))
        numa_seen = (int(st["entropy_l3_numa"]), int(st["gpu_numa"]))
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.set_profiling(False)
    stats = ctx.stats()
    stream = stream.copy()
    # The timed steps carry a HIP event pair around every kernel launch (the per-kernel times the roofline needs); every pair costs the
    # stream about 10 us of idle (measured with rocprofv3: 78 such gaps = 1 ms per suffix sort).  The device-stage times are therefore
    # taken once more from a few steps WITHOUT events -- what a caller of the library sees -- and reported as stage_ms_unprofiled.
    clean_acc = {}
    clean_steps = 3
    for _ in range(clean_steps):
        encode_step()
        st = ctx.stats()
        for k in ("ms_sa", "ms_bwt", "ms_dc", "ms_d2h", "ms_entropy", "ms_total"):
            clean_acc[k] = clean_acc.get(k, 0.0) + st[k] / clean_steps

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed_max = float(tmax.item())

    # host_input leg: where the reference's encode starts (the block in host memory: src/main.rs:87-95 -> src/block/dc.rs:41; SURVEY 8(d)
    # "end-to-end ... including H2D/D2H").  dk_block_encode from pinned and from pageable host memory, the H2D copy inside the timed region,
    # same barrier / max-over-ranks protocol; reported beside the headline, never as `value` (whose input is resident in HBM).
    host_input = None
    if not args.no_host_input:
        hsteps = max(1, min(args.steps, 5))
        host_input = {"steps": hsteps}
        try:
            pinned = torch.from_numpy(block).pin_memory()
            sources = (("pinned", pinned.numpy()), ("pageable", block))
        except RuntimeError as e:  # (no pinned memory to be had: the pageable form alone)
            host_input["pinned_error"] = str(e)
            sources = (("pageable", block),)
        for label, arr in sources:
            ctx.block_encode_into(args.model, arr, out_buf)
            acc = {}
            barrier()
            th = time.perf_counter()
            for _ in range(hsteps):
                s2 = ctx.block_encode_into(args.model, arr, out_buf)
                st = ctx.stats()
                for k in ("ms_h2d", "ms_sa", "ms_bwt", "ms_dc", "ms_d2h", "ms_entropy", "ms_total"):
                    acc[k] = acc.get(k, 0.0) + st[k] / hsteps
            barrier()
            th = torch.tensor([time.perf_counter() - th], dtype=torch.float64, device=dev)
            if use_dist:
                dist.all_reduce(th, op=dist.ReduceOp.MAX)
            host_input[label] = {"MBps": round(world * n * hsteps / float(th.item()) / 1e6, 3), "ms_per_step": round(1e3 * float(th.item()) / hsteps, 3),
                                 "stage_ms": {kk: round(v, 3) for kk, v in acc.items()},
                                 "h2d_GBs": round(n / (acc["ms_h2d"] * 1e-3) / 1e9, 1) if acc["ms_h2d"] > 0 else None,
                                 "stream_identical": bool(s2.tobytes() == stream.tobytes())}
        del sources
    # every rank's view of its own timed steps, for rank 0's line: a rank that fell back to fewer coding threads shows here
    mine = rank_report(rank, args.steps, elapsed, stage_acc, threads_seen, groups_seen, groups, n, len(stream), numa_seen)
    if use_dist:
        reports = [None] * world
        dist.all_gather_object(reports, mine, group=host_group)
    else:
        reports = [mine]

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
            if use_dist:
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            decode_mbps = world * n * args.steps / float(dt.item()) / 1e6
            roundtrip_ok = bool(torch.equal(d_out, d_in))
            dstats = ctx.stats()
        except dark_amd.DarkError as e:  # blocks with byte 0xFF encode but cannot be decoded (reference format)
            roundtrip_ok = "undecodable by reference format: %s" % e
            dstats = None

    # the two ways the final bitstreams can reach rank 0, timed on the stream just produced (the headline used args.gather)
    gather_ms = None
    if use_dist:
        gather_ms = {"world_size": world}
        for how in ("rccl", "host"):
            gather_streams(stream, how)
            barrier()
            tg = time.perf_counter()
            for _ in range(3):
                got = gather_streams(stream, how)
            barrier()
            dtg = torch.tensor([time.perf_counter() - tg], dtype=torch.float64, device=dev)
            dist.all_reduce(dtg, op=dist.ReduceOp.MAX)
            gather_ms[how] = round(1e3 * float(dtg.item()) / 3, 3)
            if rank == 0:  # rank 0's own stream must come back intact through either route
                gather_ms[how + "_ok"] = bool(len(got) == world and got[0].cpu().numpy().tobytes() == stream.tobytes())

    # pipelined leg: B independent blocks per GPU, the device path of block i+1 overlapping the host coding of earlier blocks
    # ("one block per core, pipelined against GPU work of the next block").  Reported beside the headline, never as it.
    # Blocks are distinct: block j of repetition r is the input rotated by a different offset (same statistics, different suffix order);
    # block 0 of every repetition is the input itself, whose stream must equal the single-block path's.
    pipelined = None
    if args.pipeline_blocks > 0:
        B = max(2, min(args.pipeline_blocks, int(2.4e9 // n), 3 * args.pipeline_threads))  # bound pinned staging (about 3 bytes per
        # input byte per block) and, on a small CPU share, the leg's duration (one serial coding pass per block and thread)
        # one serial coding pass per block and worker: a block count that is a multiple of the workers keeps every worker busy to the
        # end (16 blocks on 15 workers would spend half of the time on the sixteenth)
        if B > args.pipeline_threads:
            B -= B % args.pipeline_threads
        cap = len(stream) + len(stream) // 4 + 65536
        outs = [np.empty(cap, dtype=np.uint8) for _ in range(B)]
        try:
            d_blocks = [d_in] + [torch.empty_like(d_in) for _ in range(B - 1)]

            def rotate(rep):
                for j in range(1, B):
                    off = (1 + rep * B + j) * 1_000_003 % n
                    d_blocks[j][:n - off] = d_in[off:]
                    d_blocks[j][n - off:] = d_in[:off]
                torch.cuda.synchronize(dev)

            rotate(0)
            ctx.dev_batch_encode(args.model, d_blocks[:2], [n] * 2, args.pipeline_threads, outs[:2])  # warm the staging slots
            ctx.dev_batch_encode(args.model, d_blocks, [n] * B, args.pipeline_threads, outs)
            secs = []
            same = True
            for rep in range(1, args.pipeline_reps + 1):
                rotate(rep)
                barrier()
                tp = time.perf_counter()
                res = ctx.dev_batch_encode(args.model, d_blocks, [n] * B, args.pipeline_threads, outs)
                barrier()
                dtp = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device=dev)
                if use_dist:
                    dist.all_reduce(dtp, op=dist.ReduceOp.MAX)
                secs.append(float(dtp.item()))
                same = same and res[0].tobytes() == stream.tobytes()
            med = sorted(secs)[len(secs) // 2]
            pipelined = {"blocks_per_gpu": B, "host_threads": args.pipeline_threads, "repetitions": len(secs), "distinct_blocks": True,
                         "MBps": round(world * B * n / med / 1e6, 1), "seconds_median": round(med, 3),
                         "seconds_all": [round(x, 3) for x in secs], "streams_identical_to_single_block_path": bool(same)}
            if not args.no_decode and roundtrip_ok is True:
                d_outs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(B)]
                dsecs = []
                for rep in range(max(1, args.pipeline_reps)):
                    barrier()
                    tp = time.perf_counter()
                    ctx.dev_batch_decode(args.model, res, [n] * B, d_outs, args.pipeline_threads)
                    barrier()
                    dtd = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device=dev)
                    if use_dist:
                        dist.all_reduce(dtd, op=dist.ReduceOp.MAX)
                    dsecs.append(float(dtd.item()))
                pipelined["decode_MBps"] = round(world * B * n / sorted(dsecs)[len(dsecs) // 2] / 1e6, 1)
                pipelined["decode_roundtrip_ok"] = bool(all(torch.equal(o, b_) for o, b_ in zip(d_outs, d_blocks)))
                del d_outs
            del d_blocks
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
        # HBM traffic per launch from the rocprofv3 PMC passes committed under profiles/ (same command, same workload; newest round)
        pmc, pmc_path = None, None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_*.json"))):
            try:
                with open(path) as f:
                    cand = json.load(f)
            except (OSError, ValueError):
                continue
            if not corpus_path and cand.get("workload") == args.workload and (pmc is None or cand.get("round", 0) >= pmc.get("round", 0)):
                pmc, pmc_path = cand, os.path.relpath(path, ROOT)
        # a committed profile describes THIS code only if it saw the same launches: the profiled command's own count of bracketed launches per
        # step and kernel slot (dk_stats, copied into the profile by tools/pmc_traffic.py --bench-line) must agree with this run's, else the
        # profile is stale (taken on an earlier commit) and no traffic figure is derived from it
        my_launches = {slot: round(v["launches"] / k, 3) for slot, v in kern.items()}
        traffic_profile_stale = None
        if pmc and not args.n:
            theirs = pmc.get("scope_launches_per_step")
            mism = {"(profile without launch counts)": []} if theirs is None else {
                slot: [theirs.get(slot, 0.0), my_launches.get(slot, 0.0)] for slot in set(theirs) | set(my_launches)
                if not slot.startswith(("k_dc_", "k_ibwt_")) and abs(theirs.get(slot, 0.0) - my_launches.get(slot, 0.0)) > 0.01}
            traffic_profile_stale = bool(mism)
            if mism:
                print("bench.py: %s is stale -- launches per step [profile, this run] differ: %s" % (pmc_path, mism), file=sys.stderr)
                pmc = None
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
            if pmc and pmc.get("workload") == args.workload and not args.n and (dom_name in pmc["kernels"] or dom_name in pmc.get("slots", {})):
                # (a slot that brackets several kernels -- k_chain, k_rerank_scan -- is in the profile's "slots" table: bytes per step over its launches)
                roofline["traffic"] = (pmc["kernels"][dom_name]["hbm_bytes_per_launch"] if dom_name in pmc["kernels"] else
                                       round(pmc["slots"][dom_name]["hbm_bytes_per_step"] / max(1.0, pmc["slots"][dom_name]["launches_per_step"])))
                roofline["traffic_source"] = pmc_path + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; a committed profile of this command, not this run)"
        fwd_ms = clean_acc["ms_sa"] + clean_acc["ms_bwt"]  # without the event pairs of the profiled steps (see above)
        # BWT-forward roofline the way SURVEY 8(d) defines it: min(B_fwd formula, PMC-measured HBM bytes) / t_fwd against the 8 TB/s peak
        fwd_roofline = None
        if pmc and pmc.get("workload") == args.workload and not args.n:
            P = 2 * -(-max(1, (n - 1).bit_length()) // 8)
            R = stats["rounds"]
            b_formula = n * (85 + R * (44 + 24 * P) + 6)
            fwd_kernels = {kk: v for kk, v in pmc["kernels"].items() if not kk.startswith("k_dc_") and not kk.startswith("k_ibwt_")}
            measured = sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for v in fwd_kernels.values())
            ach = min(b_formula, measured) / (fwd_ms * 1e-3) / 1e9
            # second figure, on USEFUL bytes: the kernels that gather single bytes / words (every gathered 64-byte line carries one
            # wanted item) are capped at their algorithmic bytes (dk_stats.kernel_bytes of this run)
            useful = 0.0
            for kk, v in fwd_kernels.items():
                per_step = v["hbm_bytes_per_launch"] * v["launches_per_step"]
                if kk in ("k_bwt_gather", "k_bwt_gather_list", "k_round_local", "k_lf_finish") and kk in kern:
                    per_step = min(per_step, kern[kk]["bytes"] / k)
                useful += per_step
            ach_u = min(b_formula, useful) / (fwd_ms * 1e-3) / 1e9
            fwd_ms_prof = per["ms_sa"] + per["ms_bwt"]  # the timed steps themselves (with the HIP-event pairs of the per-kernel times)
            fwd_roofline = {"B_fwd_formula_bytes": b_formula, "rounds": R, "passes_P": P, "measured_hbm_bytes": round(measured),
                            "t_fwd_ms": round(fwd_ms, 3), "achieved_GBs": round(ach, 1), "frac_of_8TBs": round(ach / HBM_PEAK_GBS, 4),
                            # the same on the counters as rocprofv3 printed them (no x2 on FETCH_SIZE for streaming kernels: the guide's correction made visible)
                            "frac_raw_counters": None if not all("raw_hbm_bytes_per_launch" in v for v in fwd_kernels.values()) else round(
                                min(b_formula, sum(v["raw_hbm_bytes_per_launch"] * v["launches_per_step"] for v in fwd_kernels.values())) / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "useful_hbm_bytes": round(useful), "achieved_useful_GBs": round(ach_u, 1),
                            "frac_of_8TBs_useful": round(ach_u / HBM_PEAK_GBS, 4),
                            "t_fwd_ms_timed_steps": round(fwd_ms_prof, 3),
                            "frac_of_8TBs_timed_steps": round(min(b_formula, measured) / (fwd_ms_prof * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "time_source": "t_fwd_ms: three extra steps of this run without event pairs; t_fwd_ms_timed_steps: the timed steps",
                            "traffic_source": pmc_path + " (HBM bytes: a committed rocprofv3 --pmc profile of this command, not this run)"}
        result = {
            "metric": "bwt_encode_MBps", "value": round(world * n * k / elapsed_max / 1e6, 3), "unit": "MB/s",
            "n_gpus": world, "steps": k, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed_max / k, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u64 integer", "data": "real" if corpus_path else ("real (text files of this image, datagen.real_text)" if args.workload == "realtext_5e7" and not args.n else "synthetic"),
            "config": {"workload": workload_name, "block_bytes": n, "model": args.model,
                       "blocks_per_gpu": 1, "parallelism": "block-per-gpu x%d" % world, "host_cpus_per_rank": share, "node_gpus": node_gpus,
                       "gather": args.gather if use_dist else "none"},
            "bwt_forward_MBps_per_gpu": round(n / (fwd_ms * 1e-3) / 1e6, 1),
            "device_forward_MBps_per_gpu": round(n / ((fwd_ms + clean_acc["ms_dc"]) * 1e-3) / 1e6, 1),
            "decode_MBps": None if decode_mbps is None else round(decode_mbps, 3),
            "roundtrip_ok": roundtrip_ok,
            "compressed_bytes": int(len(stream)), "ratio": round(len(stream) / n, 4),
            "value_host_input_MBps": None if not host_input or "pinned" not in host_input else host_input["pinned"]["MBps"],
            "host_input": host_input,
            # (ms_h2d: the H2D copy of the host_input leg's pinned form -- the timed steps of `value` start with the block in HBM and have none)
            "stage_ms": {**{kk: round(v, 3) for kk, v in per.items()},
                         "ms_h2d": None if not host_input or "pinned" not in host_input else host_input["pinned"]["stage_ms"]["ms_h2d"]},
            "stage_ms_unprofiled": {kk: round(v, 3) for kk, v in clean_acc.items()},
            "sa_rounds": stats["rounds"], "sort_passes": stats["sort_passes"], "dc_runs": stats["dc_runs"],
            "host_entropy_threads": stats["entropy_threads"], "host_cpu_share_per_rank": share,
            **thread_summary(reports, thread_plan),
            "kernel_ms_per_step": {kk: round(v["ms"] / k, 3) for kk, v in sorted(kern.items(), key=lambda x: -x[1]["ms"])},
            "kernel_launches_per_step": my_launches,
            "roofline": roofline,
            "traffic_profile_stale": traffic_profile_stale,  # true: the newest committed PMC profile of this workload saw other launch counts -> traffic: null
            "bwt_forward_roofline": fwd_roofline,
            "pipelined": pipelined,
            "gather_ms": gather_ms,
            "datagen_s": round(t_gen, 2),
        }
        if corpus_path:
            result["corpus"] = {"path": corpus_path, "file_bytes": os.path.getsize(corpus_path), "block_bytes": n}
            if os.path.basename(corpus_path) == "book1" and n == 768771 and args.model == "dark":
                # the one number the reference publishes for this path (README.md:20): the .dark file = 4-byte header + stream
                result["corpus"].update({"dark_file_bytes": int(len(stream)) + 4, "reference_dark_file_bytes": BOOK1_DARK_FILE_BYTES,
                                         "matches_reference_size": bool(len(stream) + 4 == BOOK1_DARK_FILE_BYTES)})
        if dstats:
            result["decode_stage_ms"] = {kk: round(dstats[kk], 3) for kk in ("ms_entropy", "ms_h2d", "ms_ibwt", "ms_total")}  # last step
            # where the host half of the decode goes, nanoseconds per distance, each part timed alone on this block (rank 0): the range
            # decoder + model fed the symbol sequence (dk_model_decode), and the rebuild of L from the distances (dk_dc_decode)
            try:
                from dark_amd import model as _model
                d_bwt = torch.empty(n, dtype=torch.uint8, device=dev)
                ctx.dev_bwt_forward(d_in, n, d_bwt)
                d_dist = torch.empty(n, dtype=torch.int32, device=dev)
                d_sym = torch.empty(n, dtype=torch.uint8, device=dev)
                init, m_runs = ctx.dev_dc_encode(d_bwt, n, d_dist, d_sym)
                dist_h = d_dist[:m_runs].cpu().numpy().view(np.uint32)
                sym_h = d_sym[:m_runs].cpu().numpy()
                del d_bwt, d_dist, d_sym
                coded = _model.encode(args.model, dist_h, sym_h)
                t_a = time.perf_counter()
                back = _model.decode(args.model, coded, sym_h)
                t_a = time.perf_counter() - t_a
                t_b = time.perf_counter()
                ctx.dc_decode(init, dist_h, n)
                t_b = time.perf_counter() - t_b
                result["decode_ns_per_distance"] = {"distances": int(m_runs),
                                                    "whole_host_stage": round(1e6 * result["decode_stage_ms"]["ms_entropy"] / m_runs, 2),
                                                    "range_decoder_and_model_alone": round(1e9 * t_a / m_runs, 2),
                                                    "dc_rebuild_alone": round(1e9 * t_b / m_runs, 2),
                                                    "model_roundtrip_ok": bool((back == dist_h).all())}
            except dark_amd.DarkError as e:
                result["decode_ns_per_distance"] = {"error": str(e)}
            result["decode_kernel_ms_per_step"] = {kk: round(v["ms"] / k, 3) for kk, v in sorted(dstats["kernels"].items(), key=lambda x: -x[1]["ms"])}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import orc  # the checker, timed as the CPU baseline (never the thing measured above)
            sample = block[:min(n, args.cpu_sample)] if args.cpu_sample else block
            runs, ref_stream, stages = [], None, None
            for _ in range(max(1, args.cpu_runs)):
                tc = time.perf_counter()
                ref_stream = orc.block_dc_encode(args.model, sample)
                runs.append(time.perf_counter() - tc)
                if runs[-1] == sorted(runs)[len(runs) // 2]:
                    stages = orc.last_stage_seconds()
            tc = sorted(runs)[len(runs) // 2]
            result["cpu_baseline"] = {"value": round(len(sample) / tc / 1e6, 3), "unit": "MB/s", "cores": 1, "kind": "port",
                                      "sample": "%s of the same block (%d bytes) as one block, median of %d runs; C restatement of the reference CPU "
                                                "path (SA-IS + BWT + DC + dark model/range coder)"
                                                % ("the whole" if len(sample) == n else "a prefix", len(sample), len(runs)),
                                      "seconds": round(tc, 2), "seconds_all": [round(x, 2) for x in runs],
                                      "stage_s": {kk: round(v, 3) for kk, v in (stages or {}).items()}}
            # parity at sample scale: the same prefix through the GPU path must give the identical coded stream
            gpu_sample = stream.tobytes() if len(sample) == n else ctx.dev_block_encode(args.model, d_in[:len(sample)], len(sample)).tobytes()
            result["cpu_baseline"]["gpu_stream_identical_on_sample"] = bool(gpu_sample == ref_stream)
        if not use_dist and not args.no_rccl_selftest:
            result["gather_ms"] = rccl_selftest(torch, dist, stream, dev)
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
