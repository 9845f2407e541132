"""N>1 path on CPU: world_size-2 gloo run of bench.py's exchange step (lengths all_gather + padded gather of the final
bitstreams to rank 0).  The per-rank streams come from the product's host entropy stage fed by oracle-made DC streams, so
no GPU is needed; the collective pattern is the one bench.py uses with backend nccl (= RCCL) on the GPU box."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from dark_amd import datagen, model
    from oracle import orc
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    # block b -> rank b (independent blocks, no data-path collective)
    block = datagen.wiki_like(60000, seed=40 + rank)
    bwt, origin = orc.bwt_forward(block)
    dc = orc.dc_encode(bwt)
    stream = np.frombuffer(model.stream_encode("dark", len(block), dc["init"], dc["d"], dc["sym"], origin), dtype=np.uint8)
    # exchange step, as in bench.py gather_streams()
    ln = torch.tensor([len(stream)], dtype=torch.int64)
    lens = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(lens, ln)
    mx = int(max(int(x.item()) for x in lens))
    pad = torch.zeros(mx, dtype=torch.uint8)
    pad[:len(stream)] = torch.from_numpy(stream.copy())
    if rank == 0:
        bufs = [torch.empty(mx, dtype=torch.uint8) for _ in range(world)]
        dist.gather(pad, bufs, dst=0)
        got = [b[:int(l.item())].numpy().tobytes() for b, l in zip(bufs, lens)]
        # rank 0 checks every gathered stream decodes to the block its owner compressed
        ok = True
        for r, s in enumerate(got):
            blk = datagen.wiki_like(60000, seed=40 + r)
            ok = ok and orc.block_dc_decode("dark", s, len(blk)) == blk.tobytes()
            ok = ok and s == orc.block_dc_encode("dark", blk)
        q.put(("rank0", ok, [len(s) for s in got]))
    else:
        dist.gather(pad, None, dst=0)
        q.put(("rank%d" % rank, True, [len(stream)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_block_parallel_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    by = {r[0]: r for r in results}
    assert by["rank0"][1] is True
    assert len(by["rank0"][2]) == 2 and by["rank0"][2][1] == by["rank1"][2][0]


def _run_bench(cmd, timeout=280):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_bench_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with no launcher must run TWO ranks (it used to run one and report n_gpus 1).  CPU stub of
    the exchange: same launcher, same exchange_streams() code path, backend gloo, streams from the product's host coder."""
    res = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-exchange"])
    assert res["n_gpus"] == 2 and res["gathered_streams_decode"] is True and len(res["stream_bytes"]) == 2
    _check_rank_reports(res, 2)


@pytest.mark.timeout(600)
def test_eight_ranks_stub_exchange():
    """the 8-rank shape of the node the driver measures on -- launcher, process group, the thread plan over eight reports, the
    padded gather of eight streams of eight different lengths -- rehearsed with backend gloo (no GPU; eight processes on this host's cores)."""
    res = _run_bench([sys.executable, "bench.py", "--gpus", "8", "--steps", "1", "--warmup", "0", "--stub-exchange", "--node-gpus", "8"], timeout=560)
    assert res["n_gpus"] == 8 and res["gathered_streams_decode"] is True
    assert len(res["stream_bytes"]) == 8 and len(set(res["stream_bytes"])) == 8  # eight lengths, every one padded to the longest on the way
    _check_rank_reports(res, 8)
    assert all("l3_group_numa" in r and "gpu_numa" in r for r in res["ranks"])


def test_l3_groups_are_claimed_on_the_gpus_memory_node_first():
    """two sockets with eight L3 groups each (2 x EPYC 9575F: what an 8 x MI355X node looks like), four GPUs per socket.  A rank
    whose GPU hangs on node 1 tries the groups of node 1 first -- nearest to where it runs -- and only then the other socket's; without a
    preference (or on a one-node host) the order is by distance alone, as before."""
    from dark_amd import entropy
    numa = [0] * 8 + [1] * 8
    assert entropy.claim_order(numa, 2, 1) == [8, 9, 10, 11, 12, 13, 14, 15, 2, 3, 1, 4, 0, 5, 6, 7]
    assert entropy.claim_order(numa, 10, 1) == [10, 11, 9, 12, 8, 13, 14, 15, 7, 6, 5, 4, 3, 2, 1, 0]
    assert entropy.claim_order(numa, 10, 0) == [7, 6, 5, 4, 3, 2, 1, 0, 10, 11, 9, 12, 8, 13, 14, 15]
    assert entropy.claim_order(numa, 2, -1) == [2, 3, 1, 4, 0, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]
    assert entropy.claim_order([-1] * 4, 1, 0) == [1, 2, 0, 3]          # no node information: by distance
    assert entropy.claim_order([0, 0, 0], 0, 5) == [0, 1, 2]             # a node without any group: by distance
    with pytest.raises(Exception):
        entropy.claim_order(numa, 16, 0)


def _check_rank_reports(res, world):
    """the line carries EVERY rank's host coding form (threads, claimed L3 group, ms_entropy), the form planned for
    all ranks together, and says so when a rank coded on fewer threads than planned."""
    from dark_amd import entropy
    plan = res["entropy_thread_plan"]
    assert plan in (1, 2, 4, 5) and [r["rank"] for r in res["ranks"]] == list(range(world))
    g4 = min(r["l3_groups_with_4_cores"] for r in res["ranks"])
    g2 = min(r["l3_groups_with_2_cores"] for r in res["ranks"])
    g5 = min(r["l3_groups_with_5_cores"] for r in res["ranks"])
    # deterministic: when there are fewer claimable L3 groups than ranks, ALL ranks drop to the narrower form together
    assert plan == entropy.plan_threads(g4, g2, world, res["host_cpu_share_per_rank"], g5)
    for r in res["ranks"]:
        assert {"ms_entropy", "ms_per_step", "host_entropy_threads", "l3_group", "stream_bytes"} <= set(r)
        assert 1 <= r["host_entropy_threads"] <= plan
        assert (r["l3_group"] >= 0) == (r["host_entropy_threads"] > 1)
    short = [r["rank"] for r in res["ranks"] if r["host_entropy_threads"] < plan]
    assert res["entropy_fallback_ranks"] == short and (res["entropy_fallback"] is None) == (not short)


def test_thread_plan_is_deterministic():
    from dark_amd import entropy
    assert entropy.plan_threads(16, 16, 8, 8) == 4      # a group of four cores for every rank
    assert entropy.plan_threads(16, 16, 8, 8, 16) == 5 and entropy.plan_threads(16, 16, 8, 8, 7) == 4 and entropy.plan_threads(16, 16, 8, 4, 16) == 4
    assert entropy.plan_threads(7, 16, 8, 8) == 2       # one rank would lose the race for a 4-core group: everybody takes two threads
    assert entropy.plan_threads(7, 7, 8, 8) == 1
    assert entropy.plan_threads(16, 16, 8, 3) == 2 and entropy.plan_threads(16, 16, 8, 1) == 1  # the CPU share caps the form
    assert entropy.l3_groups(1) >= entropy.l3_groups(2) >= entropy.l3_groups(4) >= entropy.l3_groups(5) >= 0
    with pytest.raises(Exception):
        entropy.set_threads(3)
    entropy.set_threads(0)


@pytest.mark.timeout(300)
def test_bench_under_torchrun():
    """the driver's documented N>1 command line: torch.distributed.run starts the ranks, bench.py must not spawn again"""
    res = _run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                      "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-exchange"])
    assert res["n_gpus"] == 2 and res["gathered_streams_decode"] is True
    _check_rank_reports(res, 2)


def test_bench_rejects_mismatched_world():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "3", "--stub-exchange"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
