"""The RCCL (backend "nccl") leg of bench.py on the world size a one-GPU box has (it had never executed anywhere).

SURVEY section 8(e): independent blocks, one rank per GPU, RCCL only gathers the final bitstreams on rank 0.  Both ways a run can reach
that code at N = 1 are driven here as child processes (the rank initialises its GPU itself, nothing of this test process is inherited):
under the driver's launcher command line (torch.distributed.run, one rank: the N > 1 code with WORLD_SIZE=1, exchange inside the timed
region) and as a plain `python bench.py` (the finished stream goes through init_process_group("nccl", world_size=1) + the device-buffer
exchange once more after the timed region)."""
import json
import os
import subprocess
import sys

import pytest

from test_dist_gloo import ROOT, _free_port

pytestmark = pytest.mark.gpu
SMALL = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "book1_like_768771", "--no-cpu-baseline", "--pipeline-blocks", "0"]


def _run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bench_rank_body_under_torchrun_one_rank():
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py"] + SMALL)
    g = res["gather_ms"]
    assert res["n_gpus"] == 1 and res["config"]["gather"] == "rccl" and res["roundtrip_ok"] is True
    assert g["world_size"] == 1 and g["rccl_ok"] is True and g["host_ok"] is True and g["rccl"] > 0 and g["host"] > 0
    assert len(res["ranks"]) == 1 and res["ranks"][0]["rank"] == 0
    print("torchrun x1:", g)


@pytest.mark.timeout(600)
def test_plain_one_gpu_run_passes_its_stream_through_rccl():
    res = _run([sys.executable, "bench.py"] + SMALL)
    g = res["gather_ms"]
    assert g["world_size"] == 1 and g["rccl_ok"] is True and g["rccl"] > 0, g
    assert res["config"]["gather"] == "none"  # nothing of it inside the timed region
    print("plain:", g)
