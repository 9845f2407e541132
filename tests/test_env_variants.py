"""Alternative code paths selected by environment variables (read once per process, hence subprocesses):
DK_ENTROPY_THREADS=2 (models and coder on two host threads), DK_SORT=onesweep (single-kernel look-back radix passes),
DK_BUCKETED=0 / DK_XCD=0 (plain rank scatter / plain tile order).  Every variant must give the same bytes."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CPU_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from dark_amd import model, datagen
from oracle import orc
t = datagen.wiki_like(400000, 5)
bwt, origin = orc.bwt_forward(t)
dc = orc.dc_encode(bwt)
for m in ("dark", "exp", "ybs", "simple"):
    s = model.stream_encode(m, len(t), dc["init"], dc["d"], dc["sym"], origin)
    assert s == orc.block_dc_encode_bwt(m, bwt, origin), m
print("ok")
"""

GPU_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import dark_amd
from dark_amd import datagen
from oracle import orc
rng = np.random.default_rng(3)
with dark_amd.Context(6 << 20) as ctx:
    keys = rng.integers(0, 1 << 62, size=300001, dtype=np.uint64)
    vals = np.arange(len(keys), dtype=np.uint32)
    k2, v2 = ctx.dbg_sort_pairs(keys, vals, 0, 64)
    assert (v2 == vals[np.argsort(keys, kind="stable")]).all()
    for t in (datagen.wiki_like(4500000, 7), np.frombuffer(b"ab" * 50000, np.uint8), rng.integers(0, 4, size=300000, dtype=np.uint8)):
        t = np.ascontiguousarray(t)
        assert (ctx.suffix_array(t) == orc.sa_sais(t)).all()
print("ok")
"""


def _run(snippet, env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", snippet % ROOT], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_two_thread_entropy_is_bit_exact():
    _run(CPU_SNIPPET, {"DK_ENTROPY_THREADS": "2"})
    _run(CPU_SNIPPET, {"DK_ENTROPY_THREADS": "1"})


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"DK_SORT": "onesweep"}, {"DK_SORT": "chunked"}, {"DK_BUCKETED": "0"}, {"DK_XCD": "0"}])
def test_gpu_variants_match_oracle(env):
    _run(GPU_SNIPPET, env)
