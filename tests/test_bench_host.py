"""bench.py's host-side helpers (no GPU): the real-corpus mode and the per-rank report."""
import os
import sys
import types

import numpy as np

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**kw):
    a = types.SimpleNamespace(corpus="", workload="enwik8_like_1e8", n=0)
    a.__dict__.update(kw)
    return a


def test_corpus_probe_and_blocks(tmp_path, monkeypatch):
    """--corpus PATH, or $DARK_CORPUS_DIR/<book1|enwik8|enwik9> for the workload that stands in for that corpus;
    rank r codes block r of the file cut into blocks of the workload's size."""
    monkeypatch.delenv("DARK_CORPUS_DIR", raising=False)
    assert bench.find_corpus(_args()) == (None, 0)
    data = np.arange(1000, dtype=np.uint32).astype(np.uint8)
    f = tmp_path / "enwik9"
    f.write_bytes(data.tobytes())
    assert bench.find_corpus(_args(corpus=str(f), n=300)) == (str(f), 300)
    for r in range(5):  # three whole blocks of 300; the 100-byte tail is left out, ranks beyond the file wrap around
        assert bench.read_corpus_block(str(f), 300, r).tobytes() == data[300 * (r % 3):300 * (r % 3) + 300].tobytes()
    assert bench.read_corpus_block(str(f), 5000, 2).tobytes() == data.tobytes()  # a block larger than the file = the file
    monkeypatch.setenv("DARK_CORPUS_DIR", str(tmp_path))
    assert bench.find_corpus(_args(workload="enwik9_block_125e6")) == (str(f), 125_000_000)
    assert bench.find_corpus(_args(workload="enwik8_like_1e8")) == (None, 0)          # no enwik8 in the directory
    assert bench.find_corpus(_args(workload="enwik9_block_125e6", n=1234)) == (None, 0)  # --n is a synthetic debug size
    assert bench.BOOK1_DARK_FILE_BYTES == 214445 and bench.CORPUS_OF["book1_like_768771"] == ("book1", 768771)


def test_rank_report_and_summary():
    rep = [bench.rank_report(r, 4, 2.0, {"ms_entropy": 1600.0, "ms_sa": 56.0, "ms_bwt": 0.0, "ms_dc": 4.0, "ms_d2h": 20.0}, [4, t, 4, 4], [8, 8, 8, 8], [16, 16],
                             10**8, 28_000_000) for r, t in ((0, 4), (1, 1))]
    assert rep[0]["ms_per_step"] == 500.0 and rep[0]["ms_entropy"] == 400.0 and rep[0]["ms_device"] == 20.0
    assert rep[1]["host_entropy_threads"] == 1 and rep[1]["host_entropy_threads_max"] == 4
    s = bench.thread_summary(rep, 4)
    assert s["entropy_fallback_ranks"] == [1] and "fewer host threads" in s["entropy_fallback"]
    assert bench.thread_summary(rep[:1], 4)["entropy_fallback"] is None
