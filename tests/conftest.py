import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_finish(session):
    """Full-size parity tests (tests/test_gpu_fullsize.py) compare with oracle runs that take minutes: start those in background threads
    now, for exactly the workloads the selected tests name -- and only where a GPU will actually run them."""
    wanted = []
    for item in session.items:
        if item.get_closest_marker("gpu") is None or not hasattr(item, "callspec"):
            continue
        w = item.callspec.params.get("workload")
        if isinstance(w, str) and w not in wanted:
            wanted.append(w)
    if not wanted:
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
    except ImportError:
        return
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_jobs
    for w in wanted:
        if w in oracle_jobs.SPECS:
            oracle_jobs.start(w)  # one thread each, all at once: the host has the cores and the longest run decides


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(GOLDEN, "vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def license_bytes():
    with open(os.path.join(GOLDEN, "LICENSE.txt"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc
    _orc.lib()
    return _orc


def seeded_inputs(seed=0, count=40, max_n=2000):
    """Small seeded inputs covering the edge cases the reference's domain has: n=1, runs, tiny and full
    alphabets, periodic text, bytes incl. 0xFF (the A7 header quirk) and all-equal blocks."""
    rng = np.random.default_rng(seed)
    out = [np.array([0], np.uint8), np.array([255], np.uint8), np.zeros(17, np.uint8),
           np.frombuffer(b"abracadabra", np.uint8), np.frombuffer(b"banana", np.uint8),
           np.frombuffer(b"abracababra", np.uint8), np.frombuffer(b"ab" * 300, np.uint8),
           np.frombuffer(b"\x00" * 9 + b"\x01", np.uint8), np.frombuffer(b"\x01" + b"\x00" * 9, np.uint8),
           np.arange(256, dtype=np.uint8), np.arange(255, -1, -1, dtype=np.uint8)]
    for i in range(count):
        n = int(rng.integers(1, max_n))
        k = [1, 2, 3, 4, 16, 96, 256][i % 7]
        t = rng.integers(0, k, size=n, dtype=np.uint8)
        if i % 5 == 0:  # inject long repeats
            seg = t[: max(1, n // 7)].copy()
            t = np.concatenate([t, seg, seg, t[: n // 3]])
        out.append(np.ascontiguousarray(t))
    return out
