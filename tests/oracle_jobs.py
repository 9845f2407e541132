"""TEST INFRASTRUCTURE: the CPU oracle at BASELINE.json's full block sizes, run in background threads while the GPU tests go on.

The oracle (oracle/dark_oracle.c: SA-IS after src/saca.rs:270-340, serial BWT, MTF + dc::encode, `dark` model + range coder) needs
15-20 s for a 1e8-byte block, about a minute at 2^28 and several minutes at 2^30 -- too long to run in line inside the 900 s the
driver gives the -m gpu suite, short enough to hide behind it: conftest.py starts one thread per workload the selected tests name
(ctypes releases the GIL during the C calls; the GPU box has 16 host cores), and a test that needs a result joins that thread.
Only tests/ imports this module."""
import threading
import time

from dark_amd import datagen

# workload -> (generator of the block, code the `dark` stream as well?)  The stream of the random block (byte 0xFF) cannot be decoded
# (src/block/dc.rs:57-73) but it can be compared on encode; coding its 2^30 distances costs the oracle another minute in the background.
SPECS = {
    "enwik8_like_1e8": (lambda: datagen.wiki_like(100_000_000, 2), True),
    "enwik9_block_125e6": (lambda: datagen.wiki_like(125_000_000, 40), True),
    "acgt_2p28": (lambda: datagen.acgt(1 << 28, 3), True),
    "random_2p30": (lambda: datagen.random_bytes(1 << 30, 50), True),
    "wordlike_1e8": (lambda: datagen.word_like(100_000_000, 5), True),
    "realtext_5e7": (lambda: datagen.real_text(50_000_000, 0), True),  # real bytes: the image's own text files (no corpus is on disk)
}
# workloads whose suffix array is kept for a direct comparison: real text holds whole files twice, hundreds of thousands of neighbouring suffixes
# with tens of kilobytes in common -- more than the pair-by-pair check of test_gpu_fullsize.py walks through
KEEP_SA = {"realtext_5e7"}
# small stand-ins with the same code path, for checking this module itself on the CPU (tests/test_oracle.py)
SPECS_SMALL = {
    "small_text": (lambda: datagen.wiki_like(200_000, 2), True),
    "small_random": (lambda: datagen.random_bytes(100_000, 50), False),
}

_jobs = {}
_lock = threading.Lock()


class _Job:
    def __init__(self, name, spec):
        self.name, self.spec = name, spec
        self.result, self.error = None, None
        self.thread = threading.Thread(target=self._run, name="oracle-" + name, daemon=True)
        self.thread.start()

    def _run(self):
        try:
            from oracle import orc
            make, with_stream = self.spec
            t0 = time.perf_counter()
            block = make()
            t1 = time.perf_counter()
            sa = None
            if self.name in KEEP_SA:
                sa = orc.sa_sais(block)
                bwt, origin = orc.bwt_forward(block, sa)
            else:
                bwt, origin = orc.bwt_forward(block)             # SA-IS + the TransformIterator convention
            t2 = time.perf_counter()
            dc = orc.dc_encode(bwt)                              # init[256], d[m], sym[m], rank[m]
            dc.pop("sparse", None)
            dc.pop("limit", None)
            t3 = time.perf_counter()
            stream = orc.block_dc_encode_bwt("dark", bwt, origin) if with_stream else None
            t4 = time.perf_counter()
            self.result = dict(block=block, sa=sa, bwt=bwt, origin=origin, dc=dc, stream=stream,
                               seconds=dict(datagen=t1 - t0, sa_bwt=t2 - t1, dc=t3 - t2, stream=t4 - t3))
        except BaseException as e:  # noqa: BLE001 -- re-raised by get()
            self.error = e


def start(name):
    """idempotent; returns at once"""
    with _lock:
        if name not in _jobs:
            spec = SPECS.get(name) or SPECS_SMALL[name]
            _jobs[name] = _Job(name, spec)
        return _jobs[name]


def get(name):
    """the finished oracle run of `name` (started now if nobody did): dict(block, bwt, origin, dc, stream, seconds)"""
    job = start(name)
    job.thread.join()
    if job.error is not None:
        raise job.error
    return job.result


def drop(name):
    """forget a finished run (its arrays are several GiB at the large sizes)"""
    with _lock:
        job = _jobs.get(name)
        if job is not None and not job.thread.is_alive():
            job.result = None
