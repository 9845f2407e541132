"""suffix sort on pathological inputs (long runs, periodic text, Fibonacci words ...): correctness against the oracle and timings"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dark_amd
from oracle import orc
cases = {
  "ab*2^23": np.frombuffer(b"ab" * (1 << 23), np.uint8),
  "fib-like": None,
  "a^n b": np.concatenate([np.zeros((1 << 22) - 1, np.uint8), np.ones(1, np.uint8)]),
  "period 1000": np.tile(np.random.default_rng(1).integers(0, 256, 1000, dtype=np.uint8), 8000),
}
# Fibonacci string (highly repetitive, many LCP levels)
a, b = b"a", b"ab"
while len(b) < (1 << 22): a, b = b, b + a
cases["fib-like"] = np.frombuffer(b[: 1 << 22], np.uint8)
with dark_amd.Context(1 << 24) as ctx:
    for name, t in cases.items():
        t = np.ascontiguousarray(t)
        t0 = time.time(); sa = ctx.suffix_array(t); dt = time.time() - t0
        st = ctx.stats()
        t1 = time.time(); want = orc.sa_sais(t); dto = time.time() - t1
        print(name, len(t), "gpu %.3fs rounds %d passes %d | oracle %.2fs | equal %s" % (dt, st["rounds"], st["sort_passes"], dto, bool((sa == want).all())), flush=True)
