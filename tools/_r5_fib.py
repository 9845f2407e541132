import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, dark_amd
from oracle import orc
a, b = b"a", b"ab"
while len(b) < (1 << 21):
    a, b = b, b + a
t = np.frombuffer(b[:1 << 21], np.uint8).copy()
wb, wo = orc.bwt_forward(t)
with dark_amd.Context(3 << 20) as c:
    bwt, origin = c.bwt_forward(t)
    st = c.stats()
    d = np.flatnonzero(np.frombuffer(bwt, np.uint8) != np.frombuffer(wb, np.uint8))
    print("origin", origin, wo, "diffs", len(d), d[:10], sorted(st["routes"]), st["rounds"])
