"""writes dc_stream.bin (a DC stream of a 16 MB prefix of the enwik8-like workload) for tools/ent_bench.cpp"""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dark_amd import datagen
from oracle import orc
t = datagen.wiki_like(100_000_000, 2)[:16_000_000]
bwt, origin = orc.bwt_forward(t)
dc = orc.dc_encode(bwt)
with open("dc_stream.bin", "wb") as f:
    f.write(struct.pack("<QQI", len(t), len(dc["d"]), origin))
    f.write(dc["init"].tobytes()); f.write(dc["d"].tobytes()); f.write(dc["sym"].tobytes())
print(len(t), len(dc["d"]))
