"""Regenerates tests/golden/vectors.json.

Two kinds of vectors live in that file:
  * "reference": copied verbatim from the reference's own known-answer test
    (/root/reference/src/saca.rs:411-412).  These pin the oracle.
  * "oracle": produced by oracle/dark_oracle.c (cross-checked against the naive suffix sort where a
    ground truth exists).  For the DC / range-coder stages the reference holds no golden bytes at all
    (roundtrip tests only), so these are regression vectors of the restatement: PARITY UNPINNED.
Inputs are the two byte strings the reference's tests use: b"abracababra" (src/block/dc.rs:189) and the
reference's LICENSE file (src/saca.rs:431, src/block/dc.rs:174), committed as LICENSE.txt (data fixture).
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import orc  # noqa: E402


def main():
    out = {"reference": {}, "oracle": {}}
    out["reference"]["saca_rs_411_412"] = [
        {"input": "abracadabra", "sa": [10, 7, 0, 3, 5, 8, 1, 4, 6, 9, 2], "origin": 2, "bwt": "rdarcaaaabb"},
        {"input": "banana", "sa": [5, 3, 1, 0, 4, 2], "origin": 3, "bwt": "nnbaaa"},
    ]
    lic = open(os.path.join(HERE, "LICENSE.txt"), "rb").read()
    inputs = {"abracababra": b"abracababra", "LICENSE": lic}
    for name, data in inputs.items():
        sa = orc.sa_sais(data)
        assert (sa == orc.sa_naive(data)).all()
        bwt, origin = orc.bwt_forward(data, sa)
        assert orc.bwt_inverse(bwt, origin).tobytes() == data
        dc = orc.dc_encode(bwt)
        entry = {
            "n": len(data),
            "input_sha256": hashlib.sha256(data).hexdigest(),
            "sa": [int(x) for x in sa],
            "bwt_hex": bwt.tobytes().hex(),
            "origin": origin,
            "dc_init": [int(x) for x in dc["init"]],
            "dc_d": [int(x) for x in dc["d"]],
            "dc_sym_hex": dc["sym"].tobytes().hex(),
            "dc_rank_hex": dc["rank"].tobytes().hex(),
            "rawdc_records_hex": orc.block_dc_encode("rawdc", data).hex(),
            "streams_hex": {m: orc.block_dc_encode(m, data).hex() for m in ("dark", "exp", "ybs", "simple")},
        }
        for m, hx in entry["streams_hex"].items():
            assert orc.block_dc_decode(m, bytes.fromhex(hx), len(data)) == data
        out["oracle"][name] = entry
    # entropy::ari::Range roundtrip vector (src/entropy/ari.rs:76-107): bytes [1,84,15,91], LSB first, p = 1/2
    bits = [(b >> i) & 1 for b in (1, 84, 15, 91) for i in range(8)]
    flat = [2048] * len(bits)
    stream = orc.bitcoder_encode(bits, flat)
    assert list(orc.bitcoder_decode(stream, flat)) == bits
    out["oracle"]["entropy_ari_range"] = {"bytes": [1, 84, 15, 91], "flat": 2048, "stream_hex": stream.hex()}
    # model-level vector (src/model/mod.rs:113-118): four (dist, ctx.symbol) pairs
    d4, s4 = [1, 2, 3, 4], [1, 2, 3, 4]
    out["oracle"]["model_mod_rs_113"] = {
        "d": d4, "sym": s4,
        "streams_hex": {m: orc.model_encode(m, d4, s4).hex() for m in ("dark", "exp", "ybs", "simple")},
    }
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote vectors.json")


if __name__ == "__main__":
    main()
