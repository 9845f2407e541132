#!/usr/bin/env python3
"""Encode a user-supplied corpus file (book1, enwik8 ...) as ONE block through the GPU path and report the size of the file
the reference would write: 4-byte n header (src/main.rs:102) + coded stream.

    python tools/check_corpus.py PATH [--model dark] [--expect-size N] [--no-oracle]

The reference publishes exactly one such number: Calgary book1 (768 771 B) -> 214 445 B with `-m dark` (README.md:20).  book1 is
not in the image; on a machine that has it this is the only available pin for the DC / range-coder half of the oracle
(SURVEY.md 8c).  Also checked: decode gives the file back, and (unless --no-oracle) the stream equals the CPU oracle's.
Exit code 0 = every check that could run passed; 1 = a mismatch; 2 = usage / environment."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KNOWN = {("book1", 768771, "dark"): 214445}  # README.md:20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--model", default="dark")
    ap.add_argument("--expect-size", type=int, default=0, help="expected size of the .dark file (header + stream)")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    data = np.fromfile(args.path, dtype=np.uint8)
    n = len(data)
    if n == 0 or n > 0x7FFFFFFE:
        print("file must hold 1 .. 2^31-2 bytes", file=sys.stderr)
        return 2
    import dark_amd
    expect = args.expect_size or KNOWN.get((os.path.basename(args.path), n, args.model), 0)
    with dark_amd.Context(n, device=args.device) as ctx:
        stream = ctx.block_encode(args.model, data)
        has_ff = bool((data == 255).any())
        back_ok = None
        if not has_ff:
            back_ok = ctx.block_decode(args.model, stream, n) == data.tobytes()
        st = ctx.stats()
    report = {"file": args.path, "bytes": n, "model": args.model, "stream_bytes": len(stream), "dark_file_bytes": len(stream) + 4,
              "expected_dark_file_bytes": expect or None, "size_matches_reference": (len(stream) + 4 == expect) if expect else None,
              "roundtrip_ok": back_ok, "contains_0xFF": has_ff, "sa_rounds": st["rounds"]}
    if not args.no_oracle:
        from oracle import orc  # the checker
        report["stream_equals_oracle"] = bool(orc.block_dc_encode(args.model, data) == stream)
    print(json.dumps(report))
    bad = (report["size_matches_reference"] is False) or (back_ok is False) or (report.get("stream_equals_oracle") is False)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
