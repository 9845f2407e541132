#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters of one rocprofv3 --pmc pass (tools/sq_counters.sh): vector / scalar / LDS instructions and wave
cycles per dispatch, and per 64-position chunk for the distance-coding kernels.

    python tools/sq_summary.py DIR --kind text --n 100000000 > profiles/rNN_pmc_sq_text.json"""
import argparse
import csv
import glob
import json
import os
import re
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("directory")
    ap.add_argument("--kind", default="")
    ap.add_argument("--n", type=int, default=0)
    args = ap.parse_args()
    paths = glob.glob(os.path.join(args.directory, "**", "*counter_collection.csv"), recursive=True)
    if not paths:
        sys.exit("no counter CSV under " + args.directory)
    per = {}
    for path in paths:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                m = re.search(r"(k_[a-z0-9_]+)", row["Kernel_Name"])
                if not m:
                    continue
                k = per.setdefault(m.group(1), {"dispatches": set(), "counters": {}})
                k["dispatches"].add(row["Dispatch_Id"])
                k["counters"][row["Counter_Name"]] = k["counters"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    out = {"tool": "tools/sq_counters.sh (rocprofv3 --pmc, counters only)", "input": args.kind, "n": args.n, "kernels": {}}
    chunks = args.n / 64.0 if args.n else None
    for name, k in sorted(per.items()):
        d = len(k["dispatches"])
        e = {"dispatches": d}
        for c, v in sorted(k["counters"].items()):
            e[c + "_per_dispatch"] = round(v / d, 1)
        if chunks and name.startswith("k_dc_"):
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
                if c in k["counters"]:
                    e[c + "_per_64_positions"] = round(k["counters"][c] / d / chunks, 1)
        out["kernels"][name] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
