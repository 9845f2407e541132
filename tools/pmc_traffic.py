#!/usr/bin/env python3
"""HBM traffic per kernel from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are collected in
SEPARATE passes of the same command, values are KiB, and on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming
reads (corrected x2 for the streaming kernels listed below, raw for kernels dominated by random 1/4-byte gathers).

On the GPU box (run from /tmp, TMPDIR=/tmp; the program itself follows `--`):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_FETCH_SIZE -- python3 bench.py --steps 2 --warmup 1 \
              --no-cpu-baseline --no-decode --pipeline-blocks 0
    rocprofv3 --pmc WRITE_SIZE ... -d gpurun_out/pmc_WRITE_SIZE -- (same)
then here:
    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE --steps 3 --workload enwik8_like_1e8 \
           > profiles/rNN_pmc_traffic_<workload>.json
(--steps = warm-up + timed steps the profiled command executed, so that figures are per step.)
"""
import argparse
import csv
import glob
import json
import os
import re
import sys

# kernels whose reads are random narrow gathers: one 64-byte request per access, FETCH_SIZE is taken as reported
GATHER_KERNELS = {"k_round_local", "k_plateau_sort", "k_plateau_ranks", "k_bwt_gather", "k_radix_scan",
                  "k_big_reduce", "k_big_spine", "k_big_apply", "k_big_back", "k_rerank_scan", "k_dc_carry_a", "k_dc_carry_b", "k_dc_carry_c",
                  "k_dc_runscan", "k_dc_init", "k_fill_u32", "k_place_active", "k_rank_active", "k_prefix_probe", "k_to_inplace",
                  "k_plateau_scan", "k_chain_ends", "k_chain_verdicts", "k_chain_apply", "k_ibwt_walk", "k_ibwt_emit", "k_ibwt_jump", "k_isa_init",
                  "k_lf_finish", "k_lf_deep", "k_lf_straddle", "k_lf_deep_wave", "k_lf_deep_block", "k_lf_periodic_groups", "k_lf_lce", "k_lf_list_to_arena"}

# dk_stats slot (dark_amd/csrc/context.hpp) of every kernel: a slot is named after its kernel, or after the common prefix of the kernels
# one LaunchScope brackets; bench.py reports HIP-event times per slot, the rocprofv3 CSVs are per kernel
SLOT_OF = {"k_chain_extract": "k_chain", "k_chain_ends": "k_chain", "k_chain_tiles": "k_chain", "k_chain_spine": "k_chain", "k_chain_verdicts": "k_chain",
           "k_chain_apply": "k_chain", "k_rerank_apply_first": "k_rerank_apply", "k_rerank_scan_a": "k_rerank_scan", "k_rerank_scan_c": "k_rerank_scan",
           "k_dc_runscan": "k_dc_carry", "k_dc_carry_a": "k_dc_carry", "k_dc_carry_b": "k_dc_carry", "k_dc_carry_c": "k_dc_carry",
           "k_fill_u32": "k_dc_carry", "k_ibwt_scan_a": "k_ibwt_hist", "k_ibwt_scan_b": "k_ibwt_hist", "k_ibwt_scan_c": "k_ibwt_hist",
           "k_big_reduce": "k_big_classify", "k_big_spine": "k_big_classify", "k_big_apply": "k_big_classify",
           "k_rank_active": "k_place_active", "k_to_inplace": "k_plateau_ranks", "k_plateau_count": "k_plateau_ranks",
           "k_plateau_scan": "k_plateau_ranks", "k_plateau_compact": "k_plateau_ranks",
           "k_isa_init": "k_isa_partition", "k_isa_split": "k_isa_partition",
           "k_lf_reduce": "k_rerank_reduce", "k_lf_straddle": "k_rerank_scan", "k_lf_apply": "k_rerank_apply", "k_lf_medium": "k_chain", "k_lf_deep": "k_chain",
           "k_lf_deep_wave": "k_chain", "k_lf_deep_block": "k_chain", "k_lf_lce": "k_chain", "k_lf_list_to_arena": "k_chain", "k_lf_periodic_groups": "k_period",
           "k_period_first": "k_period", "k_period_spine": "k_period", "k_period_fill": "k_period", "k_period_search": "k_period", "k_period_count": "k_period",
           "k_cls_reduce": "k_big_classify", "k_cls_spine": "k_big_classify", "k_cls_apply": "k_big_classify", "k_sort_groups": "k_radix_sort_small",
           "k_run_probe": "k_sym_hist", "k_period_probe": "k_sym_hist"}


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    if not m:
        return None
    k = m.group(1)
    # the first pass of the initial sort is its own template instance (keys from the text) and its own dk_stats slot
    args = m.group(2).replace(" ", "") if m.group(2) else ""
    if (k == "k_radix_scatter" and args.startswith("<false,true")) or (k == "k_radix_hist" and args == "<3>"):  # <PAIRS, TEXT, BLOCK, PACKED>; HS_TEXT = 3
        k += "_text"
    return k


def collect(directory, counter):
    per = {}
    paths = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if len(paths) != 1:
        sys.exit("%s: expected the counter CSV of exactly one profiled process, found %d (stale files of an earlier pass?)" % (directory, len(paths)))
    for path in paths:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                k = short(row["Kernel_Name"])
                if not k:
                    continue
                e = per.setdefault(k, [0, 0.0])
                e[0] += 1
                e[1] += float(row["Counter_Value"]) * 1024.0  # KiB -> bytes
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--steps", type=int, required=True, help="steps (warm-up + timed) the profiled command ran")
    ap.add_argument("--workload", required=True)
    ap.add_argument("--round", type=int, default=5)
    ap.add_argument("--bench-line", default="", help="the JSON line the profiled command printed (kernel-trace pass): its kernel_launches_per_step -- the library's own "
                                                     "count of bracketed launches per slot -- goes into the profile, bench.py checks later runs against it")
    args = ap.parse_args()
    fetch = collect(args.fetch_dir, "FETCH_SIZE")
    write = collect(args.write_dir, "WRITE_SIZE")
    kernels = {}
    total = 0.0
    for k in sorted(set(fetch) | set(write)):
        fl, fb = fetch.get(k, [0, 0.0])
        wl, wb = write.get(k, [0, 0.0])
        launches = max(fl, wl)
        corr = 1.0 if k in GATHER_KERNELS else 2.0
        hbm = fb * corr + wb
        total += hbm / args.steps
        kernels[k] = {"slot": SLOT_OF.get(k, k), "launches_per_step": launches / args.steps, "fetch_raw_bytes_per_step": round(fb / args.steps),
                      "fetch_correction": corr, "write_bytes_per_step": round(wb / args.steps),
                      "hbm_bytes_per_launch": round(hbm / launches) if launches else 0,
                      "raw_hbm_bytes_per_launch": round((fb + wb) / launches) if launches else 0}  # the counters as printed, no correction
    kernels = dict(sorted(kernels.items(), key=lambda kv: -(kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_per_step"])))
    # per slot, for bench.py: HBM bytes per step of all the kernels a slot brackets
    slots = {}
    for k, v in kernels.items():
        e = slots.setdefault(v["slot"], {"hbm_bytes_per_step": 0, "launches_per_step": 0.0})
        e["hbm_bytes_per_step"] += round(v["hbm_bytes_per_launch"] * v["launches_per_step"])
        e["launches_per_step"] += v["launches_per_step"]
    scope_launches = None
    if args.bench_line:
        with open(args.bench_line) as f:
            for line in f:
                if line.startswith("{"):
                    scope_launches = json.loads(line).get("kernel_launches_per_step")
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py "
                          "--workload %s --steps S --warmup W --no-cpu-baseline --no-decode --pipeline-blocks 0" % args.workload,
               "slots": slots,
               "scope_launches_per_step": scope_launches,
               "workload": args.workload, "round": args.round,
               "unit_note": "counter values are KiB; FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads on gfx950 "
                            "(MI355X_MICROARCH.md, HBM section): fetch_corrected = 2 x raw for streaming kernels, raw for random 4-byte / "
                            "1-byte gathers (one 64-byte request per access)",
               "kernels": kernels, "hbm_bytes_per_step_all_kernels": round(total)}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
