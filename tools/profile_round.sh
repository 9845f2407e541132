#!/bin/bash
# Runs on the GPU box (through gpurun, from the repo root): rocprofv3 kernel-trace statistics and the two PMC passes of the bench
# command for one workload; everything lands under gpurun_out/prof_<workload>/.  The PMC passes are separate runs with counters only
# (no trace domains).  bench.py runs WARMUP + STEPS steps with HIP-event pairs and 3 more without: all of them are in the counters.
# usage: tools/profile_round.sh WORKLOAD [STEPS] [WARMUP]   (results: gpurun_out/prof_<workload>/, to be copied to profiles/rNN_*)
set -e -o pipefail
WL=${1:-enwik8_like_1e8}
STEPS=${2:-2}
WARM=${3:-1}
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
OUT=$ROOT/gpurun_out/prof_$WL
rm -rf $OUT
mkdir -p $OUT
ARGS="--workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline --pipeline-blocks 0 --no-decode --no-rccl-selftest --no-host-input"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py $ARGS > $OUT/bench_profiled.json 2> $OUT/stats.err
echo "$WL kernel-trace pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_FETCH_SIZE -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
echo "$WL FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_WRITE_SIZE -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1
echo "$WL WRITE_SIZE pass done"
cd $ROOT
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --steps $((STEPS + WARM + 3)) --workload $WL --round ${ROUND:-5} --bench-line $OUT/bench_profiled.json > $OUT/pmc_traffic.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# keep the merge small: the raw counter CSVs are large
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/stats
ls -la $OUT
