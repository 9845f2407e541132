#!/bin/bash
# Runs on the GPU box (through gpurun, from the repo root): rocprofv3 kernel-trace statistics and the two PMC passes of the bench
# command; everything lands under gpurun_out/.  The PMC passes are separate runs with counters only (no trace domains).
set -e -o pipefail
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
OUT=$ROOT/gpurun_out
rm -rf $OUT/prof_stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --pipeline-blocks 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $ROOT/bench.py $ARGS > $OUT/prof_stats.log 2>&1
echo "kernel-trace pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_FETCH_SIZE -- python3 $ROOT/bench.py $ARGS --no-decode > $OUT/pmc_fetch.log 2>&1
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_WRITE_SIZE -- python3 $ROOT/bench.py $ARGS --no-decode > $OUT/pmc_write.log 2>&1
echo "WRITE_SIZE pass done"
find $OUT/prof_stats -name "*kernel_stats.csv" | head -3
