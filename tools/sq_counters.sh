#!/bin/bash
# Runs on the GPU box (through gpurun, from the repo root): instruction counters of the distance-coding kernels (k_dc_main is bound by
# instruction issue) for the three flavours of input, one rocprofv3 --pmc pass each (counters only, no trace
# domains), summarised per kernel by tools/sq_summary.py into gpurun_out/sq_<kind>.json -- copy those to profiles/rNN_pmc_sq_<kind>.json.
# usage: tools/sq_counters.sh
set -e -o pipefail
ROOT=$(pwd)
export TMPDIR=/tmp REPS=1
cd /tmp
for spec in "text 100000000" "acgt 268435456" "random 1073741824"; do
  set -- $spec
  OUT=$ROOT/gpurun_out/sq_$1
  rm -rf $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT -- python3 $ROOT/tools/stage_time.py dc $2 $1 > $OUT.log 2>&1
  python3 $ROOT/tools/sq_summary.py $OUT --kind $1 --n $2 > $ROOT/gpurun_out/sq_$1.json
  rm -rf $OUT
  echo "$1 done"
done
