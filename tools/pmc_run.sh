#!/bin/bash
# Development aid (GPU box): SQ counters of the blend kernels for one config.
#   tools/pmc_run.sh <tag> <cfg> ["quick_bench settings"]   -> gpurun_out/<tag>/sq.json
set -e
TAG=$1; CFG=${2:-c3}; SET=${3:-}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
ROOT=$PWD
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY \
  -d $OUT/pmc_sq --output-format csv -- python $ROOT/tools/quick_bench.py $CFG 3 "$SET" > $OUT/pmc_sq.log 2>&1 || tail -5 $OUT/pmc_sq.log
cd $ROOT
python tools/pmc_summary.py $OUT/sq.json $OUT/pmc_sq > /dev/null
python - <<PY
import json
d = json.load(open("$OUT/sq.json"))
for k, c in d.items():
    if "render_" in k:
        print(k[:48], {n: round(v / 1e6, 1) for n, v in c.items() if n != "launches"})
PY
rm -rf $OUT/pmc_sq
