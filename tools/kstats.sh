#!/bin/bash
# Development aid (GPU box): per-kernel mean durations of one config via rocprofv3 --kernel-trace --stats.
#   tools/kstats.sh <tag> <cfg> ["quick_bench settings"]
TAG=$1; CFG=${2:-c3}; SET=${3:-}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python $ROOT/tools/quick_bench.py $CFG 10 "$SET" > $OUT/kt.log 2>&1 || tail -5 $OUT/kt.log
cd $ROOT
F=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$F")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel stats ($CFG $SET): calls, mean us, share")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):5d} {float(r["AverageNs"])/1e3:9.1f} {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
cp $F $OUT/kernel_stats.csv; rm -rf $OUT/kt
