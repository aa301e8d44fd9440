#!/bin/bash
# GPU box: collect everything that goes under profiles/ for one round (run through gpurun; output in gpurun_out/<tag>).
#   tools/round_profiles.sh r02
TAG=${1:-r03}
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
# 1. bench lines: c3 full (driver contract), the other BASELINE configs single-GPU
timeout 300 python bench.py > $OUT/${TAG}_bench_c3.json 2> $OUT/bench_c3.err
for c in c1 c2 c4 c5; do timeout 200 python bench.py --config $c --no-cpu-baseline --steps 20 --warmup 5 > $OUT/${TAG}_bench_$c.json 2> $OUT/bench_$c.err; done
# 1b. the north-star-literal configuration (no matrix pipe anywhere) and config c5 with densification on
timeout 200 python bench.py --no-cpu-baseline --valu > $OUT/${TAG}_bench_c3_valu.json 2> $OUT/bench_c3_valu.err
timeout 300 python bench.py --config c5 --densify-every 5 --steps 40 --warmup 10 > $OUT/${TAG}_bench_c5_densify.json 2> $OUT/bench_c5_densify.err
timeout 200 python tools/adam_bench.py > $OUT/${TAG}_adam.txt 2>&1
timeout 200 python tools/adam_bench.py 5000000 128 >> $OUT/${TAG}_adam.txt 2>&1
# 2. kernel trace of the SAME command as the bench (rocprofv3 --kernel-trace --stats)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python $ROOT/bench.py --no-cpu-baseline > $OUT/kt_bench.json 2> $OUT/kt.err
cd $ROOT
python tools/rocprof_summary.py $OUT/kt $OUT/${TAG}_kernel_trace_c3.md
# 3. counters, separate passes (no tracing in the same run)
cd /tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY \
  -d $OUT/pmc_sq --output-format csv -- python $ROOT/tools/quick_bench.py c3 4 > $OUT/pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch --output-format csv -- python $ROOT/tools/quick_bench.py c3 4 > $OUT/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write --output-format csv -- python $ROOT/tools/quick_bench.py c3 4 > $OUT/pmc_write.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT/${TAG}_pmc_sq_counters.json $OUT/pmc_sq
python tools/pmc_summary.py $OUT/${TAG}_pmc_hbm_raw.json $OUT/pmc_fetch $OUT/pmc_write
python tools/pmc_summary.py --traffic $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_hbm_raw.json
# 4. work statistics of the blend kernels (lane utilisation)
timeout 200 python tools/pair_stats.py c3 > $OUT/${TAG}_pair_stats_c3.txt 2>&1
rm -rf $OUT/kt $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
