#!/bin/bash
# GPU box: collect everything that goes under profiles/ for one round (run through gpurun; output in gpurun_out/<tag>).
#   tools/round_profiles.sh r06 [quick]
TAG=${1:-r06}
QUICK=${2:-}
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
# 1. bench lines: c3 full (driver contract), the other BASELINE configs single-GPU
timeout 300 python bench.py > $OUT/${TAG}_bench_c3.json 2> $OUT/bench_c3.err
for c in c1 c2 c4 c5; do timeout 200 python bench.py --config $c --no-cpu-baseline --steps 20 --warmup 5 > $OUT/${TAG}_bench_$c.json 2> $OUT/bench_$c.err; done
# 1a'. round 6: the step replayed from a HIP graph at c2 / c3 (the c1 line above carries its own graph_replay), one view as eight tile-row bands
for c in c2 c3; do timeout 200 python bench.py --config $c --graph --no-cpu-baseline --steps 20 --warmup 5 > $OUT/${TAG}_bench_${c}_graph.json 2> $OUT/bench_${c}_graph.err; done
for c in c3 c4 c5; do timeout 300 python bench.py --config $c --band-split 8 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/${TAG}_bench_${c}_bands8.json 2> $OUT/bench_${c}_bands8.err; done
# 1b. the north-star-literal configuration (no matrix pipe anywhere), config c5 with densification on, several views per GPU
timeout 200 python bench.py --no-cpu-baseline --valu > $OUT/${TAG}_bench_c3_valu.json 2> $OUT/bench_c3_valu.err
timeout 300 python bench.py --config c5 --densify-every 5 --steps 40 --warmup 10 > $OUT/${TAG}_bench_c5_densify.json 2> $OUT/bench_c5_densify.err
timeout 300 python bench.py --config c4 --views-per-iter 8 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_c4_8views.json 2> $OUT/bench_c4_8views.err
timeout 300 python bench.py --config c3 --views-per-iter 8 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_c3_8views.json 2> $OUT/bench_c3_8views.err
# 1b'. the LSeg width of the reference's README (NUM_SEMANTIC_CHANNELS 512) on c4's 2M Gaussians, and the fp32 shape of the blend backward
timeout 200 python bench.py --config c4 --feat-dim 512 --no-cpu-baseline --steps 20 --warmup 5 > $OUT/${TAG}_bench_c4_C512.json 2> $OUT/bench_c4_C512.err
F3DGS_BWD_BF16=0 timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/${TAG}_bench_c3_bwd_fp32.json 2> $OUT/bench_c3_bwd_fp32.err
timeout 200 python tools/adam_bench.py > $OUT/${TAG}_adam.txt 2>&1
timeout 200 python tools/adam_bench.py 5000000 128 >> $OUT/${TAG}_adam.txt 2>&1
timeout 200 python tools/feature_loss_bench.py > $OUT/${TAG}_feature_loss.txt 2>&1
timeout 200 python tools/lowres_step_bench.py c4 20 > $OUT/${TAG}_lowres_step.txt 2>&1
timeout 200 python tools/lowres_step_bench.py c3 20 >> $OUT/${TAG}_lowres_step.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w tools/pipe_probe.hip -o /tmp/pipe_probe 2>/dev/null && timeout 60 /tmp/pipe_probe > $OUT/${TAG}_pipe_probe.txt 2>&1
# instruction costs and the blend kernels' inner streams by themselves (round 6)
hipcc --offload-arch=gfx950 -O3 -mcode-object-version=5 -w tools/ubench/inst_rate.hip -o /tmp/inst_rate 2>/dev/null && timeout 120 /tmp/inst_rate > $OUT/${TAG}_inst_rate.txt 2>&1
timeout 60 tools/ubench/blend_stream > $OUT/${TAG}_blend_stream.txt 2>&1
timeout 60 tools/ubench/blend_stream json > $OUT/${TAG}_blend_stream.json 2>/dev/null
timeout 120 python tools/ratio_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_ratio_sweep.txt
python tools/kernel_resources.py > $OUT/${TAG}_kernel_resources.txt 2>&1
if [ -z "$QUICK" ]; then
# 1c. SURVEY 8(d): the reduced c3-shaped PyTorch-CPU run (minutes of host time)
timeout 700 python bench.py --cpu-reduced-c3 > $OUT/${TAG}_cpu_c3_reduced.json 2> $OUT/cpu_c3_reduced.err || echo '{"status": "did not complete in 10 min", "did_not_complete_in_10_min": true}' > $OUT/${TAG}_cpu_c3_reduced.json
# 1d. the parity numbers the three-way comparison prints (flips, adjudication counts, the reference's two builds against each other)
timeout 600 python -m pytest tests/test_gpu_vs_ref.py -q -s -k "full_size or eight_views" 2>&1 | grep -E "^(c[2-5] |c4 view|c4 x|[0-9]+ passed)" > $OUT/${TAG}_parity_adjudication.txt
fi
# 2. kernel trace of the SAME command as the bench (rocprofv3 --kernel-trace --stats)
cd /tmp
F3DGS_BENCH_NO_UBENCH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python $ROOT/bench.py --no-cpu-baseline > $OUT/kt_bench.json 2> $OUT/kt.err
cd $ROOT
python tools/rocprof_summary.py $OUT/kt $OUT/${TAG}_kernel_trace_c3.md
# 3. counters, separate passes (no tracing in the same run)
cd /tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY \
  -d $OUT/pmc_sq --output-format csv -- python $ROOT/tools/quick_bench.py c3 4 > $OUT/pmc_sq.log 2>&1
# LDS side of the blend kernels (round 6: the pixel-lane backward is co-limited by vector issue and LDS)
timeout 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS \
  -d $OUT/pmc_lds --output-format csv -- python $ROOT/tools/quick_bench.py c3 4 > $OUT/pmc_lds.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch --output-format csv -- python $ROOT/tools/quick_bench.py c3 4 > $OUT/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write --output-format csv -- python $ROOT/tools/quick_bench.py c3 4 > $OUT/pmc_write.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT/${TAG}_pmc_sq_counters.json $OUT/pmc_sq
python tools/pmc_summary.py $OUT/${TAG}_pmc_lds_counters.json $OUT/pmc_lds
python tools/pmc_summary.py $OUT/${TAG}_pmc_hbm_raw.json $OUT/pmc_fetch $OUT/pmc_write
python tools/pmc_summary.py --traffic $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_hbm_raw.json
# 3b. decision bias of the product on borderline blend decisions (c4, the two views where round 4 saw the largest product-vs-fp64 medians)
if [ -z "$QUICK" ]; then
for v in 4 5; do timeout 200 python tools/adjudicate_probe.py c4 $v strict 2>&1 | grep -E "^(mode|c4 yaw|adjudicated|decision bias)" >> $OUT/${TAG}_decision_bias_c4.txt; done
fi
# 4. work statistics of the blend kernels (lane utilisation)
timeout 200 python tools/pair_stats.py c3 > $OUT/${TAG}_pair_stats_c3.txt 2>&1
rm -rf $OUT/kt $OUT/pmc_sq $OUT/pmc_lds $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
