#!/bin/bash
# Profiles of one round (run ON the GPU box from the repo root): kernel stats of the default bench, FETCH/WRITE PMC passes of one
# launch, SQ counters, and the FETCH/WRITE calibration kernels.  Output under gpurun_out/prof_$TAG; copy the summaries to profiles/.
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline > $OUT/bench_stats.json 2> $OUT/bench_stats.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 bench.py --warmup 0 --steps 1 --no-cpu-baseline > $OUT/bench_pmc_$c.json 2> $OUT/bench_pmc_$c.err || exit 1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib_$c -- ./tools/pmc_calib > $OUT/calib_$c.json 2> $OUT/calib_$c.err || exit 1
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -- python3 bench.py --warmup 0 --steps 1 --no-cpu-baseline > $OUT/bench_pmc_sq1.json 2> $OUT/bench_pmc_sq1.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --warmup 0 --steps 1 --no-cpu-baseline > $OUT/bench_pmc_sq2.json 2> $OUT/bench_pmc_sq2.err || exit 1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -- python3 bench.py --warmup 0 --steps 1 --no-cpu-baseline > $OUT/bench_pmc_tcc.json 2> $OUT/bench_pmc_tcc.err || echo "tcc pass failed"
# the two other bench lines of the round: kernel stats only (their own roofline objects come from the HIP events in bench.py)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_evolT -- python3 bench.py --evolT --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_stats_evolT.json 2> $OUT/bench_stats_evolT.err || echo "evolT stats pass failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sweep -- python3 bench.py --sweep columns --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_stats_sweep.json 2> $OUT/bench_stats_sweep.err || echo "sweep stats pass failed"
find $OUT -name "*.csv" | head -40
