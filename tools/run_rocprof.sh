#!/bin/bash
# Profile the integrate kernel on the GPU box (run via gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats over the SAME command the bench line comes from (bench.py)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ busy/wait) over tools/prof_integrate.py,
#      which also launches k_calib_rmw sweeps of exactly known bytes for calibration.
# Outputs land in gpurun_out/prof_<tag>/ ; tools/pmc_reduce.py turns them into JSON summaries.
set -u
TAG=${1:-r01c}
STEPS=${2:-20}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace -o bench --output-format csv -- \
  python $ROOT/bench.py --steps $STEPS --warmup 2 --cpu-baseline 0 --scene-b 0 > $ROOT/$OUT/bench_under_rocprof.json 2> $ROOT/$OUT/bench_under_rocprof.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $ROOT/$OUT/pmc_$C -o pmc --output-format csv -- \
    python $ROOT/tools/prof_integrate.py --steps 4 --warmup 1 --calib 2 > $ROOT/$OUT/prof_$C.json 2> $ROOT/$OUT/prof_$C.err
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
  -d $ROOT/$OUT/pmc_SQ -o pmc --output-format csv -- \
  python $ROOT/tools/prof_integrate.py --steps 4 --warmup 1 --calib 1 > $ROOT/$OUT/prof_SQ.json 2> $ROOT/$OUT/prof_SQ.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD TCC_HIT_sum TCC_MISS_sum \
  -d $ROOT/$OUT/pmc_SQ2 -o pmc --output-format csv -- \
  python $ROOT/tools/prof_integrate.py --steps 4 --warmup 1 --calib 1 > $ROOT/$OUT/prof_SQ2.json 2> $ROOT/$OUT/prof_SQ2.err
cd $ROOT
for d in trace pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ pmc_SQ2; do
  python tools/pmc_reduce.py $OUT/$d > $OUT/summary_$d.json 2>> $OUT/reduce.err
done
find $OUT -name "*_kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# the raw per-dispatch CSVs are large; keep only the summaries + the stats table
find $OUT -name "*.csv" ! -name "kernel_stats.csv" -size +2M -delete
ls -la $OUT
