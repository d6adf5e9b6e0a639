#!/bin/bash
# Profile the path on the GPU box (run via gpurun from the repo root).  EVERY pass runs bench.py itself -- the
# command the bench line comes from -- so trace durations and counters belong to the very launches bench.py times:
#   1. rocprofv3 --kernel-trace --stats                      (durations; extras on: k_mc_*, k_raycast appear too)
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes)  HBM bytes; bench.py --calib 2 adds k_calib_rmw sweeps of
#                                                            exactly known bytes to the same process for calibration
#   3. --pmc SQ_* (two passes)                               VALU / wait / instruction mix
# bench.py warms up through the COUNTING instance of k_integrate, so the non-counting instance in every table is
# exactly the timed launches (first-after-reset launch excluded).  Outputs: gpurun_out/prof_<tag>/;
# tools/pmc_reduce.py -> summary_*.json; tools/make_profile_summary.py <tag> copies the judged parts to profiles/.
# (every rocprofv3 command runs under `timeout 400`: a counter pass that aborts inside the tool can otherwise sit until gpurun's limit)
# usage: tools/run_rocprof.sh TAG [STEPS [PMC_STEPS [EXTRA_BENCH_ARGS [lite]]]]
#   EXTRA_BENCH_ARGS  e.g. "--layout f32w", "--color 0", "--res 4096 --planes 512 --width 1280 --height 960": another
#                     pmc_traffic.json key (bench.py quotes roofline.traffic per key);  lite = trace + FETCH + WRITE only, no extras legs;
#                     noextras = all passes, no extras legs (a key whose fused2 leg runs the SAME kernel as the timed launches: no colour)
set -u
TAG=${1:-r03}
STEPS=${2:-20}
PMC_STEPS=${3:-6}
EXTRA=${4:-}
LITE=${5:-}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
BENCH="python $ROOT/bench.py --warmup 2 --cpu-baseline 0 --scene-b 0 --host-path 0 --keys 0 $EXTRA"
[ -n "$LITE" ] && BENCH="$BENCH --extras 0"   # lite: trace + FETCH + WRITE only; noextras: the SQ passes too, without the extras legs
timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace -o bench --output-format csv -- \
  $BENCH --steps $STEPS > $ROOT/$OUT/bench_under_rocprof.json 2> $ROOT/$OUT/bench_under_rocprof.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C -d $ROOT/$OUT/pmc_$C -o pmc --output-format csv -- \
    $BENCH --steps $PMC_STEPS --calib 2 > $ROOT/$OUT/bench_pmc_$C.json 2> $ROOT/$OUT/bench_pmc_$C.err
done
if [ -z "$LITE" ] || [ "$LITE" = noextras ]; then
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
  -d $ROOT/$OUT/pmc_SQ -o pmc --output-format csv -- \
  $BENCH --steps $PMC_STEPS > $ROOT/$OUT/bench_pmc_SQ.json 2> $ROOT/$OUT/bench_pmc_SQ.err
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD TCC_HIT_sum TCC_MISS_sum \
  -d $ROOT/$OUT/pmc_SQ2 -o pmc --output-format csv -- \
  $BENCH --steps $PMC_STEPS > $ROOT/$OUT/bench_pmc_SQ2.json 2> $ROOT/$OUT/bench_pmc_SQ2.err
fi
cd $ROOT
for d in trace pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ pmc_SQ2; do
  [ -d $OUT/$d ] || continue
  python tools/pmc_reduce.py $OUT/$d > $OUT/summary_$d.json 2>> $OUT/reduce.err
done
find $OUT -name "*_kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# the raw per-dispatch CSVs are large; keep only the summaries + the stats table
find $OUT -name "*.csv" ! -name "kernel_stats.csv" -size +2M -delete
ls -la $OUT
