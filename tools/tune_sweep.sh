#!/bin/bash
# Launch-shape sweep of k_integrate at the headline size (run via gpurun from the repo root).
# Appends "<settings> -> ms_per_step frac frac_layout" lines to gpurun_out/tune_sweep.log.
mkdir -p gpurun_out
run() { printf "%s color=%s layout=%s -> " "$1" "$2" "$3" >> gpurun_out/tune_sweep.log; env $1 timeout 300 python bench.py --steps 30 --warmup 4 --cpu-baseline 0 --extras 0 --color $2 --layout $3 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],3), round(j['roofline']['frac'],3), round(j['roofline']['frac_layout'],3))" >> gpurun_out/tune_sweep.log; }
for C in 1 0; do
  for R in 8 16 32 64 128; do run "TSDF_HIP_ROWS_PER_BLOCK=$R" $C packed; done
  run "TSDF_HIP_FAST_PROJECTION=0" $C packed
  run "TSDF_HIP_FAST_PROJECTION=1" $C packed
done
cat gpurun_out/tune_sweep.log
