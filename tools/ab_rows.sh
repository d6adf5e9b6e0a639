#!/bin/bash
# launch-shape knobs of k_integrate re-swept on the non-temporal kernel (run-time knobs, shipped library) + gather policy variants
mkdir -p gpurun_out/ab_rows
run() { # name, env...
  local n=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 3 --extras 0 --scene-b 0 --cpu-baseline 0 --host-path 0 > gpurun_out/ab_rows/$n.json 2>> gpurun_out/ab_rows/err.log || echo "$n failed"
  python - "$n" gpurun_out/ab_rows/$n.json <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print(f"{sys.argv[1]:14s} kernel_ms {d['roofline']['kernel_ms']:.3f} placement {d['config']['plane_placement']['probe_sweep_ms']}")
except Exception as e: print(sys.argv[1], "no result", e)
P
}
for rep in 1 2; do
  for r in 8 16 32 64 128; do run rows$r.$rep TSDF_HIP_ROWS_PER_BLOCK=$r; done
  for n in base g_sc1 g_nt g_sc0; do run $n.$rep TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/$n/libtsdf_hip.so; done
done | tee gpurun_out/ab_rows/summary.txt
