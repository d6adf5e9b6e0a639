#!/bin/bash
# Cache-policy A/B of the voxel stream (TSDF_STREAM_LD_AUX / TSDF_STREAM_ST_AUX, tsdf_buffer.h) and re-tuning on top of it.
# Build the variants here (tools/build_variant.py NAME flags...), then: gpurun -- 'VARIANTS="base nt ..." bash tools/ab_nt.sh'
mkdir -p gpurun_out/ab_nt
if [ -n "$TESTS" ]; then (timeout 600 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -40) > gpurun_out/ab_nt/tests.log; tail -40 gpurun_out/ab_nt/tests.log; fi
for rep in ${REPS:-1 2}; do for n in $VARIANTS; do
  TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/$n/libtsdf_hip.so timeout 200 python bench.py --steps 20 --warmup 3 --extras $((rep==1)) --scene-b 0 --cpu-baseline 0 --host-path 0 $BENCH_ARGS > gpurun_out/ab_nt/$n.$rep.json 2>> gpurun_out/ab_nt/err.log || echo "$n failed"
  python - "$n" gpurun_out/ab_nt/$n.$rep.json <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[2])); e=d.get('extras',{})
    print(f"{sys.argv[1]:10s} kernel_ms {d['roofline']['kernel_ms']:.3f} placement {d['config']['plane_placement']['probe_sweep_ms']}", e.get('reconstruct_phase_ms'), e.get('reconstruct_classify_d_bytes_requested'), e.get('renderView_ms'))
except Exception as e: print(sys.argv[1], "no result", e)
P
done; done | tee gpurun_out/ab_nt/summary.txt
