#!/bin/bash
# What binds k_integrate?  Timing-only variants of the headline instance (WRONG results: never run tests against them):
# occupancy caps, no stores, no frame gather, no voxel loads.  Build here (tools/ab_bound.sh build), run on the GPU box
# (gpurun -- tools/ab_bound.sh run): one bench line per variant into gpurun_out/ab_bound/.
set -e
cd "$(dirname "$0")/.."
VARIANTS="base: wpe5:-DTSDF_WPE_MAX=5 wpe4:-DTSDF_WPE_MAX=4 wpe3:-DTSDF_WPE_MAX=3 nostore:-DTSDF_EXP_NO_STORE=1 nogather:-DTSDF_EXP_NO_GATHER=1 novload:-DTSDF_EXP_NO_VLOAD=1,-DTSDF_EARLY_VOXEL_LOADS=0 novload_nostore:-DTSDF_EXP_NO_VLOAD=1,-DTSDF_EARLY_VOXEL_LOADS=0,-DTSDF_EXP_NO_STORE=1 ${EXTRA_VARIANTS}"
if [ "$1" = build ]; then
  for v in $VARIANTS; do n=${v%%:*}; f=${v#*:}; python tools/build_variant.py $n ${f//,/ } > /dev/null & done; wait
  ls cpu_tsdf_amd/lib/variants/*/libtsdf_hip.so
else
  mkdir -p gpurun_out/ab_bound
  for rep in ${REPS:-1}; do for v in $VARIANTS; do n=${v%%:*}
    TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/$n/libtsdf_hip.so timeout 200 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 ${BENCH_ARGS} > gpurun_out/ab_bound/$n.$rep.json 2>> gpurun_out/ab_bound/err.log || echo "$n failed"
    python - "$n" gpurun_out/ab_bound/$n.$rep.json <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[2])); print(f"{sys.argv[1]:18s} kernel_ms {d['roofline']['kernel_ms']:.3f}  ms_per_step {d['ms_per_step']:.3f}")
except Exception as e: print(sys.argv[1], "no result", e)
P
  done; done | tee gpurun_out/ab_bound/summary.txt
fi
