#!/bin/bash
# eight waves per SIMD for the ALLIN PACKED instances (64 VGPRs: fits without a spill on the row path since round 5's diet)
O=gpurun_out/r05_c19; mkdir -p $O
run() { env $3 timeout 120 python bench.py --steps 20 --warmup 3 --extras $4 --cpu-baseline 0 --host-path 0 $2 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); f2=(d.get('extras') or {}).get('fused2') or {}
    print('$1 $2', round(d['roofline']['kernel_ms'],3), 'fused2', f2.get('ms_per_frame'), d['config']['plane_placement']['probe_sweep_ms'][-1])
except Exception as e: print('$1 $2 failed', e)"; }
for rep in 1 2; do
run shipped "--color 1" "X=1" 0; run w8 "--color 1" "TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/w8/libtsdf_hip.so" 0
done | tee $O/summary.txt
run shipped "--color 0" "X=1" 0 | tee -a $O/summary.txt; run w8 "--color 0" "TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/w8/libtsdf_hip.so" 0 | tee -a $O/summary.txt
run shipped "--res 4096 --planes 512 --width 1280 --height 960" "X=1" 0 | tee -a $O/summary.txt; run w8 "--res 4096 --planes 512 --width 1280 --height 960" "TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/w8/libtsdf_hip.so" 0 | tee -a $O/summary.txt
(TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/w8/libtsdf_hip.so timeout 300 python -m pytest tests/test_integrate_gpu.py tests/test_implied_d_gpu.py -m gpu -q -x 2>&1 | tail -3) | tee -a $O/summary.txt
