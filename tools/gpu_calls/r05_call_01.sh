#!/bin/bash
# round 5, call 1: parity of the rebuilt integrate kernels, then A/B of the instruction-diet variants against the round-4 library
mkdir -p gpurun_out/r05_c1
(timeout 900 python -m pytest tests/test_integrate_gpu.py tests/test_fused2_gpu.py tests/test_implied_d_gpu.py tests/test_div_gpu.py -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r05_c1/pytest.txt
tail -5 gpurun_out/r05_c1/pytest.txt
for rep in 1 2; do for n in r4 v16 v16_pk2 v16_noktab v16_nof2; do
  TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/$n/libtsdf_hip.so timeout 300 python bench.py --steps 20 --warmup 3 --extras 2 --cpu-baseline 0 --host-path 0 > gpurun_out/r05_c1/$n.$rep.json 2>> gpurun_out/r05_c1/err.log || echo "$n failed"
  python - "$n" gpurun_out/r05_c1/$n.$rep.json <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[2])); f2=d.get('extras',{}).get('fused2',{})
    print(f"{sys.argv[1]:12s} kernel_ms {d['roofline']['kernel_ms']:.3f}  ms_per_step {d['ms_per_step']:.3f} fused2 {f2.get('ms_per_frame')} place {d['config']['plane_placement']}")
except Exception as e: print(sys.argv[1], "no result", e)
P
done; done | tee gpurun_out/r05_c1/summary.txt
for n in r4 v16; do
  TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/$n/libtsdf_hip.so timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 --color 0 > gpurun_out/r05_c1/$n.c0.json 2>> gpurun_out/r05_c1/err.log
  python -c "
import json,sys; d=json.load(open('gpurun_out/r05_c1/$n.c0.json')); print('$n', 'color=0', d['roofline']['kernel_ms'])" | tee -a gpurun_out/r05_c1/summary.txt
done
