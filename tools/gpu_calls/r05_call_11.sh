#!/bin/bash
O=gpurun_out/r05_c11; mkdir -p $O
line() { python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[2])); f2=d.get('extras',{}).get('fused2',{})
    print(f"{sys.argv[1]:22s} kernel_ms {d['roofline']['kernel_ms']:.3f}  ms_per_step {d['ms_per_step']:.3f} fused2 {f2.get('ms_per_frame')} moved_GB {d['roofline']['bytes_moved']['per_launch']/1e9:.2f} frac {d['roofline']['frac']:.3f} place {d['config']['plane_placement']['probe_sweep_ms']}")
except Exception as e: print(sys.argv[1], "no result", e)
P
}
(timeout 900 python -m pytest tests/test_integrate_gpu.py tests/test_implied_d_gpu.py tests/test_fused2_gpu.py tests/test_dropin_gpu.py tests/test_evidence_gpu.py::test_api_sequences_keep_the_implied_distance_record_right -m gpu -q 2>&1 | tail -4) | tee $O/pytest.txt
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 > $O/head.$rep.json 2>> $O/err.log; line "default colour" $O/head.$rep.json
  timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 --color 0 > $O/c0.$rep.json 2>> $O/err.log; line "default no colour" $O/c0.$rep.json
done | tee $O/summary.txt
TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/nopf/libtsdf_hip.so timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 --color 0 > $O/c0.nopf.json 2>> $O/err.log; line "no prefetch, no colour" $O/c0.nopf.json | tee -a $O/summary.txt
timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 --color 0 --res 1024 > $O/c0.1024.json 2>> $O/err.log; line "1024^3 no colour" $O/c0.1024.json | tee -a $O/summary.txt
