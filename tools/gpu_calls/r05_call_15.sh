#!/bin/bash
O=gpurun_out/r05_c15; mkdir -p $O
line() { python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[2])); f2=d.get('extras',{}).get('fused2',{})
    print(f"{sys.argv[1]:22s} kernel_ms {d['roofline']['kernel_ms']:.3f}  ms_per_step {d['ms_per_step']:.3f} fused2 {f2.get('ms_per_frame')} moved_GB {d['roofline']['bytes_moved']['per_launch']/1e9:.2f} frac {d['roofline']['frac']} place {d['config']['plane_placement']['probe_sweep_ms']}")
except Exception as e: print(sys.argv[1], "no result", e)
P
}
(timeout 1200 python -m pytest tests/test_tiles_gpu.py tests/test_integrate_gpu.py tests/test_implied_d_gpu.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -15) | tee $O/pytest.txt
for rep in 1 2; do
  for t in 1 0; do
  TSDF_HIP_TILES=$t timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 > $O/head.$t.$rep.json 2>> $O/err.log; line "colour tiles=$t" $O/head.$t.$rep.json
  TSDF_HIP_TILES=$t timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 --color 0 > $O/c0.$t.$rep.json 2>> $O/err.log; line "no colour tiles=$t" $O/c0.$t.$rep.json
  done
done | tee $O/summary.txt
for t in 1 0; do TSDF_HIP_TILES=$t timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 --res 4096 --planes 512 --width 1280 --height 960 > $O/slab.$t.json 2>> $O/err.log; line "slab tiles=$t" $O/slab.$t.json | tee -a $O/summary.txt; done
python tools/gpu_calls/tile_stats.py 2>&1 | grep -v amdgpu.ids | tail -2
