#!/bin/bash
# round 6, call 39: classify's append loop without its dependent LDS lookup (the case index travels, the triangle count is looked up at the flush)
O=gpurun_out/r06_c39; mkdir -p $O
timeout 900 python -m pytest tests/test_query_gpu.py tests/test_lab_gpu.py tests/test_multi_gpu.py tests/test_zslab_gpu.py tests/test_dropin_gpu.py tests/test_programs_gpu.py tests/test_baseline_configs_gpu.py::test_config3_2048_cubed_colour_through_weight_saturation_then_mesh -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
for i in 1 2 3; do for v in new old; do
  if [ $v = new ]; then unset TSDF_HIP_LIB_PATH; else export TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/mc_old/libtsdf_hip.so; fi
  timeout 200 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --host-path 0 --scene-b 0 --keys 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extras']
print('$v', e['reconstruct_phase_ms'], e['reconstruct_ms'], e['reconstruct_triangles'])"; done; done | tee $O/mc_ab.txt
