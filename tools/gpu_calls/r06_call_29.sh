#!/bin/bash
# round 6, call 29: marching cubes after the old emit path was removed -- every test that meshes, then reconstruct by phase
O=gpurun_out/r06_c29; mkdir -p $O
timeout 900 python -m pytest tests/test_query_gpu.py tests/test_lab_gpu.py tests/test_multi_gpu.py tests/test_zslab_gpu.py tests/test_zslab_hip_ranks_gpu.py tests/test_dropin_gpu.py tests/test_programs_gpu.py tests/test_fullsize_gpu.py tests/test_baseline_configs_gpu.py::test_config3_2048_cubed_colour_through_weight_saturation_then_mesh tests/test_baseline_configs_gpu.py::test_config0_256_cubed_one_frame_equals_the_reference_itself -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
for i in 1 2 3; do timeout 200 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --host-path 0 --scene-b 0 --keys 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extras']
print({k: e[k] for k in e if 'reconstruct' in k})" ; done | tee $O/mc_timing.txt
