#!/bin/bash
# round 6, call 28: marching cubes emit by vertex (k_mc_emit_v) -- parity, then the two emit kernels alternated on one box
O=gpurun_out/r06_c28; mkdir -p $O
timeout 900 python -m pytest tests/test_query_gpu.py tests/test_lab_gpu.py tests/test_multi_gpu.py tests/test_zslab_gpu.py tests/test_dropin_gpu.py tests/test_programs_gpu.py tests/test_baseline_configs_gpu.py::test_config3_2048_cubed_colour_through_weight_saturation_then_mesh -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest.txt
for i in 1 2 3; do for e in 1 0; do TSDF_HIP_MC_EMIT=$e timeout 200 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --host-path 0 --scene-b 0 --keys 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extras']
print('emit_by_vertex=$e', e['reconstruct_phase_ms'], e['reconstruct_ms'], e['reconstruct_triangles'])" ; done; done | tee $O/mc_emit_ab.txt
