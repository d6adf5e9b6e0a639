#!/bin/bash
# marching cubes with one word per active cell (keys-only radix sort): parity tests, then reconstruct timing by phase at 2048^3
O=gpurun_out/r06_c11; mkdir -p $O
timeout 900 python -m pytest tests/test_query_gpu.py tests/test_lab_gpu.py tests/test_multi_gpu.py tests/test_zslab_gpu.py tests/test_dropin_gpu.py tests/test_programs_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
for i in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --host-path 0 --scene-b 0 --keys 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extras']
print({k: e[k] for k in e if 'reconstruct' in k or 'march' in k or 'mc_' in k})" ; done | tee $O/mc_timing.txt
