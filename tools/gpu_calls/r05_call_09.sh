#!/bin/bash
O=gpurun_out/r05_c9; mkdir -p $O
for k in 1 2 3 4; do
timeout 600 python -m pytest tests/test_zslab_hip_ranks_gpu.py -m gpu -q 2>&1 | tail -3
done
timeout 900 python -m pytest tests/test_zslab_gpu.py tests/test_bench_dist_gpu.py tests/test_fused2_gpu.py -m gpu -q 2>&1 | tail -5
