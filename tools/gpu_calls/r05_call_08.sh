#!/bin/bash
O=gpurun_out/r05_c8; mkdir -p $O
for k in 1 2 3; do
TSDF_TEST_TRACE=2 timeout 600 python -m pytest "tests/test_zslab_hip_ranks_gpu.py::test_hip_slabs_in_separate_processes_equal_one_volume[3]" -m gpu -q -s 2>&1 | cut -c1-300 > $O/trace$k.txt
echo "== run $k"; grep -n "rank\|xception\|passed\|failed" $O/trace$k.txt | grep -v "Gloo\|socket" | tail -14
done
