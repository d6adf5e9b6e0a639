#!/bin/bash
O=gpurun_out/r05_c7; mkdir -p $O
TSDF_TEST_TRACE=1 AMD_SERIALIZE_KERNEL=3 timeout 600 python -m pytest "tests/test_zslab_hip_ranks_gpu.py::test_hip_slabs_in_separate_processes_equal_one_volume[3]" -m gpu -q -s 2>&1 | cut -c1-400 > $O/trace.txt
grep -n "rank\|exception\|passed\|failed" $O/trace.txt | head -60
timeout 900 python -m pytest tests/test_evidence_gpu.py::test_api_sequences_keep_the_implied_distance_record_right tests/test_fused2_gpu.py -m gpu -q 2>&1 | cut -c1-600 > $O/pytest.txt
grep -n "^E  \|passed\|failed\|Error" $O/pytest.txt | head -40
