#!/bin/bash
O=gpurun_out/r05_c5; mkdir -p $O
python tools/gpu_calls/diag_rgb.py 2>&1 | grep -v amdgpu.ids | head -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(timeout 2400 python -m pytest tests/test_zslab_hip_ranks_gpu.py tests/test_evidence_gpu.py tests/test_integrate_gpu.py tests/test_fused2_gpu.py tests/test_implied_d_gpu.py -m gpu -q --durations=12 2>&1 | tail -40) | tee $O/pytest_new.txt
