#!/bin/bash
for n in y_lean y_lean_p2 y_pk2 y_lean_e1; do echo "== $n"; TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/$n/libtsdf_hip.so python tools/gpu_calls/diag_rgb.py 2>&1 | grep -v amdgpu.ids | head -4; done
echo "== default test lib"; TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/libtsdf_hip_test.so python tools/gpu_calls/diag_rgb.py 2>&1 | grep -v amdgpu.ids | head -4
echo "== default product lib"; python tools/gpu_calls/diag_rgb.py 2>&1 | grep -v amdgpu.ids | head -4
