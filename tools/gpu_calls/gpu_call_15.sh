cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
for v in default early6; do
unset TSDF_HIP_LIB_PATH
if [ $v != default ]; then export TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so; fi
timeout 300 python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --extras 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v kernel_ms', round(j['roofline']['kernel_ms'],3))"
done; done
export TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/early6/libtsdf_hip.so
(timeout 600 python -m pytest tests/test_integrate_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q) 2>&1 | tail -2
