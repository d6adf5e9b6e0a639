cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e
mkdir -p $O
V=cpu_tsdf_amd/lib/variants
bench() { env $3 TSDF_HIP_LIB_PATH=$1 timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --extras 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', 'kernel_ms', j['roofline']['kernel_ms'], 'ms_per_step', j['ms_per_step'], 'frac', j['roofline']['frac'])"; }
for rep in 1 2; do
  bench cpu_tsdf_amd/lib/libtsdf_hip.so rest6
  bench $V/rest5/libtsdf_hip.so rest5
  bench cpu_tsdf_amd/lib/libtsdf_hip.so norest TSDF_HIP_REST_BITS=0
done 2>&1 | tee $O/ab_rest.txt
(timeout -s KILL 300 python -m pytest tests/test_integrate_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q) 2>&1 | tail -5
