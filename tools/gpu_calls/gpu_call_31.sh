cd $GRAFT_REPO_ROOT
bench() { env $2 timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --extras 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'kernel_ms', j['roofline']['kernel_ms'])"; }
for rep in 1 2; do
  bench tx256_rows32 "TSDF_HIP_TX_LOG2_MAX=8"
  bench tx128_rows32 "TSDF_HIP_TX_LOG2_MAX=7"
  bench tx128_rows64 "TSDF_HIP_TX_LOG2_MAX=7 TSDF_HIP_ROWS_PER_BLOCK=64"
  bench tx64_rows32 "TSDF_HIP_TX_LOG2_MAX=6"
  bench tx64_rows64 "TSDF_HIP_TX_LOG2_MAX=6 TSDF_HIP_ROWS_PER_BLOCK=64"
  bench tx64_rows128 "TSDF_HIP_TX_LOG2_MAX=6 TSDF_HIP_ROWS_PER_BLOCK=128"
  bench tx32_rows128 "TSDF_HIP_TX_LOG2_MAX=5 TSDF_HIP_ROWS_PER_BLOCK=128"
done
timeout -s KILL 200 env TSDF_HIP_TX_LOG2_MAX=6 python -m pytest tests/test_integrate_gpu.py -m gpu -x -q 2>&1 | tail -2
