cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e
bench() { env $2 timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --extras 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'kernel_ms', j['roofline']['kernel_ms'])"; }
for rep in 1 2; do
  for r in 8 16 32 64 128 256; do bench rows$r TSDF_HIP_ROWS_PER_BLOCK=$r; done
done 2>&1 | tee $O/sweep_rows.txt
