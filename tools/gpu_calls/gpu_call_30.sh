cd $GRAFT_REPO_ROOT
V=cpu_tsdf_amd/lib/variants
bench() { TSDF_HIP_LIB_PATH=$1 timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --extras 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', 'kernel_ms', j['roofline']['kernel_ms'])"; }
for rep in 1 2; do
  bench cpu_tsdf_amd/lib/libtsdf_hip.so shipped
  for v in maxilp maxocc bias0; do bench $V/$v/libtsdf_hip.so $v; done
done
