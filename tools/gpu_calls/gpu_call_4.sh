set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python - <<'PY'
import ctypes as C
from cpu_tsdf_amd import capi
o=(C.c_int*2)()
print("occupancy rc", capi.load().tsdf_hip_selftest_occupancy_mc(o), list(o))
PY
(timeout 300 python -m pytest tests/test_query_gpu.py -m gpu -x -q) > gpurun_out/r02a/pytest4.log 2>&1
tail -3 gpurun_out/r02a/pytest4.log
for v in default; do
  if [ $v != default ]; then export TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --scene-b 0 > gpurun_out/r02a/bench4_$v.json 2> gpurun_out/r02a/bench4_$v.err
  python - <<PY
import json
j=json.loads(open('gpurun_out/r02a/bench4_$v.json').read().strip().splitlines()[-1])
print("$v", j['extras']['reconstruct_phase_ms'], j['extras']['reconstruct_ms'])
PY
done
