cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
for v in default probe1 probe2 zb8 zb128; do
  unset TSDF_HIP_LIB_PATH
  if [ $v != default ]; then export TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so; fi
  timeout 300 python bench.py --steps 4 --warmup 2 --cpu-baseline 0 --scene-b 0 > gpurun_out/r02a/bench5_$v.json 2> gpurun_out/r02a/bench5_$v.err
  python - <<PY
import json
j=json.loads(open('gpurun_out/r02a/bench5_$v.json').read().strip().splitlines()[-1])
print("$v", j['extras'].get('reconstruct_phase_ms'), j['extras'].get('reconstruct_ms'), j['extras'].get('error'))
PY
done
