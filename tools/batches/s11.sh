set -u
for v in k2nopipe k2pipe4 k2pipe3 k2nopipe k2pipe4 k2pipe3; do
  TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --host-path 0 --scene-b 0 > gpurun_out/bench_s11_$v.json 2>/dev/null
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
d=[json.loads(l) for l in open(f'gpurun_out/bench_s11_{v}.json').read().splitlines() if l.startswith('{')][-1]
f=d['extras']['fused2']
print(v, 'single',round(d['roofline']['kernel_ms'],2),'fused ms/frame',round(f['ms_per_frame'],3), f['one_sweep_per_pair'], 'frac', round(f['frac_of_hbm_peak'],3))
PY
done
echo "== fused tests on k2pipe4"
TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/k2pipe4/libtsdf_hip.so timeout 600 python -m pytest tests/test_fused2_gpu.py -q -p no:cacheprovider 2>&1 | tail -3
