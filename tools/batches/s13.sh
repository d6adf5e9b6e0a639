set -u
echo "== query / multi / zslab / dropin tests (chain 4)"
timeout 1500 python -m pytest tests/test_query_gpu.py tests/test_multi_gpu.py tests/test_zslab_gpu.py tests/test_dropin_gpu.py tests/test_product_lib_gpu.py -q -p no:cacheprovider 2>&1 | tail -5
rv() { timeout 600 python bench.py --steps 4 --warmup 2 --cpu-baseline 0 --host-path 0 --scene-b 0 $2 > gpurun_out/bench_s13_$1.json 2>/dev/null; python - "$1" <<'PY'
import json,sys
d=[json.loads(l) for l in open(f'gpurun_out/bench_s13_{sys.argv[1]}.json').read().splitlines() if l.startswith('{')][-1]
e=d['extras']; print(sys.argv[1], 'renderView_ms', round(e['renderView_ms'],3), 'steps', e['renderView_mean_steps'], 'hits', e['renderView_hits'])
PY
}
echo "== renderView at 2048^3"
rv chain4 ""
for v in rc1 rc2 rc8; do TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so rv $v ""; done
echo "== renderView at 1024^3"
rv chain4_1024 "--res 1024"
TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/rc1/libtsdf_hip.so rv rc1_1024 "--res 1024"
TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/rc8/libtsdf_hip.so rv rc8_1024 "--res 1024"
