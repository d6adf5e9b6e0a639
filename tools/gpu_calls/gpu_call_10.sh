cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
(timeout 600 python -m pytest tests/test_query_gpu.py tests/test_zslab_gpu.py tests/test_multi_gpu.py tests/test_dropin_gpu.py tests/test_programs_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q) 2>&1 | tail -4
timeout 300 python tools/mc_probe.py 2>&1 | tail -1
timeout 300 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --scene-b 0 > gpurun_out/r02b/bench10.json 2> gpurun_out/r02b/bench10.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r02b/bench10.json').read().strip().splitlines()[-1])
print({k:v for k,v in j['extras'].items() if 'reconstruct' in k or 'renderView_ms' in k})
PY
