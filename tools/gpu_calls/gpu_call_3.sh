set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
(time timeout 300 python -m pytest tests/test_query_gpu.py tests/test_zslab_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q) > gpurun_out/r02a/pytest3.log 2>&1
tail -5 gpurun_out/r02a/pytest3.log
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline 0 --scene-b 0 > gpurun_out/r02a/bench3.json 2> gpurun_out/r02a/bench3.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r02a/bench3.json').read().strip().splitlines()[-1])
print(j['extras'])
PY
