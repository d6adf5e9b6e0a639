cd $GRAFT_REPO_ROOT
(timeout -s KILL 120 python -m pytest tests/test_query_gpu.py tests/test_zslab_gpu.py -m gpu -x -q) 2>&1 | tail -3
timeout -s KILL 200 python bench.py --steps 4 --warmup 2 --cpu-baseline 0 --scene-b 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v for k,v in j['extras'].items() if 'renderView' in k})"
