cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_div_gpu.py tests/test_integrate_gpu.py tests/test_fullsize_gpu.py tests/test_wdepth_gpu.py -m gpu -x -q) 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --extras 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms', j['roofline']['kernel_ms'], 'frac', j['roofline']['frac'])"; done
