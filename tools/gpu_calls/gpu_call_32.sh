cd $GRAFT_REPO_ROOT
(time timeout -s KILL 900 python -m pytest tests -m gpu -x -q) 2>&1 | tail -6
timeout -s KILL 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -s KILL 300 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'])"
