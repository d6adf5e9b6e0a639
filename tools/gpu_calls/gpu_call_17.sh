cd $GRAFT_REPO_ROOT
(time timeout 900 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -x -q --durations=5 -k config3) 2>&1 | tail -12
