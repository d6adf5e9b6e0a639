cd $GRAFT_REPO_ROOT
timeout 300 python tools/mc_probe.py 2>&1 | tail -1
(timeout 300 python -m pytest tests/test_query_gpu.py tests/test_zslab_gpu.py -m gpu -x -q) 2>&1 | tail -3
