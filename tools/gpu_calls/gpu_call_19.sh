cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d
mkdir -p $O
(time timeout -s KILL 420 python -m pytest tests/test_lab_gpu.py tests/test_example.py tests/test_rgbn_gpu.py -m gpu -q) > $O/pytest_lab.log 2>&1; tail -40 $O/pytest_lab.log
