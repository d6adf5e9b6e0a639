cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
(time timeout 900 python -m pytest tests/test_bench_dist_gpu.py -m gpu -x -q) > gpurun_out/r02b/pytest8.log 2>&1
tail -40 gpurun_out/r02b/pytest8.log
