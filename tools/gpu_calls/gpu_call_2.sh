set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
(time timeout 300 python -m pytest tests/test_wdepth_gpu.py tests/test_zslab_gpu.py -m gpu -x -q) > gpurun_out/r02a/pytest2.log 2>&1
tail -5 gpurun_out/r02a/pytest2.log
time bash tools/run_rocprof.sh r02a 20 6 > gpurun_out/r02a/rocprof.log 2>&1
tail -30 gpurun_out/r02a/rocprof.log
