cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
(time timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_dropin_gpu.py tests/test_wdepth_gpu.py tests/test_query_gpu.py tests/test_zslab_gpu.py tests/test_integrate_gpu.py tests/test_programs_gpu.py -m gpu -x -q) > gpurun_out/r02b/pytest7.log 2>&1
tail -40 gpurun_out/r02b/pytest7.log
