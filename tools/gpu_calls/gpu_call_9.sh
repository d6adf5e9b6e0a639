cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
(time timeout 1200 python -m pytest tests -m gpu -x -q --durations=8) > gpurun_out/r02b/pytest9.log 2>&1
tail -25 gpurun_out/r02b/pytest9.log
timeout 300 python tools/cpp_path_timing.py 60 > gpurun_out/r02b/cpp_path_timing.json 2> gpurun_out/r02b/cpp_path_timing.err
cat gpurun_out/r02b/cpp_path_timing.json; tail -3 gpurun_out/r02b/cpp_path_timing.err
