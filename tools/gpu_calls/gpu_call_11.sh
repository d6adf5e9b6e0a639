cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
time bash tools/run_rocprof.sh r02 20 6 > gpurun_out/r02c/rocprof.log 2>&1
tail -5 gpurun_out/r02c/rocprof.log
