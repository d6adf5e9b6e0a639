cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
bash tools/run_rocprof.sh r02 20 6 > gpurun_out/r02c/rocprof2.log 2>&1
tail -3 gpurun_out/r02c/rocprof2.log
