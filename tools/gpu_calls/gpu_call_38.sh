cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j
mkdir -p $O
(time timeout -s KILL 600 bash tools/run_rocprof.sh r02 20 6) > $O/rocprof.log 2>&1; tail -2 $O/rocprof.log
(time timeout -s KILL 900 python -m pytest tests -m gpu -x -q) 2>&1 | tail -5
