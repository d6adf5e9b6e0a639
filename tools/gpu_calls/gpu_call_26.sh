cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g
mkdir -p $O
(time timeout -s KILL 900 python tests/evidence/long_run_parity.py --res 2048 --frames 1000 --check-every 250 --pipelined 1) > $O/long_run_2048_1000frames.json 2> $O/long_run.err; tail -c 600 $O/long_run_2048_1000frames.json; tail -3 $O/long_run.err
timeout -s KILL 300 python tools/cpp_path_timing.py 60 > $O/cpp_path_timing.json 2>> $O/long_run.err; cat $O/cpp_path_timing.json
