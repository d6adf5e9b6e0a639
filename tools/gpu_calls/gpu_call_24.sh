cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g
mkdir -p $O
timeout -s KILL 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
timeout -s KILL 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>> $O/bench_default.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
