cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f
mkdir -p $O
(time timeout -s KILL 900 bash tools/run_rocprof.sh r02 20 6) > $O/rocprof.log 2>&1; tail -3 $O/rocprof.log
(time timeout -s KILL 900 python -m pytest tests -m gpu -x -q --durations=8) > $O/pytest_full.log 2>&1; tail -14 $O/pytest_full.log
timeout -s KILL 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 700 $O/bench_default.json; echo
timeout -s KILL 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>> $O/bench_default.err
timeout -s KILL 300 python bench.py --res 4096 --planes 512 --width 1280 --height 960 --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 > $O/bench_config4_one_slab.json 2>> $O/bench_default.err
: > $O/bench_variants.jsonl
for v in "--color 0" "--layout f32w" "--color 0 --layout f32w"; do
  timeout -s KILL 300 python bench.py $v --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 >> $O/bench_variants.jsonl 2>> $O/bench_default.err
done
tail -3 $O/bench_default.err
