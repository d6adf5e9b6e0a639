cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c
mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -x -q) > $O/pytest_full.log 2>&1; tail -4 $O/pytest_full.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>> $O/bench_default.err
timeout 300 python bench.py --res 4096 --planes 512 --width 1280 --height 960 --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 > $O/bench_config4_one_slab.json 2>> $O/bench_default.err
: > $O/bench_variants.jsonl
for v in "--color 0" "--layout f32w" "--color 0 --layout f32w"; do
  timeout 300 python bench.py $v --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 >> $O/bench_variants.jsonl 2>> $O/bench_default.err
done
(time timeout 900 python tests/evidence/long_run_parity.py --res 2048 --frames 1000 --check-every 250 --pipelined 1) > $O/long_run_2048_1000frames.json 2> $O/long_run.err; tail -c 400 $O/long_run_2048_1000frames.json; tail -3 $O/long_run.err
timeout 300 python tools/cpp_path_timing.py 60 > $O/cpp_path_timing.json 2>> $O/bench_default.err; cat $O/cpp_path_timing.json
