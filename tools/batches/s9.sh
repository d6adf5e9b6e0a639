set -u
bash tools/run_rocprof.sh r04_c0 20 6 "--color 0" lite > gpurun_out/run_rocprof_r04_c0.log 2>&1; echo c0 rc=$?
bash tools/run_rocprof.sh r04_f32w 20 6 "--layout f32w" lite > gpurun_out/run_rocprof_r04_f32w.log 2>&1; echo f32w rc=$?
bash tools/run_rocprof.sh r04_config4slab 12 4 "--res 4096 --planes 512 --width 1280 --height 960" lite > gpurun_out/run_rocprof_r04_slab.log 2>&1; echo slab rc=$?
echo "== refcull + scene b trace"
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_r04_refcull -o bench --output-format csv -- python $ROOT/bench.py --steps 12 --warmup 2 --cpu-baseline 0 --host-path 0 --extras 0 --principal-offset 0.6 > $ROOT/gpurun_out/prof_r04_refcull/bench_under_rocprof.json 2> $ROOT/gpurun_out/prof_r04_refcull/bench.err; echo rc=$?
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_r04_sceneb -o sceneb --output-format csv -- python -c "import sys; sys.path.insert(0, '$ROOT'); import bench, json; print(json.dumps(bench.scene_b_leg(2048, 1, 0.0)))" > $ROOT/gpurun_out/prof_r04_sceneb/scene_b.json 2> $ROOT/gpurun_out/prof_r04_sceneb/err.log; echo rc=$?
cd $ROOT
for t in refcull sceneb; do find gpurun_out/prof_r04_$t -name "*_kernel_stats.csv" -exec cp {} gpurun_out/prof_r04_$t/kernel_stats.csv \; ; find gpurun_out/prof_r04_$t -name "*.csv" -size +2M -delete; done
tail -2 gpurun_out/prof_r04_sceneb/scene_b.json | cut -c1-300
head -5 gpurun_out/prof_r04_sceneb/kernel_stats.csv | cut -c1-200
