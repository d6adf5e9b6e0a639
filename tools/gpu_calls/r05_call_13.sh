#!/bin/bash
# round 5 profiles: bench.py under rocprofv3 for the four bench keys (trace + FETCH + WRITE; SQ passes for the headline)
bash tools/run_rocprof.sh r05 20 6 "" > gpurun_out/r05_prof.log 2>&1
bash tools/run_rocprof.sh r05_c0 20 6 "--color 0" lite >> gpurun_out/r05_prof.log 2>&1
bash tools/run_rocprof.sh r05_config4slab 20 6 "--res 4096 --planes 512 --width 1280 --height 960" lite >> gpurun_out/r05_prof.log 2>&1
bash tools/run_rocprof.sh r05_f32w 20 6 "--layout f32w" lite >> gpurun_out/r05_prof.log 2>&1
tail -3 gpurun_out/r05_prof.log
for t in r05 r05_c0 r05_config4slab r05_f32w; do ls gpurun_out/prof_$t | head -20; done
# the plain bench line of the committed tree (what the driver will run)
timeout 600 python bench.py > gpurun_out/r05_bench_final_tree.json 2> gpurun_out/r05_bench_final_tree.err; tail -c 600 gpurun_out/r05_bench_final_tree.json
