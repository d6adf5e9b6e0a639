#!/bin/bash
# final sequence of round 5 on the eight-wave colour instance: profiles (four keys), the plain bench line, then the whole GPU suite
timeout 700 bash tools/run_rocprof.sh r05 20 6 "" > gpurun_out/r05_prof2.log 2>&1
timeout 300 bash tools/run_rocprof.sh r05_c0 20 6 "--color 0" lite >> gpurun_out/r05_prof2.log 2>&1
timeout 300 bash tools/run_rocprof.sh r05_config4slab 20 6 "--res 4096 --planes 512 --width 1280 --height 960" lite >> gpurun_out/r05_prof2.log 2>&1
timeout 300 bash tools/run_rocprof.sh r05_f32w 20 6 "--layout f32w" lite >> gpurun_out/r05_prof2.log 2>&1
ls gpurun_out/prof_r05/summary_pmc_SQ2.json gpurun_out/prof_r05_f32w/summary_pmc_WRITE_SIZE.json
timeout 500 python bench.py > gpurun_out/r05_bench_final_tree3.json 2> gpurun_out/r05_bench_final_tree3.err; tail -c 200 gpurun_out/r05_bench_final_tree3.json
mkdir -p gpurun_out/r05_c20
(timeout 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) | tee gpurun_out/r05_c20/pytest_full.txt
