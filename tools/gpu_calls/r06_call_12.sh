#!/bin/bash
# final profiles of round 6 on the committed kernels (the sources changed after call 7: plane-loads switch, one-word marching-cubes cells):
# trace + FETCH + WRITE (+ SQ) for five bench keys through bench.py's own launches, then the plain bench line of the tree
L=gpurun_out/r06_prof.log; : > $L
timeout 800 bash tools/run_rocprof.sh r06 20 6 "" >> $L 2>&1
timeout 800 bash tools/run_rocprof.sh r06_c0 20 6 "--color 0" noextras >> $L 2>&1
timeout 400 bash tools/run_rocprof.sh r06_config4slab 20 6 "--res 4096 --planes 512 --width 1280 --height 960" lite >> $L 2>&1
timeout 400 bash tools/run_rocprof.sh r06_saturated 20 6 "--presaturate 110" lite >> $L 2>&1
timeout 400 bash tools/run_rocprof.sh r06_f32w 20 6 "--layout f32w" lite >> $L 2>&1
ls gpurun_out/prof_r06/summary_pmc_SQ2.json gpurun_out/prof_r06_c0/summary_pmc_SQ2.json gpurun_out/prof_r06_f32w/summary_pmc_WRITE_SIZE.json gpurun_out/prof_r06_saturated/summary_pmc_WRITE_SIZE.json gpurun_out/prof_r06_config4slab/summary_pmc_WRITE_SIZE.json
