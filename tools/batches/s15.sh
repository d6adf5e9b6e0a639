set -u
bash tools/run_rocprof.sh r04 20 6 "" > gpurun_out/run_rocprof_r04.log 2>&1; echo default rc=$?
bash tools/run_rocprof.sh r04_c0 20 6 "--color 0" lite > gpurun_out/run_rocprof_r04_c0.log 2>&1; echo c0 rc=$?
bash tools/run_rocprof.sh r04_f32w 20 6 "--layout f32w" lite > gpurun_out/run_rocprof_r04_f32w.log 2>&1; echo f32w rc=$?
bash tools/run_rocprof.sh r04_config4slab 12 4 "--res 4096 --planes 512 --width 1280 --height 960" lite > gpurun_out/run_rocprof_r04_slab.log 2>&1; echo slab rc=$?
