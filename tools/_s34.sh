set -x
mkdir -p gpurun_out/s34
bash tools/run_rocprof.sh r04 20 6 "" > gpurun_out/s34/prof_r04.log 2>&1
bash tools/run_rocprof.sh r04_c0 20 6 "--color 0" lite > gpurun_out/s34/prof_c0.log 2>&1
bash tools/run_rocprof.sh r04_config4slab 20 6 "--res 4096 --planes 512 --width 1280 --height 960" lite > gpurun_out/s34/prof_slab.log 2>&1
bash tools/run_rocprof.sh r04_f32w 20 6 "--layout f32w" lite > gpurun_out/s34/prof_f32w.log 2>&1
timeout 400 python tests/evidence/fuzz_product_vs_oracle.py --cases 150 --seed 46 --ref-cull 0.3 > gpurun_out/s34/fuzz_product_vs_oracle_seed46.log 2>&1; echo "rc=$?" >> gpurun_out/s34/fuzz_product_vs_oracle_seed46.log
tail -3 gpurun_out/s34/fuzz_product_vs_oracle_seed46.log
ls gpurun_out/prof_r04 gpurun_out/prof_r04_c0
