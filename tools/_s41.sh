mkdir -p gpurun_out/s41
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/s41/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s41/pytest.log
tail -4 gpurun_out/s41/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s41/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/s41/smoke.log
tail -2 gpurun_out/s41/smoke.log
bash tools/run_rocprof.sh r04 20 6 "" > gpurun_out/s41/prof_r04.log 2>&1
bash tools/run_rocprof.sh r04_c0 20 6 "--color 0" lite > gpurun_out/s41/prof_c0.log 2>&1
bash tools/run_rocprof.sh r04_config4slab 20 6 "--res 4096 --planes 512 --width 1280 --height 960" lite > gpurun_out/s41/prof_slab.log 2>&1
bash tools/run_rocprof.sh r04_f32w 20 6 "--layout f32w" lite > gpurun_out/s41/prof_f32w.log 2>&1
timeout 400 python tests/evidence/fuzz_product_vs_oracle.py --cases 150 --seed 47 --ref-cull 0.3 > gpurun_out/s41/fuzz_product_vs_oracle_seed47.log 2>&1; echo "rc=$?" >> gpurun_out/s41/fuzz_product_vs_oracle_seed47.log
tail -2 gpurun_out/s41/fuzz_product_vs_oracle_seed47.log
timeout 400 python tests/evidence/fuzz_dropin_vs_reference.py --cases 60 --seed 48 --ref-cull 0.3 > gpurun_out/s41/fuzz_dropin_vs_reference_seed48.log 2>&1; echo "rc=$?" >> gpurun_out/s41/fuzz_dropin_vs_reference_seed48.log
tail -2 gpurun_out/s41/fuzz_dropin_vs_reference_seed48.log
