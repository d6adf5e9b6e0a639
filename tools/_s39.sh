mkdir -p gpurun_out/s39
timeout 900 python -m pytest tests/test_implied_d_gpu.py tests/test_integrate_gpu.py tests/test_fused2_gpu.py tests/test_product_lib_gpu.py -q -x -p no:cacheprovider > gpurun_out/s39/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s39/pytest.log
tail -3 gpurun_out/s39/pytest.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline 0 --host-path 0"
timeout 600 $B > gpurun_out/s39/bench_default.json 2> gpurun_out/s39/err.txt
TSDF_HIP_IMPLIED_D=0 timeout 600 $B --extras 0 > gpurun_out/s39/bench_default_noimplied.json 2> gpurun_out/s39/err.txt
timeout 600 $B --extras 0 --color 0 > gpurun_out/s39/bench_c0.json 2> gpurun_out/s39/err.txt
timeout 600 $B --extras 0 --res 4096 --planes 512 --width 1280 --height 960 > gpurun_out/s39/bench_slab.json 2> gpurun_out/s39/err.txt
timeout 600 $B --extras 0 --layout f32w > gpurun_out/s39/bench_f32w.json 2> gpurun_out/s39/err.txt
