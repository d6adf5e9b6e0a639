set -x
mkdir -p gpurun_out/s35
timeout 900 python -m pytest tests/test_implied_d_gpu.py tests/test_integrate_gpu.py tests/test_fused2_gpu.py -q -x -p no:cacheprovider > gpurun_out/s35/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s35/pytest.log
tail -4 gpurun_out/s35/pytest.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline 0 --host-path 0 --extras 0"
for on in 1 0 1 0; do
  TSDF_HIP_IMPLIED_D=$on timeout 600 $B > gpurun_out/s35/bench_default_${on}_$RANDOM.json 2> gpurun_out/s35/err.txt
done
for on in 1 0; do
  TSDF_HIP_IMPLIED_D=$on timeout 600 $B --color 0 > gpurun_out/s35/bench_c0_$on.json 2> gpurun_out/s35/err.txt
  TSDF_HIP_IMPLIED_D=$on timeout 600 $B --res 4096 --planes 512 --width 1280 --height 960 > gpurun_out/s35/bench_slab_$on.json 2> gpurun_out/s35/err.txt
done
