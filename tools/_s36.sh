set -x
mkdir -p gpurun_out/s36
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline 0 --host-path 0 --extras 0"
V=$(pwd)/cpu_tsdf_amd/lib/variants/early/libtsdf_hip.so
for i in 1 2; do
  timeout 600 $B > gpurun_out/s36/base_default_$i.json 2> gpurun_out/s36/err.txt
  TSDF_HIP_LIB_PATH=$V timeout 600 $B > gpurun_out/s36/early_default_$i.json 2> gpurun_out/s36/err.txt
done
timeout 600 $B --color 0 > gpurun_out/s36/base_c0.json 2> gpurun_out/s36/err.txt
TSDF_HIP_LIB_PATH=$V timeout 600 $B --color 0 > gpurun_out/s36/early_c0.json 2> gpurun_out/s36/err.txt
timeout 600 $B --res 4096 --planes 512 --width 1280 --height 960 > gpurun_out/s36/base_slab.json 2> gpurun_out/s36/err.txt
TSDF_HIP_LIB_PATH=$V timeout 600 $B --res 4096 --planes 512 --width 1280 --height 960 > gpurun_out/s36/early_slab.json 2> gpurun_out/s36/err.txt
TSDF_HIP_LIB_PATH=$V timeout 600 python -m pytest tests/test_implied_d_gpu.py tests/test_integrate_gpu.py -q -x -p no:cacheprovider > gpurun_out/s36/pytest_early.log 2>&1
tail -3 gpurun_out/s36/pytest_early.log
