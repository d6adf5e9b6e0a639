mkdir -p gpurun_out/s40
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline 0 --host-path 0 --extras 0"
E=$(pwd)/cpu_tsdf_amd/lib/variants/early/libtsdf_hip.so
L=$(pwd)/cpu_tsdf_amd/lib/variants/late/libtsdf_hip.so
for i in 1 2 3; do
  TSDF_HIP_LIB_PATH=$L timeout 600 $B > gpurun_out/s40/late_default_$i.json 2> gpurun_out/s40/err.txt
  TSDF_HIP_LIB_PATH=$E timeout 600 $B > gpurun_out/s40/early_default_$i.json 2> gpurun_out/s40/err.txt
done
for k in "c0 --color 0" "slab --res 4096 --planes 512 --width 1280 --height 960" "f32w --layout f32w"; do
  set -- $k; n=$1; shift
  TSDF_HIP_LIB_PATH=$L timeout 600 $B $@ > gpurun_out/s40/late_$n.json 2> gpurun_out/s40/err.txt
  TSDF_HIP_LIB_PATH=$E timeout 600 $B $@ > gpurun_out/s40/early_$n.json 2> gpurun_out/s40/err.txt
done
TSDF_HIP_LIB_PATH=$L timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline 0 --host-path 0 > gpurun_out/s40/late_full.json 2> gpurun_out/s40/err.txt
TSDF_HIP_LIB_PATH=$E timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline 0 --host-path 0 > gpurun_out/s40/early_full.json 2> gpurun_out/s40/err.txt
