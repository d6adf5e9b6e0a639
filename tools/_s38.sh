mkdir -p gpurun_out/s38
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline 0 --host-path 0 --extras 0"
for v in nod6 next6 nod7 next7 nod6 next6 nod7 next7; do
  TSDF_HIP_LIB_PATH=$(pwd)/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so timeout 600 $B > gpurun_out/s38/${v}_$RANDOM.json 2> gpurun_out/s38/err.txt
done
for v in nod7 next7 next6; do
  TSDF_HIP_LIB_PATH=$(pwd)/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so timeout 600 $B --color 0 > gpurun_out/s38/c0_${v}.json 2> gpurun_out/s38/err.txt
done
