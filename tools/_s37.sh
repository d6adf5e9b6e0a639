set -x
mkdir -p gpurun_out/s37
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline 0 --host-path 0 --extras 0"
for rpb in 64 32 128 256 64; do
  TSDF_HIP_ROWS_PER_BLOCK=$rpb timeout 600 $B > gpurun_out/s37/rpb${rpb}_$RANDOM.json 2> gpurun_out/s37/err.txt
done
TSDF_HIP_ZFAST=0 timeout 600 $B > gpurun_out/s37/zfast0.json 2> gpurun_out/s37/err.txt
TSDF_HIP_ROWS_PER_BLOCK=128 timeout 600 $B --color 0 > gpurun_out/s37/c0_rpb128.json 2> gpurun_out/s37/err.txt
timeout 600 $B --color 0 > gpurun_out/s37/c0_rpb64.json 2> gpurun_out/s37/err.txt
