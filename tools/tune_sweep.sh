mkdir -p gpurun_out
run() { echo "$1" >> gpurun_out/t5_bench.log; env $1 timeout 300 python bench.py --steps 40 --warmup 4 --cpu-baseline 0 --color $2 2>&1 | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['frac'], j['value'])" >> gpurun_out/t5_bench.log; }
for C in 1 0; do
 echo "== COLOR $C" >> gpurun_out/t5_bench.log
 for rep in 1 2; do
  run "TSDF_HIP_ROWS_PER_BLOCK=8" $C
  run "TSDF_HIP_ROWS_PER_BLOCK=16" $C
  run "TSDF_HIP_ROWS_PER_BLOCK=32" $C
  run "TSDF_HIP_ROWS_PER_BLOCK=64" $C
  run "TSDF_HIP_ROWS_PER_BLOCK=32 TSDF_HIP_NONTEMPORAL=1" $C
  run "TSDF_HIP_ROWS_PER_BLOCK=32 TSDF_HIP_FAST_PROJECTION=0" $C
  run "TSDF_HIP_ROWS_PER_BLOCK=32 TSDF_HIP_SKIP_UNCHANGED=0" $C
 done
done
cat gpurun_out/t5_bench.log
