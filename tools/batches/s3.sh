set -u
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_s3.log 2>&1; echo rc=$?
tail -15 gpurun_out/pytest_s3.log
for v in w7 w8; do
  echo "== instance coverage + diag on $v"
  TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so timeout 600 python -m pytest tests/test_integrate_gpu.py tests/test_fused2_gpu.py -q -p no:cacheprovider > gpurun_out/pytest_s3_$v.log 2>&1; echo rc=$?
  tail -5 gpurun_out/pytest_s3_$v.log
  TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so timeout 300 python tests/evidence/diag_wave7.py 96 > gpurun_out/diag_$v.json 2> gpurun_out/diag_$v.err; echo rc=$?
done
echo "== bench default"
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 > gpurun_out/bench_s3.json 2> gpurun_out/bench_s3.err; echo rc=$?
for v in k2w4 k2w6; do
  echo "== bench $v"
  TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --scene-b 0 --host-path 0 > gpurun_out/bench_s3_$v.json 2> gpurun_out/bench_s3_$v.err; echo rc=$?
done
echo "== bench general instance (allin off)"
TSDF_HIP_ALLIN=0 timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --extras 0 --host-path 0 > gpurun_out/bench_s3_general.json 2> gpurun_out/bench_s3_general.err; echo rc=$?
echo "== bench refcull"
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --extras 0 --host-path 0 --principal-offset 0.6 > gpurun_out/bench_s3_refcull.json 2> gpurun_out/bench_s3_refcull.err; echo rc=$?
echo "== scene b sweeps"
for r in 8 16 32 64; do
  TSDF_HIP_ROWS_PER_BLOCK=$r timeout 300 python -c "import bench, json; print(json.dumps(bench.scene_b_leg(2048, 1, 0.0)))" > gpurun_out/sceneb_rpb$r.json 2> gpurun_out/sceneb_rpb$r.err; echo rpb $r rc=$?; cat gpurun_out/sceneb_rpb$r.json | cut -c1-400
done
