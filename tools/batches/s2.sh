set -u
echo "== diag default"
timeout 300 python tests/evidence/diag_wave7.py 96 > gpurun_out/diag_default.json 2> gpurun_out/diag_default.err; echo rc=$?
for v in w7 w7_noband w7_scratch w8; do
  echo "== diag $v"
  TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so timeout 300 python tests/evidence/diag_wave7.py 96 > gpurun_out/diag_$v.json 2> gpurun_out/diag_$v.err; echo rc=$?
done
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_s2.log 2>&1; echo rc=$?
tail -30 gpurun_out/pytest_s2.log
echo "== bench default"
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 > gpurun_out/bench_s2.json 2> gpurun_out/bench_s2.err; echo rc=$?
for v in k2w4 k2w6; do
  echo "== bench $v"
  TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --scene-b 0 --host-path 0 > gpurun_out/bench_s2_$v.json 2> gpurun_out/bench_s2_$v.err; echo rc=$?
done
echo "== rocprof refcull (principal offset 0.6) and scene b"
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_s2_refcull -o bench --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --cpu-baseline 0 --host-path 0 --scene-b 1 --principal-offset 0.6 > $ROOT/gpurun_out/bench_s2_refcull.json 2> $ROOT/gpurun_out/bench_s2_refcull.err; echo rc=$?
cd $ROOT
find gpurun_out/prof_s2_refcull -name "*_kernel_stats.csv" -exec cp {} gpurun_out/s2_refcull_kernel_stats.csv \;
find gpurun_out/prof_s2_refcull -name "*.csv" -size +2M -delete
cat gpurun_out/s2_refcull_kernel_stats.csv | cut -c1-60,200-400 | head -30
