set -u
echo "== nproc $(nproc)"; rocm-smi --showmeminfo vram 2>/dev/null | head -5
echo "== diag default"
timeout 300 python tests/evidence/diag_wave7.py 96 > gpurun_out/diag_default.json 2> gpurun_out/diag_default.err; echo rc=$?
for v in w7 w7_noband w7_scratch w8; do
  echo "== diag $v"
  TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so timeout 300 python tests/evidence/diag_wave7.py 96 > gpurun_out/diag_$v.json 2> gpurun_out/diag_$v.err; echo rc=$?
done
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_s1.log 2>&1; echo rc=$?
tail -40 gpurun_out/pytest_s1.log
echo "== bench default"
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 > gpurun_out/bench_s1.json 2> gpurun_out/bench_s1.err; echo rc=$?
for v in k2w4 k2w6; do
  echo "== bench $v"
  TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --scene-b 0 --host-path 0 > gpurun_out/bench_s1_$v.json 2> gpurun_out/bench_s1_$v.err; echo rc=$?
done
