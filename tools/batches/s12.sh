set -u
export OMP_WAIT_POLICY=passive
echo "== fuzz product vs oracle (default path, half of the cases off-centre)"
timeout 1200 python tests/evidence/fuzz_product_vs_oracle.py --cases 120 --seed 41 --ref-cull 0.5 > gpurun_out/r04_fuzz_product_vs_oracle_seed41.log 2>&1; echo rc=$?
tail -2 gpurun_out/r04_fuzz_product_vs_oracle_seed41.log | cut -c1-300
echo "== fuzz drop-in vs compiled reference (no non-reference call, half of the cases off-centre)"
timeout 1200 python tests/evidence/fuzz_dropin_vs_reference.py --cases 80 --seed 42 --ref-cull 0.5 > gpurun_out/r04_fuzz_dropin_vs_reference_seed42.log 2>&1; echo rc=$?
tail -2 gpurun_out/r04_fuzz_dropin_vs_reference_seed42.log | cut -c1-300
