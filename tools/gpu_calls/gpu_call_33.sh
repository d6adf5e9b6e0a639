cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
(time timeout -s KILL 800 python tests/evidence/fuzz_product_vs_oracle.py --cases 120 --seed 2) > gpurun_out/r02h/fuzz_product_vs_oracle_seed2.log 2>&1
tail -5 gpurun_out/r02h/fuzz_product_vs_oracle_seed2.log; grep -c " ok$" gpurun_out/r02h/fuzz_product_vs_oracle_seed2.log; grep "DIFF" gpurun_out/r02h/fuzz_product_vs_oracle_seed2.log | head -20
