#!/bin/bash
# round 6, final sequence on the committed tree: profiles of five bench keys (tools/gpu_calls/r06_call_12.sh), the plain bench line,
# the whole GPU suite, and two fresh seeds of the product-vs-oracle hunt + one of the drop-in-vs-reference hunt on the new kernels
bash tools/gpu_calls/r06_call_12.sh
( time timeout 500 python bench.py > gpurun_out/r06_bench_final_tree.json 2> gpurun_out/r06_bench_final_tree.err ) 2>&1 | tail -3
mkdir -p gpurun_out/r06_c13
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r06_c13/pytest_full.txt
for seed in 61 62; do timeout 900 python tests/evidence/fuzz_product_vs_oracle.py --cases 150 --seed $seed --ref-cull 0.3 > gpurun_out/r06_c13/fuzz_product_vs_oracle_seed$seed.log 2>&1; echo "fuzz seed $seed rc=$?"; tail -2 gpurun_out/r06_c13/fuzz_product_vs_oracle_seed$seed.log; done
timeout 900 python tests/evidence/fuzz_dropin_vs_reference.py --cases 60 --seed 63 --ref-cull 0.3 > gpurun_out/r06_c13/fuzz_dropin_vs_reference_seed63.log 2>&1; echo "dropin fuzz rc=$?"; tail -2 gpurun_out/r06_c13/fuzz_dropin_vs_reference_seed63.log
# the round's gain on ONE box: the round-5 library (built from commit f76a815 into cpu_tsdf_amd/lib/variants/r05) against this tree's
timeout 500 python tools/ab_alt.py --rounds 5 --out gpurun_out/r06_c13/ab_r05_vs_r06_c1.txt --bench "--color 1" r05=lib=r05 r06= 2>&1 | tail -4
timeout 500 python tools/ab_alt.py --rounds 5 --out gpurun_out/r06_c13/ab_r05_vs_r06_c0.txt --bench "--color 0" r05=lib=r05 r06= 2>&1 | tail -4
