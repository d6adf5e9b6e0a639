#!/bin/bash
# round 6, call 26: the product-vs-oracle fuzz extended with the round's features (pipelined row loop knobs, block heights,
# frame pairing through the ring, marching cubes at w_min on both sides of the elision condition, max_weight 0.5)
O=gpurun_out/r06_c26; mkdir -p $O
for seed in 68 69; do timeout 1100 python tests/evidence/fuzz_product_vs_oracle.py --cases 150 --seed $seed --ref-cull 0.3 > $O/fuzz_product_vs_oracle_seed$seed.log 2>&1; echo "fuzz seed $seed rc=$?"; tail -1 $O/fuzz_product_vs_oracle_seed$seed.log; done
grep -c paired $O/fuzz_product_vs_oracle_seed68.log; grep -c "kp [1-9]" $O/fuzz_product_vs_oracle_seed68.log
