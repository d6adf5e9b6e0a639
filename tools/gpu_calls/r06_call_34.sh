#!/bin/bash
# round 6, call 34: more hunt seeds on the final kernels (product vs oracle with the round's features; drop-in vs compiled reference with pairing)
O=gpurun_out/r06_c34; mkdir -p $O
for seed in 71 72 73 74; do timeout 1100 python tests/evidence/fuzz_product_vs_oracle.py --cases 150 --seed $seed --ref-cull 0.3 2>&1 | grep -E "^case|cases, seed" > $O/fuzz_product_vs_oracle_seed$seed.log; tail -1 $O/fuzz_product_vs_oracle_seed$seed.log; done
for seed in 75 76; do timeout 900 python tests/evidence/fuzz_dropin_vs_reference.py --cases 80 --seed $seed --ref-cull 0.3 2>&1 | grep -E "^case|cases, seed" > $O/fuzz_dropin_vs_reference_seed${seed}_pairing.log; tail -1 $O/fuzz_dropin_vs_reference_seed${seed}_pairing.log; done
grep -h -c "kp [1-9]" $O/fuzz_product_vs_oracle_seed7*.log
