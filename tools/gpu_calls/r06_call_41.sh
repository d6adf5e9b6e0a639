#!/bin/bash
# round 6, call 41: a long hunt on the final kernels -- ten seeds of product vs oracle, four of drop-in vs compiled reference; one summary line per seed
O=gpurun_out/r06_c41; mkdir -p $O
: > $O/fuzz_summary.txt
for seed in $(seq 80 89); do
  timeout 600 python tests/evidence/fuzz_product_vs_oracle.py --cases 150 --seed $seed --ref-cull 0.3 2>&1 | grep -E "^case|cases, seed" > $O/p_$seed.log
  echo "product_vs_oracle $(tail -1 $O/p_$seed.log) ; through the pipelined loops: $(grep -c 'kp [1-9]' $O/p_$seed.log) ; paired: $(grep -c paired $O/p_$seed.log) ; DIFF lines: $(grep -c DIFF $O/p_$seed.log)" >> $O/fuzz_summary.txt
done
for seed in 90 91 92 93; do
  timeout 600 python tests/evidence/fuzz_dropin_vs_reference.py --cases 80 --seed $seed --ref-cull 0.3 2>&1 | grep -E "^case|cases, seed" > $O/d_$seed.log
  echo "dropin_vs_reference $(tail -1 $O/d_$seed.log) ; paired: $(grep -c paired $O/d_$seed.log) ; DIFF lines: $(grep -c DIFF $O/d_$seed.log)" >> $O/fuzz_summary.txt
done
cat $O/fuzz_summary.txt
