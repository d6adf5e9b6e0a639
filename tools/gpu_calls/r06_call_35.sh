#!/bin/bash
# round 6, call 35: does v_pk_fma_f32 cost one issue slot or two on this chip?  (what DESIGN 3.1's reading of the packed colour update needs)
mkdir -p gpurun_out/r06_c35
for i in 1 2 3; do ./tools/ubench/pk_fma_rate; done | tee gpurun_out/r06_c35/pk_fma_rate.txt
