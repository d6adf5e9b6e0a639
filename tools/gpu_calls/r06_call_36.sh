#!/bin/bash
# round 6, call 36: the price list -- SIMD cycles per wave64 instruction for the opcodes k_integrate's row loop is made of
mkdir -p gpurun_out/r06_c36
timeout 60 ./tools/ubench/valu_cost | tee gpurun_out/r06_c36/valu_cost.txt
