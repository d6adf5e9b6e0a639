#!/bin/bash
# block order re-checked on the faster kernels: planes-fastest (zfast, the default since round 4) against x-chunks-fastest, colour and not
O=gpurun_out/r06_c16; mkdir -p $O
timeout 500 python tools/ab_alt.py --rounds 5 --out $O/ab_zfast_c1.txt --bench "--color 1" zfast1= zfast0=TSDF_HIP_ZFAST=0 2>&1 | tail -4
timeout 500 python tools/ab_alt.py --rounds 5 --out $O/ab_zfast_c0.txt --bench "--color 0" zfast1= zfast0=TSDF_HIP_ZFAST=0 2>&1 | tail -4
