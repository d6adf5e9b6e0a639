#!/bin/bash
# the colour update on float PAIRS (12 packed fmas per quad instead of 24, TSDF_COLOR_F2) at eight waves (spills) and seven (72 VGPRs, fits):
# A/B by alternation against the shipped eight-wave instance; then the tests the last full run did not reach
O=gpurun_out/r06_c14; mkdir -p $O
timeout 700 python tools/ab_alt.py --rounds 5 --out $O/ab_color_f2.txt --bench "--color 1" w8=lib=w8 w8f2=lib=w8f2 w7=lib=w7 w7f2=lib=w7f2 2>&1 | tail -6
timeout 300 python tools/ab_alt.py --rounds 3 --out $O/ab_color_f2_slab.txt --bench "--res 4096 --planes 512 --width 1280 --height 960 --color 1" w8=lib=w8 w7f2=lib=w7f2 2>&1 | tail -4
timeout 900 python -m pytest tests/test_fused2_gpu.py tests/test_integrate_gpu.py tests/test_implied_d_gpu.py -x -q -m gpu 2>&1 | tail -4
