#!/bin/bash
# per-wave "band seen" flag sets (no barrier at the block's end) in k_integrate / k_integrate_p / k_integrate2: parity, then
# A/B by alternation against the block-wide set (variant bandblk), colour and not; rows per block with the pipelined kernel
O=gpurun_out/r06_c03; mkdir -p $O
timeout 900 python -m pytest tests/test_integrate_gpu.py tests/test_implied_d_gpu.py tests/test_fused2_gpu.py tests/test_query_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.txt
timeout 600 python tools/ab_alt.py --rounds 5 --out $O/ab_band_c0.txt --bench "--color 0" perwave= blockwide=lib=bandblk rows64=TSDF_HIP_ROWS_PER_BLOCK=64 rows128=TSDF_HIP_ROWS_PER_BLOCK=128 2>&1 | tail -7
timeout 600 python tools/ab_alt.py --rounds 5 --out $O/ab_band_c1.txt --bench "--color 1" perwave= blockwide=lib=bandblk rows64=TSDF_HIP_ROWS_PER_BLOCK=64 2>&1 | tail -6
