#!/bin/bash
# k_integrate_pc (the headline instance on the two-stage pipeline): parity, then A/B by alternation against k_integrate's own
# row loop (TSDF_HIP_PIPE=1: colour on the old kernel) and the four-wave register budget (variant pc4); rows per block 64 (default)
O=gpurun_out/r06_c04; mkdir -p $O
timeout 900 python -m pytest tests/test_integrate_gpu.py tests/test_implied_d_gpu.py tests/test_fused2_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
timeout 600 python tools/ab_alt.py --rounds 5 --out $O/ab_pipec_c1.txt --bench "--color 1" old=TSDF_HIP_PIPE=1 pipec5=TSDF_HIP_PIPE=3 pipec4=lib=pc4 2>&1 | tail -6
timeout 300 python tools/ab_alt.py --rounds 3 --out $O/ab_slab.txt --bench "--res 4096 --planes 512 --width 1280 --height 960 --color 1" old=TSDF_HIP_PIPE=1 pipec5=TSDF_HIP_PIPE=3 2>&1 | tail -5
timeout 200 python tools/ab_alt.py --rounds 3 --out $O/ab_c0_rows.txt --bench "--color 0" rows64= rows32=TSDF_HIP_ROWS_PER_BLOCK=32 2>&1 | tail -5
