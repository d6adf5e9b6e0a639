#!/bin/bash
# stage A of the pipelined kernels requests the plane words BEFORE the projection: A/B by alternation, colourless and colour (pipe 3), against the old order
O=gpurun_out/r06_c10; mkdir -p $O
timeout 300 python -m pytest tests/test_integrate_gpu.py -x -q -m gpu -k "pipelined or every_reachable" 2>&1 | tail -3
timeout 400 python tools/ab_alt.py --rounds 5 --out $O/ab_planefirst_c0.txt --bench "--color 0" first= last=lib=planelast 2>&1 | tail -4
timeout 500 python tools/ab_alt.py --rounds 5 --out $O/ab_planefirst_c1.txt --bench "--color 1" kint=TSDF_HIP_PIPE=1 pc_first=TSDF_HIP_PIPE=3 pc_last=TSDF_HIP_PIPE=3,lib=planelast 2>&1 | tail -5
