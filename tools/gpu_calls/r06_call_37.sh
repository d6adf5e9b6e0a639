#!/bin/bash
# round 6, call 37: LLVM keeps 16 SGPRs back for a trap handler (FeatureTrapHandler: (800 / 8) - 16 -> 80 at eight waves), which is what puts
# SGPR spills (v_readlane / v_writelane, 4.5 vector cycles each) into the row loops.  A build without that reservation (94 SGPRs, half the lane
# moves) against the shipped library, alternated: does the hardware still hold eight waves, and is it faster?
O=gpurun_out/r06_c37; mkdir -p $O
timeout 700 python tools/ab_alt.py --rounds 5 --out $O/ab_notrap_c1.txt --bench "--color 1" shipped= notrap=lib=notrap 2>&1 | tail -4
timeout 500 python tools/ab_alt.py --rounds 5 --out $O/ab_notrap_c0.txt --bench "--color 0" shipped= notrap=lib=notrap 2>&1 | tail -4
