#!/bin/bash
O=gpurun_out/r06_c18; mkdir -p $O
timeout 600 python tools/ab_alt.py --rounds 5 --out $O/ab_setprio_late_c1.txt --bench "--color 1" prio0=lib=prio0 update_first=lib=priom2 2>&1 | tail -4
