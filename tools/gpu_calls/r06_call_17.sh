#!/bin/bash
# s_setprio around a row's load issue in k_integrate (colour: the old row loop): A/B by alternation; and the inverse (the update first)
O=gpurun_out/r06_c17; mkdir -p $O
timeout 600 python tools/ab_alt.py --rounds 5 --out $O/ab_setprio_c1.txt --bench "--color 1" prio0=lib=prio0 prio2=lib=prio2 2>&1 | tail -4
