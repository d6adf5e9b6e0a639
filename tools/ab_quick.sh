#!/bin/bash
# quick A/B of k_integrate at the headline size: colour and no colour, both layouts (run via gpurun)
for L in packed f32w; do for CL in 1 0; do timeout 150 python bench.py --cpu-baseline 0 --steps 30 --extras 0 --color $CL --layout $L 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'color=$CL', round(d['ms_per_step'],3), round(d['roofline']['frac_layout'],3))"; done; done
