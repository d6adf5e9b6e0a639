#!/bin/bash
# colourless: is the time per BLOCK?  rows per block and block order (product knobs, no rebuild); every command under its own timeout
O=gpurun_out/r05_c17; mkdir -p $O
run() { env $1 timeout 120 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 $2 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1 $2', round(d['roofline']['kernel_ms'],3), d['config']['plane_placement']['probe_sweep_ms'])
except Exception as e: print('$1 $2 failed', e)"; }
for rp in 32 64 128 256; do run TSDF_HIP_ROWS_PER_BLOCK=$rp "--color 0"; done | tee $O/summary.txt
run TSDF_HIP_ZFAST=0 "--color 0" | tee -a $O/summary.txt
for rp in 64 128; do run TSDF_HIP_ROWS_PER_BLOCK=$rp "--color 1"; done | tee -a $O/summary.txt
