#!/bin/bash
# does the voxel STORES' acknowledgement gate the rows?  cache-policy variants of the stream stores (tsdf_buffer.h), timing only for `nostore`
O=gpurun_out/r05_c18; mkdir -p $O
run() { TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/$1/libtsdf_hip.so timeout 120 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 $2 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1 $2', round(d['roofline']['kernel_ms'],3), d['config']['plane_placement']['probe_sweep_ms'][-1])
except Exception as e: print('$1 $2 failed', e)"; }
for n in r4x st0 st1 st16 st17 st18 st3 ld0st0 nostore; do
  if [ $n = r4x ]; then
    timeout 120 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 --color 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shipped --color 0', round(d['roofline']['kernel_ms'],3))"
    timeout 120 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shipped --color 1', round(d['roofline']['kernel_ms'],3))"
  else
    run $n "--color 0"; run $n "--color 1"
  fi
done | tee $O/summary.txt
