#!/bin/bash
# round 6, call 23: where k_mc_classify's 3.3 ms go with the band skip on -- the probe builds of tsdf_march.hip
# (MC_PROBE 1: every wave on the quiet path = loads + ballot + barrier; 2: and no barrier; 3: masks built, nothing listed)
O=gpurun_out/r06_c40; mkdir -p $O
for v in shipped mcp3 mcp4 mcp5; do
  for i in 1 2 3; do
    if [ $v = shipped ]; then unset TSDF_HIP_LIB_PATH; else export TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so; fi
    timeout 200 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --host-path 0 --scene-b 0 --keys 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extras']
print('$v', e.get('reconstruct_phase_ms'), e.get('reconstruct_active_cells'), e.get('reconstruct_ms'))"
  done
done | tee $O/mc_probes.txt
