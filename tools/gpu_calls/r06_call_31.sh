#!/bin/bash
# round 6, call 31: the headline key's profile once more (trace + FETCH + WRITE + SQ through bench.py's own launches): the
# marching-cubes kernels in the trace changed (emit by vertex, k_mc_need_rows, no k_mc_counts / k_mc_expand_rgb); k_integrate did not
L=gpurun_out/r06_prof31.log; : > $L
rm -rf gpurun_out/prof_r06
timeout 800 bash tools/run_rocprof.sh r06 20 6 "" >> $L 2>&1
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/prof_r06/bench_under_rocprof.json') if l.startswith('{')][-1])
print('kernel_ms under trace', d['roofline']['kernel_ms'], 'reconstruct', d['extras'].get('reconstruct_phase_ms'))
PY
grep -E "k_mc|k_integrate|k_raycast" gpurun_out/prof_r06/kernel_stats.csv | cut -c1-160 | head -12
