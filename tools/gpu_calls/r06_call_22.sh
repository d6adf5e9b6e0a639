#!/bin/bash
# round 6, call 22: the driver's own bench command on the final tree, wall clock included
mkdir -p gpurun_out/r06_c22
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_c22/bench_driver_cmd.json 2> gpurun_out/r06_c22/bench_driver_cmd.err
tail -5 gpurun_out/r06_c22/bench_driver_cmd.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06_c22/bench_driver_cmd.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], {k: (v.get('kernel_ms') if isinstance(v, dict) else v) for k, v in d['extras']['keys'].items()})
print(d['host_path'])
PY
