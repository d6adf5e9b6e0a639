#!/bin/bash
# round 6, call 32: the whole GPU tier and the driver's bench command on the tree with the new marching-cubes kernels
O=gpurun_out/r06_c32; mkdir -p $O
( time timeout 1100 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest_gpu_full.txt 2>&1
tail -22 $O/pytest_gpu_full.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
tail -4 $O/bench_driver_cmd.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06_c32/bench_driver_cmd.json') if l.startswith('{')][-1])
e = d['extras']
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], {k: (v.get('kernel_ms') if isinstance(v, dict) else v) for k, v in e['keys'].items()})
print(e['reconstruct_ms'], e['reconstruct_phase_ms'], e['renderView_ms'])
PY
