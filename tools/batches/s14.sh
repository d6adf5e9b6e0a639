set -u
export OMP_WAIT_POLICY=passive
echo "== pytest gpu (full)"
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_s14.log 2>&1; rc=$?; echo rc=$rc
tail -8 gpurun_out/pytest_s14.log
echo "== bench default"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_s14.json 2> gpurun_out/bench_s14.err; echo rc=$?
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/bench_s14.json').read().splitlines() if l.startswith('{')][-1]
print('ms/step',d['ms_per_step'],'frac',d['roofline']['frac'],'traffic',d['roofline']['traffic'], d['config']['last_launch'])
print('fused2',d['extras']['fused2']['ms_per_frame'],'scene_b',d['extras']['scene_b']['gpu_ms_per_frame'],'host_path',{k:v for k,v in d['host_path'].items() if k!='note'})
print('cpu',d['cpu_baseline']['frames_per_s'], 'host_us', d['host_us_per_step']['total'])
PY
