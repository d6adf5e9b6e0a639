set -u
echo "== pytest gpu"
timeout 2000 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_s8.log 2>&1; rc=$?; echo rc=$rc
tail -8 gpurun_out/pytest_s8.log
if [ $rc -ne 0 ]; then exit 1; fi
echo "== smoke"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== rocprof r04 default key"
bash tools/run_rocprof.sh r04 20 6 "" > gpurun_out/run_rocprof_r04.log 2>&1; echo rc=$?
tail -3 gpurun_out/run_rocprof_r04.log
python - <<'PY'
import json
for f in ['bench_under_rocprof.json','bench_pmc_FETCH_SIZE.json']:
    try:
        d=[json.loads(l) for l in open('gpurun_out/prof_r04/'+f).read().splitlines() if l.startswith('{')][-1]
        print(f, d['roofline']['kernel_ms'], d['roofline']['frac'], d['extras'].get('fused2',{}).get('ms_per_frame') if 'extras' in d else None)
    except Exception as e: print(f,'ERR',e)
PY
