set -u
echo "== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_s4.log 2>&1; echo rc=$?
tail -12 gpurun_out/pytest_s4.log
echo "== w7: integrate + fused modules"
TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/w7/libtsdf_hip.so timeout 600 python -m pytest tests/test_integrate_gpu.py tests/test_fused2_gpu.py -q -p no:cacheprovider > gpurun_out/pytest_s4_w7.log 2>&1; echo rc=$?
tail -4 gpurun_out/pytest_s4_w7.log
echo "== scene b sweeps"
sb() { timeout 300 python -c "import bench, json; r=bench.scene_b_leg(2048, 1, 0.0); print(json.dumps({k:r.get(k) for k in ('gpu_ms_per_frame','launch','error')}))"; }
echo default; sb
for t in 5 7 8; do echo log2tx $t; TSDF_HIP_LIVE_LOG2TX=$t sb; done
echo w7; TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/w7/libtsdf_hip.so sb
echo w7 log2tx5; TSDF_HIP_LIVE_LOG2TX=5 TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/w7/libtsdf_hip.so sb
echo "== bench general instance w7"
TSDF_HIP_ALLIN=0 TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/w7/libtsdf_hip.so timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --extras 0 --host-path 0 > gpurun_out/bench_s4_general_w7.json 2> gpurun_out/bench_s4_general_w7.err; echo rc=$?
python -c "import json; d=json.loads(open('gpurun_out/bench_s4_general_w7.json').read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['config']['last_launch'])"
echo "== rocprof refcull + scene b"
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_s4_refcull -o bench --output-format csv -- python $ROOT/bench.py --steps 8 --warmup 2 --cpu-baseline 0 --host-path 0 --scene-b 1 --principal-offset 0.6 > $ROOT/gpurun_out/bench_s4_refcull.json 2> $ROOT/gpurun_out/bench_s4_refcull.err; echo rc=$?
cd $ROOT
find gpurun_out/prof_s4_refcull -name "*_kernel_stats.csv" -exec cp {} gpurun_out/s4_refcull_kernel_stats.csv \;
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_s4_refcull/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if n.startswith(('void k_integrate', 'void k_rows', 'k_cull')):
        by[(n[:44] + n[n.find('>('):n.find('>(')+1] if False else n.split('(')[0][:70], r.get('Grid_Size_X', r.get('Grid_Size')), )].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(by.items()):
    v.sort()
    print(k, 'n', len(v), 'min', v[0] / 1e3, 'med', v[len(v) // 2] / 1e3, 'max', v[-1] / 1e3, 'us')
PY
find gpurun_out/prof_s4_refcull -name "*.csv" -size +2M -delete
