set -u
echo "== pytest gpu"
timeout 1700 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_s5.log 2>&1; echo rc=$?
tail -12 gpurun_out/pytest_s5.log
echo "== bench default"
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 > gpurun_out/bench_s5.json 2> gpurun_out/bench_s5.err; echo rc=$?
echo "== bench refcull"
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --extras 0 --host-path 0 --principal-offset 0.6 > gpurun_out/bench_s5_refcull.json 2> gpurun_out/bench_s5_refcull.err; echo rc=$?
echo "== bench zfast headline"
TSDF_HIP_ZFAST=1 timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --extras 0 --host-path 0 > gpurun_out/bench_s5_zfast.json 2> gpurun_out/bench_s5_zfast.err; echo rc=$?
echo "== bench slab 4096x4096x512 1280x960: zfast auto(on) / off"
for z in -1 0; do
TSDF_HIP_ZFAST=$z timeout 900 python bench.py --steps 8 --warmup 2 --cpu-baseline 0 --extras 0 --host-path 0 --res 4096 --planes 512 --width 1280 --height 960 > gpurun_out/bench_s5_slab_z$z.json 2> gpurun_out/bench_s5_slab_z$z.err; echo rc=$?
done
echo "== bench nocolor"
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --extras 0 --host-path 0 --color 0 > gpurun_out/bench_s5_nocolor.json 2> gpurun_out/bench_s5_nocolor.err; echo rc=$?
echo "== host overhead at a 256-plane slab (world 1 on RCCL)"
TSDF_BENCH_FORCE_DIST=1 MASTER_PORT=29611 timeout 600 python bench.py --steps 40 --warmup 4 --cpu-baseline 0 --extras 0 --host-path 0 --planes 256 > gpurun_out/bench_s5_slab256_rccl1.json 2> gpurun_out/bench_s5_slab256_rccl1.err; echo rc=$?
timeout 600 python bench.py --steps 40 --warmup 4 --cpu-baseline 0 --extras 0 --host-path 0 --planes 256 > gpurun_out/bench_s5_slab256.json 2> gpurun_out/bench_s5_slab256.err; echo rc=$?
echo "== dry run 8 ranks"
timeout 900 python bench.py --dry-run-ranks 8 --steps 20 --warmup 3 --cpu-baseline 0 --extras 0 --host-path 0 > gpurun_out/bench_s5_dry8.json 2> gpurun_out/bench_s5_dry8.err; echo rc=$?
echo "== scene b log2tx 4"
sb() { timeout 300 python -c "from cpu_tsdf_amd import capi; capi.use_test_library(); import bench, json; r=bench.scene_b_leg(2048, 1, 0.0); print(json.dumps({k:r.get(k) for k in ('gpu_ms_per_frame','launch','error')}))"; }
echo default; sb
echo log2tx 4; TSDF_HIP_LIVE_LOG2TX=4 sb
for f in gpurun_out/bench_s5*.json; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
except Exception as e:
    print(f,'ERR',e); sys.exit(0)
ex=d.get('extras',{})
print(f, 'ms/step',round(d['ms_per_step'],3),'kernel_ms',round(d['roofline']['kernel_ms'],3),'frac',round(d['roofline']['frac'],3), d['config'].get('last_launch',{}).get('instance'), 'host',{k:(round(v,1) if isinstance(v,float) else v) for k,v in d.get('host_us_per_step',{}).items() if k!='note'})
if ex.get('fused2'): print('   fused2', ex['fused2'].get('ms_per_frame'), ex['fused2'].get('one_sweep_per_pair'))
if ex.get('scene_b'): print('   scene_b', ex['scene_b'].get('gpu_ms_per_frame'), ex['scene_b'].get('frac_of_hbm_peak'))
PY
done
