set -u
summ() { python - "$1" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
except Exception as e:
    print(f,'ERR',e); sys.exit(0)
ex=d.get('extras',{})
print(f, 'ms/step',round(d['ms_per_step'],3),'kernel_ms',round(d['roofline']['kernel_ms'],3),'frac',round(d['roofline']['frac'],3), d['config'].get('last_launch',{}).get('instance'), d['config']['plane_placement']['probe_sweep_ms'][-1], 'host',round(d.get('host_us_per_step',{}).get('total',0),1))
if ex.get('fused2'): print('   fused2', ex['fused2'].get('ms_per_frame'), ex['fused2'].get('one_sweep_per_pair'))
if ex.get('scene_b'): print('   scene_b', ex['scene_b'].get('gpu_ms_per_frame'), ex['scene_b'].get('frac_of_hbm_peak'))
PY
}
B="--steps 10 --warmup 3 --cpu-baseline 0 --host-path 0"
echo "== default (zfast on)"; timeout 600 python bench.py $B > gpurun_out/bench_s6.json 2> gpurun_out/bench_s6.err; summ gpurun_out/bench_s6.json
echo "== early loads"; TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/early/libtsdf_hip.so timeout 600 python bench.py $B --extras 0 > gpurun_out/bench_s6_early.json 2> gpurun_out/bench_s6_early.err; summ gpurun_out/bench_s6_early.json
echo "== nocolor default / early"
timeout 600 python bench.py $B --extras 0 --color 0 > gpurun_out/bench_s6_nc.json 2> gpurun_out/bench_s6_nc.err; summ gpurun_out/bench_s6_nc.json
TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/early/libtsdf_hip.so timeout 600 python bench.py $B --extras 0 --color 0 > gpurun_out/bench_s6_nc_early.json 2> gpurun_out/bench_s6_nc_early.err; summ gpurun_out/bench_s6_nc_early.json
echo "== refcull (zfast on)"; timeout 600 python bench.py $B --extras 0 --principal-offset 0.6 > gpurun_out/bench_s6_refcull.json 2> gpurun_out/bench_s6_refcull.err; summ gpurun_out/bench_s6_refcull.json
echo "== rccl world 1, 256 planes"
TSDF_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout 600 python bench.py --steps 40 --warmup 4 --cpu-baseline 0 --extras 0 --host-path 0 --planes 256 > gpurun_out/bench_s6_slab256_rccl1.json 2> gpurun_out/bench_s6_slab256_rccl1.err; echo rc=$?; summ gpurun_out/bench_s6_slab256_rccl1.json; tail -5 gpurun_out/bench_s6_slab256_rccl1.err
echo "== pytest subset (integrate, fused, multi) with zfast default on"
timeout 900 python -m pytest tests/test_integrate_gpu.py tests/test_fused2_gpu.py tests/test_multi_gpu.py tests/test_zslab_gpu.py -q -p no:cacheprovider 2>&1 | tail -4
