set -u
summ() { python - "$1" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=[json.loads(l) for l in open(f).read().strip().splitlines() if l.startswith('{')][-1]
except Exception as e:
    print(f,'ERR',e); sys.exit(0)
print(f, 'ms/step',round(d['ms_per_step'],3),'kernel_ms',round(d['roofline']['kernel_ms'],3),'frac',round(d['roofline']['frac'],3), d['config'].get('last_launch',{}).get('instance'), d['config']['plane_placement']['probe_sweep_ms'][-1])
PY
}
B="--steps 10 --warmup 3 --cpu-baseline 0 --host-path 0 --extras 0"
for r in 16 64 128; do echo "== rows_per_block $r"; TSDF_HIP_ROWS_PER_BLOCK=$r timeout 600 python bench.py $B > gpurun_out/bench_s7_rpb$r.json 2>/dev/null; summ gpurun_out/bench_s7_rpb$r.json; done
for v in ga1 ga2 ga16 w8a; do echo "== variant $v"; TSDF_HIP_LIB_PATH=$PWD/cpu_tsdf_amd/lib/variants/$v/libtsdf_hip.so timeout 600 python bench.py $B > gpurun_out/bench_s7_$v.json 2>/dev/null; summ gpurun_out/bench_s7_$v.json; done
echo "== default again"; timeout 600 python bench.py $B > gpurun_out/bench_s7_default.json 2>/dev/null; summ gpurun_out/bench_s7_default.json
