#!/bin/bash
# round 5, call 2: second A/B (no count table / v_cvt_pk_u8 colour / early voxel loads), slab + colourless keys, SQ counters of the best
O=gpurun_out/r05_c3; mkdir -p $O
line() { python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[2])); f2=d.get('extras',{}).get('fused2',{})
    print(f"{sys.argv[1]:22s} kernel_ms {d['roofline']['kernel_ms']:.3f}  ms_per_step {d['ms_per_step']:.3f} fused2 {f2.get('ms_per_frame')} moved_GB {d['roofline']['bytes_moved']['per_launch']/1e9:.2f} place {d['config']['plane_placement']['probe_sweep_ms']}")
except Exception as e: print(sys.argv[1], "no result", e)
P
}
for rep in 1 2; do for n in y_pk2 y_lean y_lean_p2 y_lean_e1; do
  TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/$n/libtsdf_hip.so timeout 300 python bench.py --steps 20 --warmup 3 --extras 2 --cpu-baseline 0 --host-path 0 > $O/$n.$rep.json 2>> $O/err.log || echo "$n failed"
  line $n $O/$n.$rep.json
done; done | tee $O/summary.txt
for n in y_lean y_lean_p2; do
  TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/$n/libtsdf_hip.so timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 --color 0 > $O/$n.c0.json 2>> $O/err.log
  line "$n color=0" $O/$n.c0.json | tee -a $O/summary.txt
  TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/$n/libtsdf_hip.so timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 --res 4096 --planes 512 --width 1280 --height 960 > $O/$n.slab.json 2>> $O/err.log
  line "$n slab" $O/$n.slab.json | tee -a $O/summary.txt
done
# parity of the pk2 build on the integrate modules (it is a candidate default)
(TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/y_lean_p2/libtsdf_hip.so timeout 600 python -m pytest tests/test_integrate_gpu.py tests/test_fused2_gpu.py tests/test_implied_d_gpu.py -m gpu -x -q 2>&1 | tail -4) | tee $O/pytest_pk2.txt
# SQ counters of the candidate
export TMPDIR=/tmp; R=$(pwd); cd /tmp
for n in y_lean_p2; do
TSDF_HIP_LIB_PATH=$R/cpu_tsdf_amd/lib/variants/$n/libtsdf_hip.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS -d $R/$O/sq_$n -o pmc --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --host-path 0 > $R/$O/sq_$n.json 2> $R/$O/sq_$n.err
cd $R; python tools/pmc_reduce.py $O/sq_$n > $O/sq_$n.summary.json; find $O/sq_$n -name "*.csv" -size +1M -delete
python -c "
import json; d=json.load(open('$O/sq_$n.summary.json')); [print(k, json.dumps(v)) for k,v in d.items() if 'k_integrate<' in k]" | tee -a $O/summary.txt
done
# the new driver-run tests on the default build
(timeout 1500 python -m pytest tests/test_zslab_hip_ranks_gpu.py tests/test_evidence_gpu.py -m gpu -q -x --durations=8 2>&1 | tail -25) | tee $O/pytest_new.txt
