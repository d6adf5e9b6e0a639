#!/bin/bash
# why is the pipelined COLOUR kernel no faster?  its SQ counters (VALU count, VALU-active, waves) beside k_integrate's (profiles/r06_summary_pmc_SQ*.json)
O=gpurun_out/r06_c09; mkdir -p $O
export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp
BENCH="python $ROOT/bench.py --warmup 2 --cpu-baseline 0 --scene-b 0 --host-path 0 --keys 0 --extras 0 --steps 6"
TSDF_HIP_PIPE=3 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $ROOT/$O/pmc_SQ -o pmc --output-format csv -- $BENCH > $ROOT/$O/bench_SQ.json 2> $ROOT/$O/bench_SQ.err
TSDF_HIP_PIPE=3 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD TCC_HIT_sum TCC_MISS_sum -d $ROOT/$O/pmc_SQ2 -o pmc --output-format csv -- $BENCH > $ROOT/$O/bench_SQ2.json 2> $ROOT/$O/bench_SQ2.err
cd $ROOT
for d in pmc_SQ pmc_SQ2; do python tools/pmc_reduce.py $O/$d > $O/summary_$d.json; done
find $O -name "*.csv" -size +1M -delete
python - <<'P'
import json
a=json.load(open('gpurun_out/r06_c09/summary_pmc_SQ.json')); b=json.load(open('gpurun_out/r06_c09/summary_pmc_SQ2.json'))
for k in a:
    if k.startswith('k_integrate_pc<') or k.startswith('k_integrate<'):
        x=a[k]; y=b.get(k,{})
        simd=x['GRBM_GUI_ACTIVE']/8*1024
        print(k, 'valu_active', round(x['SQ_ACTIVE_INST_VALU']*4/simd,3), 'waves', round(x['SQ_WAVE_CYCLES']*4/simd,2), 'VALU G', round(y.get('SQ_INSTS_VALU',0)/1e9,3), 'SALU G', round(y.get('SQ_INSTS_SALU',0)/1e9,3), 'VMEM_RD M', round(y.get('SQ_INSTS_VMEM_RD',0)/1e6), 'ms', round(x['GRBM_GUI_ACTIVE']/8/2.4e6,2), 'disp', x['dispatches'])
P
