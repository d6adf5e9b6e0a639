#!/bin/bash
O=gpurun_out/r05_c16; mkdir -p $O
export TMPDIR=/tmp; R=$(pwd); cd /tmp
for t in 1 0; do
TSDF_HIP_TILES=$t rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD -d $R/$O/sq_$t -o pmc --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --host-path 0 --color 0 > $R/$O/sq_$t.json 2> $R/$O/sq_$t.err
TSDF_HIP_TILES=$t rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY -d $R/$O/tcc_$t -o pmc --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --extras 0 --cpu-baseline 0 --host-path 0 --color 0 > $R/$O/tcc_$t.json 2> $R/$O/tcc_$t.err
done
cd $R
for t in 1 0; do for k in sq tcc; do python tools/pmc_reduce.py $O/${k}_$t > $O/${k}_$t.summary.json; find $O/${k}_$t -name "*.csv" -size +1M -delete; python -c "
import json; d=json.load(open('$O/${k}_$t.summary.json')); [print('tiles=$t', k, json.dumps({a: round(b/1e9,3) for a,b in v.items()})) for k,v in d.items() if 'k_integrate<' in k and ',false,true,true,false>' in k]"; done; done | tee $O/summary.txt
for z in 1 0; do TSDF_HIP_ZFAST=$z timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 --color 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('zfast=$z colourless', d['roofline']['kernel_ms'])" | tee -a $O/summary.txt; done
for rp in 16 64; do TSDF_HIP_ROWS_PER_BLOCK=$rp timeout 300 python bench.py --steps 20 --warmup 3 --extras 0 --cpu-baseline 0 --host-path 0 --color 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rows_per_block=$rp colourless', d['roofline']['kernel_ms'])" | tee -a $O/summary.txt; done
