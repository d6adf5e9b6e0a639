#!/bin/bash
# A/B helper (run via gpurun from the repo root): integrate parity tests, 4 layout/colour bench lines, one SQ PMC pass.
mkdir -p gpurun_out/pmc_v9; (timeout 600 python -m pytest tests/test_integrate_gpu.py tests/test_div_gpu.py -m gpu -x -q 2>&1 | tail -8)
for L in packed f32w; do for CL in 1 0; do timeout 150 python bench.py --cpu-baseline 0 --steps 20 --extras 0 --color $CL --layout $L > gpurun_out/bench_v9_${L}_c$CL.json 2>> gpurun_out/bench.err; python -c "
import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_layout'])" gpurun_out/bench_v9_${L}_c$CL.json; done; done
export TMPDIR=/tmp; R=$(pwd); cd /tmp; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_v9/sq -o pmc --output-format csv -- python $R/tools/prof_integrate.py --steps 4 --warmup 1 --calib 1 > $R/gpurun_out/pmc_v9/prof.json 2> $R/gpurun_out/pmc_v9/prof.err; cd $R; python tools/pmc_reduce.py gpurun_out/pmc_v9/sq > gpurun_out/pmc_v9/summary.json; find gpurun_out/pmc_v9 -name "*.csv" -size +1M -delete; python -c "
import json; d=json.load(open('gpurun_out/pmc_v9/summary.json')); print(json.dumps(d.get('k_integrate')))"
