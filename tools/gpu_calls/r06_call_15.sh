#!/bin/bash
# round 6, final sequence on the committed tree: profiles of five bench keys (tools/gpu_calls/r06_call_12.sh), the plain bench line,
# then the whole GPU suite
bash tools/gpu_calls/r06_call_12.sh
( time timeout 500 python bench.py > gpurun_out/r06_bench_final_tree.json 2> gpurun_out/r06_bench_final_tree.err ) 2>&1 | tail -3
mkdir -p gpurun_out/r06_c15
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r06_c15/pytest_full.txt
