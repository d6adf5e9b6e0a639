#!/bin/bash
# profiles of the committed kernels once more (the integrate source gained a switch, off: same code, new hash) + the plain bench line
bash tools/gpu_calls/r06_call_12.sh
( time timeout 500 python bench.py > gpurun_out/r06_bench_final_tree.json 2> gpurun_out/r06_bench_final_tree.err ) 2>&1 | tail -3
