#!/bin/bash
# round 6, call 21: the whole GPU tier on the final tree, with the slowest tests listed
mkdir -p gpurun_out/r06_c21
( time timeout 1100 python -m pytest tests -m gpu -x -q --durations=30 ) > gpurun_out/r06_c21/pytest_gpu_full.txt 2>&1
tail -45 gpurun_out/r06_c21/pytest_gpu_full.txt
