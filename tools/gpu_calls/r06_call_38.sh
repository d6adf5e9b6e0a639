#!/bin/bash
# round 6, call 38: the whole GPU tier on the final tree
O=gpurun_out/r06_c38; mkdir -p $O
( time timeout 1100 python -m pytest tests -m gpu -x -q --durations=10 ) > $O/pytest_gpu_full.txt 2>&1
tail -18 $O/pytest_gpu_full.txt
