#!/bin/bash
# round 6, call 20: the multi-handle tests with the new relay pairing case
mkdir -p gpurun_out/r06_c20
timeout 600 python -m pytest tests/test_multi_gpu.py -x -q > gpurun_out/r06_c20/pytest_multi.txt 2>&1
tail -5 gpurun_out/r06_c20/pytest_multi.txt
