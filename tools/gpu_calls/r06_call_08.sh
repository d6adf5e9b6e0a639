#!/bin/bash
# the colourless key's profile without the extras legs (its fused2 leg now runs the SAME pipelined kernel as the timed launches), and
# the world-1 RCCL leg of the predicted-scaling table
L=gpurun_out/r06_prof2.log; : > $L
timeout 900 bash tools/run_rocprof.sh r06_c0 20 6 "--color 0" noextras >> $L 2>&1
ls gpurun_out/prof_r06_c0/summary_pmc_SQ2.json
cp profiles/r06_predicted_scaling.json gpurun_out/r06_predicted_scaling.json
timeout 300 python tools/predict_scaling.py --rccl-only gpurun_out/r06_predicted_scaling.json
