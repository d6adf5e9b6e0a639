#!/bin/bash
# round 6, call 27: the C++ drop-in with frame pairing against the compiled reference (test + a hunt seed with pairing on in 40 % of the cases)
O=gpurun_out/r06_c27; mkdir -p $O
timeout 600 python -m pytest tests/test_dropin_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_dropin.txt
timeout 900 python tests/evidence/fuzz_dropin_vs_reference.py --cases 80 --seed 70 --ref-cull 0.3 > $O/fuzz_dropin_vs_reference_seed70.log 2>&1; echo "dropin fuzz rc=$?"; tail -1 $O/fuzz_dropin_vs_reference_seed70.log; grep -c paired $O/fuzz_dropin_vs_reference_seed70.log
