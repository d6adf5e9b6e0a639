cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
(time timeout -s KILL 900 python tests/evidence/fuzz_dropin_vs_reference.py --cases 80 --seed 1) > gpurun_out/r02h/fuzz_dropin_vs_reference_seed1.log 2>&1
tail -6 gpurun_out/r02h/fuzz_dropin_vs_reference_seed1.log; grep -c " ok$" gpurun_out/r02h/fuzz_dropin_vs_reference_seed1.log; grep "DIFF\|Error\|error" gpurun_out/r02h/fuzz_dropin_vs_reference_seed1.log | head -20
