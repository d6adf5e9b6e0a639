cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
(time timeout -s KILL 800 python tests/evidence/fuzz_programs.py --cases 60 --seed 2) > gpurun_out/r02h/fuzz_programs_seed2.log 2>&1
tail -4 gpurun_out/r02h/fuzz_programs_seed2.log; grep -c "  ok" gpurun_out/r02h/fuzz_programs_seed2.log; grep "DIFF\|Error\|Traceback" gpurun_out/r02h/fuzz_programs_seed2.log | head
