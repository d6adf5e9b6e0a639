mkdir -p gpurun_out/s42
timeout 900 python tests/evidence/long_run_parity.py --frames 1000 --check-every 250 --pipelined 1 --pairing 0 > gpurun_out/s42/long_run.log 2>&1; echo "rc=$?" >> gpurun_out/s42/long_run.log
tail -5 gpurun_out/s42/long_run.log
ls gpurun_out/*.json gpurun_out/s42 2>/dev/null | head
