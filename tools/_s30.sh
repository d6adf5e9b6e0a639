set -x
mkdir -p gpurun_out/s30
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/s30/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s30/pytest.log
tail -5 gpurun_out/s30/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s30/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/s30/smoke.log
tail -3 gpurun_out/s30/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s30/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/s30/bench.log
tail -3 gpurun_out/s30/bench.log
