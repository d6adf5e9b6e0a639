set -x
mkdir -p gpurun_out/s33
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/s33/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s33/pytest.log
tail -6 gpurun_out/s33/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s33/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/s33/smoke.log
tail -2 gpurun_out/s33/smoke.log
