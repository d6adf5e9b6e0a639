mkdir -p gpurun_out/s43
TSDF_HIP_ALLIN=0 TSDF_HIP_ROWS_PER_BLOCK=16 TSDF_HIP_BLOCKS_PER_CU=4 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/s43/pytest_knobs.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s43/pytest_knobs.log
tail -25 gpurun_out/s43/pytest_knobs.log
