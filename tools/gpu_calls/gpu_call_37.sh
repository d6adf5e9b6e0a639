cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
for p in 1 2 3; do echo "process $p, alloc_tries 3"; timeout -s KILL 200 python tools/bimodal_probe.py 2>&1 | tail -5; done 2>&1 | tee gpurun_out/r02i/placement.txt
for p in 1 2; do echo "process $p, alloc_tries 1"; TSDF_HIP_ALLOC_TRIES=1 timeout -s KILL 200 python tools/bimodal_probe.py 2>&1 | tail -5; done 2>&1 | tee -a gpurun_out/r02i/placement.txt
