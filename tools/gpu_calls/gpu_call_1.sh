set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
nproc; free -g | head -2; rocm-smi --showmeminfo vram | head -5
(time timeout 1200 python -m pytest tests -m gpu -x -q --durations=15) > gpurun_out/r02a/pytest.log 2>&1
tail -30 gpurun_out/r02a/pytest.log
(time timeout 400 python bench.py --steps 20 --warmup 3) > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
tail -c 3000 gpurun_out/r02a/bench.json; tail -5 gpurun_out/r02a/bench.err
