#!/bin/bash
# (1) the software-pipelined colourless kernel (k_integrate_p): parity tests first; (2) A/B by alternation against the plain
# row loop (TSDF_HIP_PIPE=0) and the 8-wave register budget (variant pipe8); (3) the row loop of the OLD colourless instance
# timing itself (-DTSDF_PHASE_TIMER=1 build of the round-5 kernel: the thread trace rocprofv3 --att cannot give here).
O=gpurun_out/r06_c02; mkdir -p $O
timeout 600 python -m pytest tests/test_integrate_gpu.py -x -q -m gpu -k "pipelined or every_reachable or all_inside or implied" 2>&1 | tail -5 | tee $O/pytest_pipe.txt
timeout 900 python tools/ab_alt.py --rounds 5 --out $O/ab_pipe_c0.txt --bench "--color 0" plain=TSDF_HIP_PIPE=0 pipe7=TSDF_HIP_PIPE=1 pipe8=lib=pipe8 2>&1 | tail -8
V=$(pwd)/cpu_tsdf_amd/lib/variants/phase/libtsdf_hip.so
B="python bench.py --warmup 2 --cpu-baseline 0 --scene-b 0 --host-path 0 --extras 0 --steps 6"
rm -f $O/phase_c0.jsonl $O/phase_c1.jsonl
TSDF_HIP_LIB_PATH=$V TSDF_HIP_PHASE_FILE=$O/phase_c0.jsonl timeout 200 $B --color 0 > $O/bench_phase_c0.json 2> $O/bench_phase_c0.err; echo "phase c0 rc=$?"
TSDF_HIP_LIB_PATH=$V TSDF_HIP_PHASE_FILE=$O/phase_c1.jsonl timeout 200 $B --color 1 > $O/bench_phase_c1.json 2> $O/bench_phase_c1.err; echo "phase c1 rc=$?"
for f in phase_c0 phase_c1; do python tools/phase_reduce.py $O/$f.jsonl --skip 3 | tee $O/$f.txt; done
python - <<'P'
import json
for n in ("bench_phase_c0", "bench_phase_c1"):
    try:
        d = json.load(open(f"gpurun_out/r06_c02/{n}.json")); print(n, "kernel_ms", round(d["roofline"]["kernel_ms"], 3))
    except Exception as e:
        print(n, "failed", e)
P
