#!/bin/bash
# the whole GPU suite on the new kernels; the default bench line (with the secondary keys and the C++ drop-in leg) and its wall
# time; the C++ drop-in's end-to-end rate (tools/cpp_path_timing.py)
O=gpurun_out/r06_c05; mkdir -p $O
( time timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) 2>&1 | tee $O/pytest_gpu.txt
( time timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4 | tee $O/bench_wall.txt
python - <<'P'
import json
d = json.load(open("gpurun_out/r06_c05/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"])
for k, v in (d.get("extras", {}).get("keys") or {}).items():
    print(k, {a: v.get(a) for a in ("kernel_ms", "frac", "bytes_moved_per_launch", "instance", "error")})
print("host_path", {k: v for k, v in d.get("host_path", {}).items() if k != "cpp_dropin" and k != "note"})
print("fused2", (d.get("extras", {}).get("fused2") or {}).get("ms_per_frame"))
P
timeout 600 python tools/cpp_path_timing.py 60 2048 > $O/cpp_path_timing.json 2> $O/cpp_path_timing.err; cat $O/cpp_path_timing.json | head -80
