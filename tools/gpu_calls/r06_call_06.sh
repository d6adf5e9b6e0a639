#!/bin/bash
# multi-set / ZSlabVolume / bench frame pairing tests; the C++ drop-in's rate after the strip fix; the predicted scaling table
O=gpurun_out/r06_c06; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_multi_gpu.py tests/test_bench_dist_gpu.py tests/test_zslab_gpu.py tests/test_dropin_gpu.py tests/test_programs_gpu.py -x -q -m gpu 2>&1 | tail -12 ) 2>&1 | tee $O/pytest.txt
timeout 600 python tools/cpp_path_timing.py 60 2048 > $O/cpp_path_timing.json 2> $O/cpp_path_timing.err
python - <<'P'
import json
d = json.load(open("gpurun_out/r06_c06/cpp_path_timing.json"))["2048^3"]
for k, v in d.items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms_in_integrateCloud_call_median", "sustained_frames_per_s", "fraction_of_resident_rate", "frames_per_s", "fused2_frames_per_s", "kernel_ms", "error")})
P
timeout 900 python tools/predict_scaling.py --color 1 > $O/predicted_scaling.json 2> $O/predicted_scaling.txt; cat $O/predicted_scaling.txt
