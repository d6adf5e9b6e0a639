cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e
mkdir -p $O
timeout -s KILL 120 python - <<'PY'
import numpy as np, ctypes as C
from cpu_tsdf_amd import capi
x = np.float32([0.25, 0.5, 0.75, 1.5, 2.5, 3.5, 254.5, 255.4, 255.5, 256.7, 1e9, -0.25, -0.5, -0.75, -3.0, np.nan, np.inf, -np.inf, 0.49999997, 1.4999999])
out = np.zeros(len(x), dtype=np.uint32)
capi.check(capi.load().tsdf_hip_selftest_cvt_pk_u8(capi.as_f32p(x), len(x), out.ctypes.data_as(C.POINTER(C.c_uint32))), "probe")
for a, b in zip(x, out): print(f"cvt_pk_u8({a!r}) -> {b:08x}  byte1={(b>>8)&255}")
PY
V=cpu_tsdf_amd/lib/variants
bench() { TSDF_HIP_LIB_PATH=$1 timeout -s KILL 200 python bench.py --steps 20 --warmup 3 --cpu-baseline 0 --extras 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', 'kernel_ms', j['roofline']['kernel_ms'], 'ms_per_step', j['ms_per_step'], 'frac', j['roofline']['frac'])"; }
for rep in 1 2; do
  bench $V/head/libtsdf_hip.so head
  bench cpu_tsdf_amd/lib/libtsdf_hip.so new
  bench $V/nohinge/libtsdf_hip.so nohinge
  bench $V/pk1/libtsdf_hip.so pk1
  bench $V/pk2/libtsdf_hip.so pk2
done 2>&1 | tee $O/ab.txt
(timeout -s KILL 400 python -m pytest tests/test_div_gpu.py tests/test_integrate_gpu.py tests/test_fullsize_gpu.py tests/test_baseline_configs_gpu.py -m gpu -x -q) 2>&1 | tail -5
for v in pk1 pk2; do echo "== parity with $v"; (TSDF_HIP_LIB_PATH=$V/$v/libtsdf_hip.so timeout -s KILL 200 python -m pytest tests/test_integrate_gpu.py -m gpu -x -q) 2>&1 | tail -3; done
