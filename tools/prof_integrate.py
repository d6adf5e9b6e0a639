#!/usr/bin/env python3
"""Small torch-free driver for rocprofv3 passes over the integrate kernel.

Runs, on one GPU: `--calib` calibration sweeps (k_calib_rmw, exactly known bytes) followed by
`--steps` integrateCloud launches (k_integrate) of Scene-A frames on a res^3 grid.  Prints one JSON
line with the known sweep bytes, the algorithmic bytes per integrate launch and HIP-event timings.
Meant to be wrapped as `rocprofv3 --pmc <counter> ... -- python tools/prof_integrate.py ...`; the
counter CSV is then reduced by tools/pmc_reduce.py.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpu_tsdf_amd import capi, synth  # noqa: E402
capi.use_test_library()  # knobs / selftest hooks live in libtsdf_hip_test.so (include/tsdf_hip_test.h)
from cpu_tsdf_amd.volume import TSDFVolumeOctree  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=2048)
    ap.add_argument("--planes", type=int, default=0, help="only the central N planes (0 = all)")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--calib", type=int, default=2)
    ap.add_argument("--color", type=int, default=1)
    ap.add_argument("--layout", choices=["auto", "f32w", "packed"], default="auto")
    ap.add_argument("--total", type=int, default=44, help="turntable length the frames are taken from")
    a = ap.parse_args()
    res = a.res
    sc = synth.scene_a(res)
    v = TSDFVolumeOctree()
    v.setResolution(res, res, res)
    v.setGridSize(sc.size, sc.size, sc.size)
    v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    v.setSensorDistanceBounds(0.0, 3 * sc.size)
    v.setIntegrateColor(bool(a.color))
    v.setLayout({"auto": capi.LAYOUT_AUTO, "f32w": capi.LAYOUT_F32W, "packed": capi.LAYOUT_PACKED}[a.layout])
    if a.planes:
        zb = (res - a.planes) // 2
        v.setZSlab(zb, zb + a.planes)
    v.reset()
    lib = capi.load()
    h = v._need()
    br, bw = C.c_uint64(), C.c_uint64()
    t0 = time.perf_counter()
    for _ in range(a.calib):
        capi.check(lib.tsdf_hip_selftest_sweep(h, C.byref(br), C.byref(bw)), "sweep")
    t_sweep = (time.perf_counter() - t0) / max(1, a.calib)
    n_obs = []
    t_int = []
    for i in range(a.warmup + a.steps):
        tr = synth.turntable_pose(i, a.total, sc.size)
        dep, col = sc.depth(tr), sc.bgra(i)
        t0 = time.perf_counter()
        n = v.integrateCloud(dep, col if a.color else None, tr, count=True)
        if i >= a.warmup:
            t_int.append(time.perf_counter() - t0)
            n_obs.append(n)
    W, H = sc.width, sc.height
    bpv, bpp = (24, 8) if a.color else (16, 4)
    packed = v.getLayout() == capi.LAYOUT_PACKED
    lbpv = (16 if a.color else 10) if packed else bpv
    print(json.dumps({
        "res": res, "planes": a.planes or res, "color": a.color, "layout": "packed" if packed else "f32w",
        "layout_bytes_per_launch": lbpv * float(np.mean(n_obs)) + bpp * W * H, "calib_launches": a.calib,
        "integrate_launches": a.warmup + a.steps,
        "sweep_bytes_read": br.value, "sweep_bytes_written": bw.value, "sweep_wall_ms": t_sweep * 1e3,
        "n_obs_mean": float(np.mean(n_obs)),
        "alg_bytes_per_launch": bpv * float(np.mean(n_obs)) + bpp * W * H,
        "integrate_wall_ms_incl_upload": float(np.mean(t_int)) * 1e3,
    }))
    v.close()


if __name__ == "__main__":
    main()
