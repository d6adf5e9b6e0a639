#!/usr/bin/env python3
"""Is the run-to-run bimodality of k_integrate (17.8 vs 18.8 ms at 2048^3) a property of the PROCESS or of the ALLOCATION?
One process creates the bench volume, times a few frames, destroys it, and does it again several times.  (It is the
allocation: tsdf_hip_create therefore probes up to TSDF_HIP_ALLOC_TRIES placements and keeps the fastest; run this with
TSDF_HIP_ALLOC_TRIES=1 to see the raw lottery.)"""
import ctypes as C
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpu_tsdf_amd import capi, synth  # noqa: E402
capi.use_test_library()  # knobs / selftest hooks live in libtsdf_hip_test.so (include/tsdf_hip_test.h)
from cpu_tsdf_amd.volume import TSDFVolumeOctree  # noqa: E402


def one(res=2048, W=640, H=480, frames=8, hold=None):
    voxel = 2.0 ** -8
    S = res * voxel
    sc = synth.Scene(S, W, H)
    v = TSDFVolumeOctree()
    v.setResolution(res, res, res)
    v.setGridSize(S, S, S)
    v.setImageSize(W, H)
    v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    v.setSensorDistanceBounds(0.0, 3.0 * S)
    v.setIntegrateColor(True)
    v.reset()
    ms = []
    for i in range(frames):
        tr = synth.turntable_pose(i, frames, S)
        dep, col = sc.depth(tr), sc.bgra(i)
        v.integrateCloud(dep, col, tr)      # warm path incl. upload
        v.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib, h = capi.load(), v._need()
    fr = torch.empty((2, H, W), dtype=torch.float32, device="cuda")
    fr[0].copy_(torch.from_numpy(sc.depth(tr)))
    fr[1].view(torch.uint8).view(H, W, 4).copy_(torch.from_numpy(sc.bgra(0)))
    T = synth.cam_from_vol_f32(tr)
    v.setStream(torch.cuda.current_stream().cuda_stream)
    for i in range(6):
        e0.record()
        capi.check(lib.tsdf_hip_integrate_device(h, C.c_void_p(fr[0].data_ptr()), C.c_void_p(fr[1].data_ptr()), capi.as_f32p(T), None), "x")
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    sw = []
    br, bw = C.c_uint64(), C.c_uint64()
    for i in range(4):   # k_calib_rmw: a pure read-modify-write sweep of the same planes (synchronous call)
        e0.record()
        capi.check(lib.tsdf_hip_selftest_sweep(h, C.byref(br), C.byref(bw)), "sweep")
        e1.record()
        torch.cuda.synchronize()
        sw.append(e0.elapsed_time(e1))
    pm, ch = (C.c_float * 8)(), C.c_int32(0)
    nt = lib.tsdf_hip_alloc_probe(h, pm, C.byref(ch))
    v.close()
    return float(np.median(ms)), float(np.median(sw[1:])), [round(float(x), 2) for x in pm[:nt]], int(ch.value)


if __name__ == "__main__":
    for k in range(5):
        a, b, pm, ch = one()
        print(f"volume {k}: k_integrate {a:.3f} ms per frame, k_calib_rmw sweep {b:.3f} ms; create probed {pm} and kept {ch}", flush=True)
