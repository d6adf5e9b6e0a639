#!/usr/bin/env python3
"""Scene B ("README-realistic": camera inside a 10 m volume, sensor range 0..3 m, ~1 % of the voxels in the
frustum) on a res^3 grid: ms per integrateCloud with the brick cull on and off.  Prints one JSON line."""
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
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--color", type=int, default=1)
    a = ap.parse_args()
    sc = synth.scene_b()
    out = {"res": a.res, "color": a.color}
    for cull in (1, 0):
        capi.set_tuning("cull", cull)
        v = TSDFVolumeOctree()
        v.setResolution(a.res, a.res, a.res)
        v.setGridSize(10.0, 10.0, 10.0)
        v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
        v.setSensorDistanceBounds(0.0, 3.0)
        v.setIntegrateColor(bool(a.color))
        v.reset()
        frames = [(synth.scene_b_pose(i, a.frames), ) for i in range(a.frames)]
        data = [(tr, sc.depth(tr), sc.bgra(i)) for i, (tr,) in enumerate(frames)]
        n = [v.integrateCloud(d, c if a.color else None, tr, count=True) for tr, d, c in data[:2]]  # warm-up
        v.synchronize()
        t0 = time.perf_counter()
        for tr, d, c in data:
            v.integrateCloud(d, c if a.color else None, tr)
        v.synchronize()
        out[f"ms_per_frame_cull{cull}"] = (time.perf_counter() - t0) / len(data) * 1e3
        out["observed_voxels"] = n
        v.close()
    capi.set_tuning("cull", 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
