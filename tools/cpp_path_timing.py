#!/usr/bin/env python3
"""ms per TSDFVolumeOctree::integrateCloud call through the C++ drop-in (templated integrateCloud on a
pcl::PointCloud<PointXYZRGBA>: strip into the pinned slot + upload + k_integrate, pipelined) at 1024^3 and 2048^3,
Scene-A frames, colour on.  Reports the time inside the call (what the caller's thread pays) and the sustained
rate (wall clock over all frames incl. a final download that drains the queue).  One JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpu_tsdf_amd import synth  # noqa: E402
from oracle import refbind  # noqa: E402  (only its ctypes wrapper of the C driver; the library under test is the drop-in)


def main():
    lib = refbind.DROPIN_LIB if os.path.exists(refbind.DROPIN_LIB) else refbind.build_dropin()
    out = {}
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    for res in (1024, 2048):
        W, H = 640, 480
        sc = synth.scene_a(res, W, H)
        dv = refbind.RefVolume(res, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=True, lib_path=lib)
        frames = [(synth.turntable_pose(i, 16, sc.size),) for i in range(16)]
        frames = [(tr, sc.depth(tr), sc.bgra(i)) for i, (tr,) in enumerate(frames)]
        for tr, dep, col in frames[:4]:  # warm-up (pinned ring, first launches)
            dv.integrate(dep, col, tr)
        dv.L.ct_voxel_center  # (keep the library alive)
        in_call = []
        t0 = time.perf_counter()
        for i in range(n_frames):
            tr, dep, col = frames[i % 16]
            in_call.append(dv.integrate(dep, col, tr))
        # drain: a tiny readback is ordered after every queued frame
        pts = np.zeros((1, 3), np.float32)
        dv.sample(pts)
        wall = time.perf_counter() - t0
        out[f"{res}^3"] = {"frames": n_frames, "ms_in_integrateCloud_call_median": float(np.median(in_call)) * 1e3,
                           "ms_in_integrateCloud_call_mean": float(np.mean(in_call)) * 1e3,
                           "sustained_ms_per_frame_incl_cloud_build": wall / n_frames * 1e3,
                           "note": "the sustained figure includes the C driver building the 9.8 MB PointXYZRGBA cloud per frame "
                                   "(single-threaded, outside the timed call) -- the caller's own work"}
        dv.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
