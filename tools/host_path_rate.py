#!/usr/bin/env python3
"""Frames/s of the HOST entry points at the headline size (2048^3, colour, 640x480): frames live in host memory,
integrateCloud is called back to back -- synchronous (upload + kernel + sync per call) vs pipelined
(tsdf_hip_integrate_async: pinned two-slot ring, upload overlapped with the previous kernel).  One JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpu_tsdf_amd import synth  # noqa: E402
from cpu_tsdf_amd.volume import TSDFVolumeOctree  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--color", type=int, default=1)
    a = ap.parse_args()
    sc = synth.scene_a(a.res)
    v = TSDFVolumeOctree()
    v.setResolution(a.res, a.res, a.res)
    v.setGridSize(sc.size, sc.size, sc.size)
    v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    v.setSensorDistanceBounds(0.0, 3 * sc.size)
    v.setIntegrateColor(bool(a.color))
    v.reset()
    data = []
    for i in range(20):
        tr = synth.turntable_pose(i, 44, sc.size)
        data.append((tr, sc.depth(tr), sc.bgra(i) if a.color else None))
    out = {"res": a.res, "color": a.color, "frames": a.frames}
    for mode in ("synchronous", "pipelined", "synchronous", "pipelined"):
        for tr, d, c in data[:4]:  # warm-up
            v.integrateCloud(d, c, tr, pipelined=(mode == "pipelined"))
        v.synchronize()
        t0 = time.perf_counter()
        for i in range(a.frames):
            tr, d, c = data[i % len(data)]
            v.integrateCloud(d, c, tr, pipelined=(mode == "pipelined"))
        v.synchronize()
        dt = time.perf_counter() - t0
        out.setdefault(mode + "_frames_per_s", []).append(a.frames / dt)
    print(json.dumps(out))
    v.close()


if __name__ == "__main__":
    main()
