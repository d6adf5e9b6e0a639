#!/usr/bin/env python3
"""Marching-cubes phase timings on a fused 2048^3 Scene-A volume for a few min-weight settings (report only).
TSDF_HIP_LIB_PATH selects an A/B build (tools/build_variant.py)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpu_tsdf_amd import capi, synth  # noqa: E402
from cpu_tsdf_amd.volume import TSDFVolumeOctree  # noqa: E402


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    sc = synth.scene_a(res)
    v = TSDFVolumeOctree()
    v.setResolution(res, res, res)
    v.setGridSize(sc.size, sc.size, sc.size)
    v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    v.setSensorDistanceBounds(0.0, 3 * sc.size)
    v.setIntegrateColor(True)
    v.reset()
    for i in range(8):
        tr = synth.turntable_pose(i, 24, sc.size)
        v.integrateCloud(sc.depth(tr), sc.bgra(i), tr)
    lib, h = capi.load(), v._need()
    out = {}
    for wmin in (1.0, 0.0, 1.0, 0.0):
        n = C.c_uint64(0)
        capi.check(lib.tsdf_hip_march(h, C.c_float(wmin), 1, C.byref(n)), "march")
        ms = (C.c_float * 3)()
        cells = C.c_uint64(0)
        lib.tsdf_hip_march_timing(h, ms, C.byref(cells))
        out[f"w_min={wmin}"] = {"classify_ms": round(ms[0], 3), "sort_scan_ms": round(ms[1], 3), "emit_ms": round(ms[2], 3),
                                "cells": int(cells.value), "triangles": int(n.value)}
    print(os.environ.get("TSDF_HIP_LIB_PATH", "default").split("/")[-2:][0], json.dumps(out))
    v.close()


if __name__ == "__main__":
    main()
