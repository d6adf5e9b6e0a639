#!/usr/bin/env python3
"""Generate tests/golden/reference_rgbn_32.npz from the REFERENCE's own code (oracle/_ref) with
setColorMode("RGBNormalized"): the d / w / getRGB() grids after each of 4 frames, the coloured mesh and a
renderColoredView.  Same scene as make_golden.py; every frame's colour image carries a block of BLACK pixels
(r = g = b = 0 makes r/i NaN in RGBNormalized::addObservation, octree.cpp:384-387, and the voxel's colour
state stays NaN from then on) and a block of saturated white."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cpu_tsdf_amd import synth  # noqa: E402
from oracle.refbind import RefVolume, available  # noqa: E402

RES, W, H, NF, TOTAL = 32, 80, 60, 4, 8


def colour_image(sc, i):
    c = sc.bgra(i).copy()
    c[20:30, 30:45, :3] = 0          # black: NaN colour state
    c[35:40, 10:20, :3] = 255        # white: intensity 441.67
    if i == 2:
        c[20:30, 30:45, :3] = (9, 200, 31)   # a later real colour cannot heal a NaN voxel
    return c


def main():
    assert available(), "build oracle/_ref first (make -C oracle ref)"
    sc = synth.scene_a(RES, W, H)
    rv = RefVolume(RES, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=True, dense=True,
                   color_mode="RGBNormalized")
    out = {"res": RES, "width": W, "height": H, "size": np.float32(sc.size), "n_frames": NF, "total": TOTAL}
    for i in range(NF):
        tr = synth.turntable_pose(i, TOTAL, sc.size)
        rv.integrate(sc.depth(tr), colour_image(sc, i), tr)
        d, w, rgb, _, _ = rv.dump_dense()
        out[f"d{i}"], out[f"w{i}"], out[f"rgb{i}"] = d, w.astype(np.uint8), rgb
    v, c, _, _ = rv.march(0.0, 1)
    out["mc_verts"], out["mc_rgb"] = v, c
    tr = synth.turntable_pose(1, TOTAL, sc.size)
    cloud, rgb = rv.render_colored_view(tr, 1)
    out["view_pose"], out["view"], out["view_rgb"] = tr, cloud[..., :6], rgb
    path = os.path.join(ROOT, "tests", "golden", "reference_rgbn_32.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
