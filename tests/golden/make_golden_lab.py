#!/usr/bin/env python3
"""Generate tests/golden/reference_lab_32.npz from the REFERENCE's own code (oracle/_ref) with setColorMode("LAB"):
the d / w / getRGB() grids after each of 4 frames, the coloured mesh and a renderColoredView, plus the reference's
own RGB2LAB / LAB2RGB (octree.cpp:436-527) on a fixed set of colours.  Same scene as make_golden.py; the colour
images carry black, white and saturated-primary blocks (the linear branches of both conversions)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cpu_tsdf_amd import synth  # noqa: E402
from oracle.refbind import RefVolume, available, ref_lab2rgb, ref_rgb2lab  # noqa: E402

RES, W, H, NF, TOTAL = 32, 80, 60, 4, 8


def colour_image(sc, i):
    c = sc.bgra(i).copy()
    c[20:30, 30:45, :3] = 0                      # black: every linear branch of RGB2LAB
    c[35:40, 10:20, :3] = 255                    # white
    c[5:12, 50:70, :3] = (255, 0, 0) if i % 2 else (0, 0, 255)   # b,g,r: alternating blue / red over the same voxels
    c[45:55, 40:60, :3] = (3, 9, 6)              # below the 0.0405 knee of the sRGB curve
    return c


def probe_colours():
    rng = np.random.RandomState(77)
    ramp = np.arange(256, dtype=np.uint8)
    grey = np.stack([ramp, ramp, ramp], 1)
    prim = np.concatenate([np.stack([ramp, 0 * ramp, 0 * ramp], 1), np.stack([0 * ramp, ramp, 0 * ramp], 1),
                           np.stack([0 * ramp, 0 * ramp, ramp], 1)])
    return np.concatenate([grey, prim, rng.randint(0, 256, (20000, 3)).astype(np.uint8)])


def main():
    assert available(), "build oracle/_ref first (make -C oracle ref)"
    sc = synth.scene_a(RES, W, H)
    rv = RefVolume(RES, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=True, dense=True,
                   color_mode="LAB")
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
    px = probe_colours()
    out["probe_rgb"], out["probe_lab"] = px, ref_rgb2lab(px)
    rng = np.random.RandomState(78)
    # means of a few observed colours, the shape LABNode::getRGB sees
    mix = (out["probe_lab"][rng.randint(0, len(px), (30000, 3))] * rng.dirichlet((1, 1, 1), 30000)[..., None]
           .astype(np.float32)).sum(1).astype(np.float32)
    out["probe_mix"], out["probe_mix_rgb"] = mix, ref_lab2rgb(mix)
    path = os.path.join(ROOT, "tests", "golden", "reference_lab_32.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
