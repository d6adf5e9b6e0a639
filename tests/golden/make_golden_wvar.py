#!/usr/bin/env python3
"""Generate tests/golden/reference_wvar_32.npz from the REFERENCE's own code (oracle/_ref): integrateCloud with
weight_by_variance_ = true (include/cpu_tsdf/impl/tsdf_volume_octree.hpp:203-204, OctreeNode::M_ / nsample_ /
getVariance, src/lib/octree.cpp:152-163,281-287).  The flag has no setter; it only becomes true through load()
(src/lib/tsdf_volume_octree.cpp:266), so -- as make_golden_wdepth.py does for weight_by_depth_ -- the empty volume is
saved, the header line patched and the file loaded back into the reference before the frames are integrated.

Scene: Scene-A turntable with noisy depth (the variance of a voxel's distance only exists with noise), 32^3 grid,
80x60 frames, colour on, dense-mode octree, 12 frames: the weighting starts once a voxel has MORE than five samples.
Stored: d / w / rgb after frames 6..12 and a .vol the reference saved after frame 8 (its M_ / nsample_ are in it), from
which the tests continue."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cpu_tsdf_amd import synth  # noqa: E402
from oracle.refbind import RefVolume, available  # noqa: E402
from tests.golden.make_golden_wdepth import patch_weighting  # noqa: E402

RES, W, H, NF, TOTAL, SAVE_AT = 32, 80, 60, 12, 12, 8


def frame(sc, i):
    tr = synth.turntable_pose(i % 4, 16, sc.size, tilt=0.03 * (i % 3))  # few distinct poses: voxels collect many samples
    dep = sc.depth(tr, noise_seed=900 + i, noise_sigma=0.004).copy()
    dep[20:24, 5:9] = np.nan
    return tr, dep, sc.bgra(i)


def variance_reference(sc, color=True, by_depth=False, lib_path=None):
    kw = {"lib_path": lib_path} if lib_path else {}
    rv = RefVolume(RES, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=color, dense=True, **kw)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "empty.vol")
        rv.save(path)
        patch_weighting(path, int(by_depth), 1)
        rv.load(path)
    return rv


def main():
    assert available(), "build oracle/_ref first (make -C oracle ref)"
    sc = synth.scene_a(RES, W, H)
    rv = variance_reference(sc)
    out = {"res": RES, "width": W, "height": H, "size": np.float32(sc.size), "n_frames": NF, "total": TOTAL, "save_at": SAVE_AT}
    vol_path = os.path.join(ROOT, "tests", "golden", "reference_wvar_32_after8.vol")
    for i in range(NF):
        tr, dep, col = frame(sc, i)
        rv.integrate(dep, col, tr)
        if i + 1 >= 6:
            d, w, rgb, leaf, _ = rv.dump_dense()
            out[f"d{i}"], out[f"w{i}"], out[f"rgb{i}"] = d, w, rgb
        if i + 1 == SAVE_AT:
            rv.save(vol_path)
    w_last = out[f"w{NF - 1}"]
    assert ((w_last % 1) != 0).mean() > 0.05, "the variance weighting never produced a fractional weight"
    path = os.path.join(ROOT, "tests", "golden", "reference_wvar_32.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", vol_path, os.path.getsize(vol_path) // 1024, "KiB;",
          "fractional weights:", float(((w_last % 1) != 0).mean()))


if __name__ == "__main__":
    main()
