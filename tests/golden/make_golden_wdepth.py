#!/usr/bin/env python3
"""Generate tests/golden/reference_wdepth_32.npz from the REFERENCE's own code (oracle/_ref): integrateCloud with
weight_by_depth_ = true (include/cpu_tsdf/impl/tsdf_volume_octree.hpp:200-202).  The flag has no setter; it only
becomes true through load() (src/lib/tsdf_volume_octree.cpp:265), so the volume is saved, the header line patched
(0 -> 1) and the file loaded back into the reference before the frames are integrated.

Scene: Scene-A turntable, 32^3 grid of 2^-8 m voxels, 80x60 frames, colour on, dense-mode octree.  The depth
images are made harsher than Scene A's on purpose: a patch at 12 m (beyond the 10 m where the weight reaches zero:
an unobserved voxel then becomes 0/0 = NaN, as in the reference), a patch at -1 m (weight > 1) and NaN pixels.
Stored: d / w / rgb after each of 4 frames."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cpu_tsdf_amd import synth  # noqa: E402
from oracle.refbind import RefVolume, available  # noqa: E402

RES, W, H, NF, TOTAL = 32, 80, 60, 4, 8


def frame(sc, i):
    """Depth + colour of frame i (shared with the tests)."""
    tr = synth.turntable_pose(i, TOTAL, sc.size)
    dep = sc.depth(tr).copy()
    dep[5:15, 10:30] = 12.0 + i      # beyond 10 m: w_new = 0
    dep[40:50, 50:70] = -1.0         # behind the camera plane: w_new = 1.1 (the voxel is then far behind: rejected)
    dep[20:24, 5:9] = np.nan
    dep[30:34, 60:64] = 9.5          # w_new = 0.05
    return tr, dep, sc.bgra(i)


def patch_weighting(path, by_depth, by_variance=0):
    """Flip the two header lines of a .vol (tsdf_volume_octree.cpp:240-241 writes them as lines 12 and 13)."""
    raw = open(path, "rb").read()
    lines = raw.split(b"\n", 14)
    assert lines[0].startswith(b"# TSDFVolumeOctree") and lines[12] in (b"0", b"1") and lines[13] in (b"0", b"1")
    lines[12], lines[13] = str(int(by_depth)).encode(), str(int(by_variance)).encode()
    open(path, "wb").write(b"\n".join(lines))


def weighted_reference(sc, color=True, lib_path=None):
    kw = {"lib_path": lib_path} if lib_path else {}
    rv = RefVolume(RES, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=color, dense=True, **kw)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "empty.vol")
        rv.save(path)
        patch_weighting(path, 1)
        rv.load(path)
    return rv


def main():
    assert available(), "build oracle/_ref first (make -C oracle ref)"
    sc = synth.scene_a(RES, W, H)
    rv = weighted_reference(sc)
    out = {"res": RES, "width": W, "height": H, "size": np.float32(sc.size), "n_frames": NF, "total": TOTAL}
    for i in range(NF):
        tr, dep, col = frame(sc, i)
        rv.integrate(dep, col, tr)
        d, w, rgb, leaf, _ = rv.dump_dense()
        assert (leaf == np.float32(sc.size / RES)).all()
        out[f"d{i}"], out[f"w{i}"], out[f"rgb{i}"] = d, w, rgb
    assert np.isnan(out[f"d{NF - 1}"]).any() and (out[f"w{NF - 1}"] % 1 != 0).any()
    path = os.path.join(ROOT, "tests", "golden", "reference_wdepth_32.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
