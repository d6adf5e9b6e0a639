#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own code (oracle/_ref, built by `make -C oracle ref`
where /root/reference exists).  The fixtures travel with the repo so the oracle and the HIP path can be
pinned on machines that have neither /root/reference nor oracle/_ref.

Scene: Scene-A turntable (cpu_tsdf_amd/synth.py), 32^3 grid of 2^-8 m voxels, 80x60 frames, colour on,
dense-mode octree (setMaxVoxelSize = voxel size).  Stored: inputs (poses), and the reference's outputs:
d/w/rgb grids after each of 5 frames, renderView clouds for 3 poses, getFxn/Gradient/Hessian at 400
points, marching-cubes meshes (3 colour modes x 2 min weights)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cpu_tsdf_amd import synth  # noqa: E402
from oracle.refbind import RefVolume, available  # noqa: E402

RES, W, H, NF, TOTAL = 32, 80, 60, 5, 8


def main():
    assert available(), "build oracle/_ref first (make -C oracle ref)"
    sc = synth.scene_a(RES, W, H)
    rv = RefVolume(RES, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=True, dense=True)
    out = {"res": RES, "width": W, "height": H, "size": np.float32(sc.size), "n_frames": NF, "total": TOTAL}
    for i in range(NF):
        tr = synth.turntable_pose(i, TOTAL, sc.size)
        rv.integrate(sc.depth(tr), sc.bgra(i), tr)
        d, w, rgb, leaf, _ = rv.dump_dense()
        assert (leaf == np.float32(sc.size / RES)).all()
        out[f"d{i}"], out[f"w{i}"], out[f"rgb{i}"] = d, w.astype(np.uint8), rgb
        assert np.array_equal(w, w.astype(np.uint8).astype(np.float32))
    views = [synth.turntable_pose(1, TOTAL, sc.size), synth.turntable_pose(5, 16, sc.size, tilt=0.4),
             synth.look_at_pose((0.02, 0.01, -0.11), target=(0.0, 0.0, 0.05))]
    out["view_poses"] = np.stack(views)
    for k, tr in enumerate(views):
        out[f"view{k}"] = rv.render_view(tr, 1)[0][..., :6]
    out["view0_ds2"] = rv.render_view(views[0], 2)[0][..., :6]
    rng = np.random.RandomState(42)
    pts = rng.uniform(-0.07, 0.07, (400, 3)).astype(np.float32)
    pts[:6] = [[0, 0, 0], [0.0625, 0, 0], [-0.0625, 0.01, 0], [0.5, 0, 0], [0.0605, 0.0605, 0.0605], [-0.0615, 0, 0.03]]
    ok, val, grad, hess = rv.sample(pts)
    out.update(sample_pts=pts, sample_ok=ok, sample_val=val, sample_grad=grad, sample_hess=hess)
    for mode in (0, 1, 2):
        for wmin in (0.0, 2.0):
            v, c, polys, _ = rv.march(wmin, mode)
            key = f"mc_m{mode}_w{int(wmin)}"
            out[key + "_verts"] = v
            if c is not None:
                out[key + "_rgb"] = c
            assert np.array_equal(polys.ravel(), np.arange(len(v), dtype=np.uint32))
    path = os.path.join(ROOT, "tests", "golden", "reference_32.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
