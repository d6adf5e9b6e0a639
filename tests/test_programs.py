"""CPU tier: the reference's OWN `integrate` program (src/prog/integrate.cpp compiled unmodified into
oracle/_ref/ref_integrate against the stand-ins in compat/) pins the oracle's restatement of its per-cloud
loop: same input directory -> the program's mesh.ply == oracle organize + integrate + marching cubes, bit for
bit.  (tests/test_programs_gpu.py then holds the product's bin/integrate to the same files.)"""
import os
import shutil

import numpy as np
import pytest

from cpu_tsdf_amd import capi, synth
from oracle.oracle import OracleVolume
from tests import sequence_util as su

needs_ref = pytest.mark.skipif(not os.path.exists(su.REF_INTEGRATE), reason="oracle/_ref/ref_integrate not built")
W, H, RES, SIZE = 160, 120, 64, 8.0
COMMON = ["--volume-size", SIZE, "--cell-size", SIZE / RES, "--max-cell-size", SIZE / RES, "--width", W, "--height", H,
          "--trunc-dist-pos", 0.3, "--trunc-dist-neg", 0.3]


def oracle_pipeline(indir, n_frames, color, world=False, units=1.0, zero_nans=False, binary_poses=False, organized=False,
                    invert=False, pose_units=1.0, min_weight=0.0):
    p = capi.default_params()
    p.res[:] = (RES, RES, RES)
    p.size[:] = (SIZE, SIZE, SIZE)
    p.image_width, p.image_height = W, H
    # the program's float globals (src/prog/integrate.cpp:350-353), widened by setCameraIntrinsics
    p.fx, p.fy = float(np.float32(525. * W / 640.)), float(np.float32(525. * H / 480.))
    p.cx, p.cy = float(np.float32(np.float32(W) / 2. - 0.5)), float(np.float32(np.float32(H) / 2. - 0.5))
    p.min_sensor_dist, p.max_sensor_dist = 0.0, 3.0
    p.max_dist_pos = p.max_dist_neg = np.float32(0.3)
    p.integrate_color = int(color)
    ov = OracleVolume(p)
    poses = []
    for i in range(n_frames):
        T = su.read_pose(os.path.join(indir, f"cloud_{i:04d}" + (".transform" if binary_poses else ".txt")), binary_poses)
        if invert:
            T = synth.eigen_affine_inverse(T)
        T[:3, 3] *= np.float64(np.float32(pose_units))
        poses.append(T)
    for i in range(n_frames):
        xyz, col = su.read_pcd(os.path.join(indir, f"cloud_{i:04d}.pcd"))
        bgra = col.view(np.uint8).reshape(-1, 4)
        w2c = synth.eigen_affine_inverse(poses[i]) if world else None
        if organized:  # the program copies the cloud as is: depth = pt.z after units / zero-nans / transform
            dep, cc, _ = ov.organize(xyz, bgra, units, zero_nans, w2c)  # only to reuse the point preparation ...
            pts = xyz.astype(np.float32) * np.float32(units) if units != 1.0 else xyz.copy()
            if zero_nans:
                pts[(pts == 0).all(1)] = np.nan
            assert not world
            dep = pts[:, 2].reshape(H, W).copy()
            cc = bgra.reshape(H, W, 4).copy()
        else:
            dep, cc, _ = ov.organize(xyz, bgra, units, zero_nans, w2c)
        rel = synth.eigen_affine_inverse(poses[0]) @ poses[i]
        ov.integrate(dep, cc if color else None, synth.cam_from_vol_f32(rel))
    return ov.march(min_weight, 1 if color else 0)


@needs_ref
@pytest.mark.parametrize("case", ["plain_color", "world_units_binary_poses", "zero_nans_nocolor", "organized"])
def test_reference_program_equals_oracle_pipeline(case):
    d = su.digit_free_dir(case.replace("_", ""))
    try:
        kw, flags = {}, []
        color = True
        if case == "world_units_binary_poses":
            kw = dict(world=True, units=0.001, binary_poses=True)
            flags = ["--world", "--cloud-units", 0.001]
        elif case == "zero_nans_nocolor":
            kw = dict(zero_nans=True)
            flags = ["--zero-nans"]
            color = False
        elif case == "organized":
            kw = dict(organized=True)
            flags = ["--organized"]
        su.make_sequence(os.path.join(d, "in"), n_frames=3, width=W, height=H, binary_poses=kw.get("binary_poses", False),
                         world=kw.get("world", False), units=kw.get("units", 1.0), organized=kw.get("organized", False))
        rc, log = su.run(su.REF_INTEGRATE, ["--in", os.path.join(d, "in"), "--out", os.path.join(d, "out")] + COMMON + flags +
                         (["--color"] if color else []))
        assert rc == 0, log[-2000:]
        v, c, f = su.read_ply(os.path.join(d, "out", "mesh.ply"))
        verts, rgb, cells = oracle_pipeline(os.path.join(d, "in"), 3, color, **kw)
        assert len(verts) > 600
        assert np.array_equal(v.view(np.uint32), verts.view(np.uint32)), "vertices of the reference program's mesh"
        assert np.array_equal(f.ravel(), np.arange(len(verts)))
        if color:
            assert np.array_equal(c, rgb)
    finally:
        shutil.rmtree(d, ignore_errors=True)
