#!/usr/bin/env python3
"""Offline hunt for divergences between the oracle and the compiled reference (oracle/_ref), wider than the fuzz cases in
the test suite: N random volumes (resolution, size, intrinsics incl. off-centre principal points inside the regime where
the reference's frustum cull is a no-op, sensor bounds, asymmetric truncation, weight limits, colour on/off, cameras
inside and outside), each fused from a real scene with noise and junk pixels, then compared voxel for voxel, through
renderView from random poses, through marching cubes (two weight thresholds) and through getFxn / gradient / Hessian at
random points.  Prints one line per case and a summary; exit code 1 if anything differs.
usage: python tests/evidence/fuzz_oracle_vs_reference.py [--cases 200] [--seed 1]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cpu_tsdf_amd import capi, synth  # noqa: E402
from cpu_tsdf_amd.volume import transform_cloud_with_normals  # noqa: E402
from oracle import refbind  # noqa: E402
from oracle.oracle import OracleVolume  # noqa: E402
from tests.test_oracle_golden import params  # noqa: E402


def same(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    if a.shape != b.shape:
        return False
    ne = a.view(np.uint32) != b.view(np.uint32)
    ne &= ~((a == b) | (np.isnan(a) & np.isnan(b)))
    return not ne.any()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="replay this case alone (same random stream) and save its inputs")
    ap.add_argument("--culled", type=int, default=0, help="1: the oracle runs getFrustumCulledVoxels' restatement in EVERY frame "
                    "(oracle_integrate_culled, round 3), as the reference does, and principal points go up to 40 %% off centre: "
                    "the regime where the cull drops voxels that project into the image")
    a = ap.parse_args()
    assert refbind.available(), "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.RandomState(a.seed)
    bad = []
    for case in range(a.cases):
        res = int(rng.choice([16, 32, 32, 64]))
        size = float(rng.choice([0.125, 0.3, 1.0, 3.0]))
        W, H = [(48, 36), (64, 48), (80, 60)][rng.randint(3)]
        f = float(rng.uniform(0.5, 1.6)) * W
        fx, fy = f, f * float(rng.uniform(0.9, 1.1))
        cx, cy = W / 2 - 0.5 + float(rng.uniform(-0.2, 0.2)) * W / 2, H / 2 - 0.5 + float(rng.uniform(-0.2, 0.2)) * H / 2
        zmin, zmax = float(rng.choice([0.0, 0.05 * size, 0.4 * size])), float(rng.uniform(1.5, 4.0)) * size
        pos, neg = float(rng.uniform(0.03, 0.25)) * size, float(rng.uniform(0.03, 0.25)) * size
        wmax = float(rng.choice([100.0, 2.0, 3.5, 1.0]))
        color = bool(rng.randint(2))
        p = params(res, W, H, size, color)
        p.fx, p.fy, p.cx, p.cy = fx, fy, cx, cy
        p.min_sensor_dist, p.max_sensor_dist = zmin, zmax
        p.max_dist_pos, p.max_dist_neg, p.max_weight = pos, neg, wmax
        if a.culled:
            cx, cy = W / 2 - 0.5 + float(rng.uniform(-0.4, 0.4)) * W / 2, H / 2 - 0.5 + float(rng.uniform(-0.4, 0.4)) * H / 2
            p.cx, p.cy = cx, cy
        while not a.culled and not capi.load().tsdf_hip_reference_cull_is_noop(C.byref(p)):
            cx, cy = W / 2 - 0.5 + 0.5 * (cx - (W / 2 - 0.5)), H / 2 - 0.5 + 0.5 * (cy - (H / 2 - 0.5))
            p.cx, p.cy = cx, cy
        rv = refbind.RefVolume(res, size, W, H, fx, fy, cx, cy, zmin, zmax, trunc=(pos, neg), max_weight=wmax, color=color)
        ov = OracleVolume(p)
        sc = synth.Scene(size, W, H, sphere=float(rng.uniform(0.15, 0.35)), box=float(rng.uniform(0.35, 0.49)))
        sc.fx, sc.fy, sc.cx, sc.cy = fx, fy, cx, cy
        poses = []
        for i in range(int(rng.randint(2, 6))):
            r = float(rng.uniform(0.1, 2.4)) * size
            eye = rng.normal(size=3)
            eye *= r / np.linalg.norm(eye)
            tr = synth.look_at_pose(eye, target=rng.uniform(-0.2, 0.2, 3) * size)
            poses.append(tr)
            dep = sc.depth(tr, noise_seed=int(rng.randint(1 << 30)), noise_sigma=0.01 * size)
            junk = rng.rand(H, W)
            dep[junk < 0.03] = np.nan
            dep[(junk >= 0.03) & (junk < 0.04)] = 0.0
            dep[(junk >= 0.04) & (junk < 0.05)] = np.inf
            col = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
            rv.integrate(dep, col, tr)
            if a.culled:
                ov.integrate_culled(dep, col if color else None, tr, synth.cam_from_vol_f32(tr))
            else:
                ov.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr))
        what = []
        d, w, rgb, _, _ = rv.dump_dense()
        shell = 0
        if not (same(d, ov.d) and same(w, ov.w) and (not color or np.array_equal(rgb, ov.rgb))):
            # The one expected kind of difference: a voxel whose g.z sits within rounding of the min_/max_sensor_dist plane
            # in some frame.  updateVoxel accepts it (g.z <= max_sensor_dist_), pcl::FrustumCulling's far / near plane
            # test -- other float arithmetic on the forward pose, inside PCL and Eigen -- may not: which way such a voxel
            # falls is not determined by the reference's own sources.  Anything else is a real DIFF.
            ne = (d.view(np.uint32) != ov.d.view(np.uint32)) | (w != ov.w)
            if color:
                ne |= (rgb != ov.rgb).any(-1)
            vs = size / res
            on_shell = np.zeros(int(ne.sum()), bool)
            idx = np.argwhere(ne)
            ctr = (idx[:, ::-1] + 0.5) * vs - size / 2          # (x, y, z) of the differing voxels
            for tr in poses:
                gz = (np.linalg.inv(tr) @ np.c_[ctr, np.ones(len(ctr))].T)[2]
                for plane in (zmin, zmax):
                    if plane > 0:
                        on_shell |= np.abs(gz - plane) <= 4e-6 * plane
            shell = int(on_shell.sum())
            if not on_shell.all():
                what.append("voxels")
        for k in range(2):
            r = float(rng.uniform(0.05, 2.0)) * size
            eye = rng.normal(size=3)
            eye *= r / np.linalg.norm(eye)
            tr = synth.look_at_pose(eye, target=rng.uniform(-0.3, 0.3, 3) * size)
            ds = int(rng.choice([1, 2, 3])) if k else 1
            ref_view, _ = rv.render_view(tr, ds)
            mine = transform_cloud_with_normals(ov.raycast(tr, ds), synth.eigen_affine_inverse(tr))
            if not same(mine[..., :6], ref_view[..., :6]) and not shell:
                what.append(f"renderView{k}")
        for wmin in (0.0, 1.5):
            mode = int(rng.choice([1, 2])) if color else int(rng.choice([0, 2]))   # 2 = setColorByConfidence
            v_r, c_r, _, _ = rv.march(wmin, mode)
            v_m, c_m, _ = ov.march(wmin, mode)
            if (not same(v_m, v_r) or (mode and not np.array_equal(c_m, c_r))) and not shell:
                what.append(f"mesh(w>={wmin})")
        pts = (rng.uniform(-0.55, 0.55, (400, 3)) * size).astype(np.float32)
        ok, val, grad, hess = ov.sample(pts)
        rok, rval, rgrad, rhess = rv.sample(pts)
        if not (np.array_equal(ok.astype(bool), rok.astype(bool)) and same(val[ok], rval[ok]) and same(grad[ok], rgrad[ok])
                and same(hess[ok], rhess[ok])) and not shell:
            what.append("getFxn")
        rv.close()
        print(f"case {case:4d}: res {res:3d} size {size:5.3f} {W}x{H} f {fx:6.1f} c ({cx - (W / 2 - 0.5):+5.1f},{cy - (H / 2 - 0.5):+5.1f}) "
              f"z [{zmin:.3f},{zmax:.2f}] trunc {pos / size:.2f}/{neg / size:.2f} wmax {wmax} colour {int(color)} "
              f"{'cull active ' if a.culled and not capi.load().tsdf_hip_reference_cull_is_noop(C.byref(p)) else ''}observed {int((ov.w > 0).sum()):7d}  {'DIFF ' + ','.join(what) if what else 'ok'}"
              + (f"  ({shell} voxel(s) on the sensor-range shell decided the other way by the reference's cull; derived outputs not compared)" if shell else ""), flush=True)
        if what:
            bad.append((case, what))
    print(f"{a.cases} cases, seed {a.seed}: {len(bad)} with differences {bad[:20]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
