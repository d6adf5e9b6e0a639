#!/usr/bin/env python3
"""CPU hunt for the two dead weightings of updateVoxel (hpp:200-204): the oracle's restatements (integrate with
weight_by_depth, integrate_variance) against the compiled reference (oracle/_ref) on random volumes, cameras, truncations,
weight limits and noisy depth sequences.  The flags have no setter: the reference gets them through a patched .vol header
(tests/golden/make_golden_wdepth.py::patch_weighting), exactly as the golden files were made.  The variance weighting acts
once a voxel has more than five samples, so sequences revisit a few poses many times.  Compared: d, w, rgb of every voxel
after the last frame (bit for bit), and that the weighting really acted (fractional weights).
usage: python tests/evidence/fuzz_oracle_weightings.py [--cases 100] [--seed 1]"""
import argparse
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cpu_tsdf_amd import capi, synth  # noqa: E402
from oracle import refbind  # noqa: E402
from oracle.oracle import OracleVolume  # noqa: E402
from tests.evidence.fuzz_oracle_vs_reference import same  # noqa: E402
from tests.golden.make_golden_wdepth import patch_weighting  # noqa: E402
from tests.test_oracle_golden import params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--detail", type=int, default=-1, help="print the differing voxels of this case")
    a = ap.parse_args()
    assert refbind.available(), "build oracle/_ref first (make -C oracle ref)"
    rng = np.random.RandomState(a.seed)
    bad, acted = [], 0
    for case in range(a.cases):
        res = int(rng.choice([16, 32, 32]))
        size = float(rng.choice([0.3, 1.0, 3.0, 12.0]))      # 12 m: voxels beyond 10 m, where weight_by_depth reaches 0 (and 0/0)
        W, H = [(48, 36), (64, 48), (80, 60)][rng.randint(3)]
        f = float(rng.uniform(0.7, 1.4)) * W
        fx, fy = f, f * float(rng.uniform(0.95, 1.05))
        cx, cy = W / 2 - 0.5 + float(rng.uniform(-0.08, 0.08)) * W / 2, H / 2 - 0.5 + float(rng.uniform(-0.08, 0.08)) * H / 2
        zmin, zmax = 0.0, float(rng.uniform(2.5, 4.0)) * size
        pos, neg = float(rng.uniform(0.03, 0.2)) * size, float(rng.uniform(0.03, 0.2)) * size
        wmax = float(rng.choice([100.0, 5.0, 20.5]))
        color = bool(rng.randint(2))
        by_depth, by_var = [(1, 0), (0, 1), (1, 1)][rng.randint(3)]
        p = params(res, W, H, size, color)
        p.fx, p.fy, p.cx, p.cy = fx, fy, cx, cy
        p.min_sensor_dist, p.max_sensor_dist = zmin, zmax
        p.max_dist_pos, p.max_dist_neg, p.max_weight = pos, neg, wmax
        # (the oracle's weighted integrates have no frustum cull: stay where the reference's cull is a no-op, as the main hunt does)
        while not capi.load().tsdf_hip_reference_cull_is_noop(C.byref(p)):
            cx, cy = W / 2 - 0.5 + 0.5 * (cx - (W / 2 - 0.5)), H / 2 - 0.5 + 0.5 * (cy - (H / 2 - 0.5))
            p.cx, p.cy = cx, cy
        skip = 0 <= case < a.detail  # (replaying one case: the earlier ones only draw their random numbers)
        rv = None if skip else refbind.RefVolume(res, size, W, H, fx, fy, cx, cy, zmin, zmax, trunc=(pos, neg), max_weight=wmax, color=color)
        with tempfile.TemporaryDirectory() as td:
            if skip:
                td = None
            if not skip:
                path = os.path.join(td, "empty.vol")
                rv.save(path)
                patch_weighting(path, by_depth, by_var)
                rv.load(path)
        ov = OracleVolume(p)
        sc = synth.Scene(size, W, H, sphere=float(rng.uniform(0.15, 0.35)), box=float(rng.uniform(0.35, 0.49)))
        sc.fx, sc.fy, sc.cx, sc.cy = fx, fy, cx, cy
        n_poses, n_frames = int(rng.randint(1, 4)), int(rng.randint(7, 15))
        poses = []
        for _ in range(n_poses):
            eye = rng.normal(size=3)
            eye *= float(rng.uniform(1.2, 2.2)) * size / np.linalg.norm(eye)
            poses.append(synth.look_at_pose(eye, target=rng.uniform(-0.1, 0.1, 3) * size))
        sigma = float(rng.choice([0.002, 0.01, 0.03])) * size
        for i in range(n_frames):
            tr = poses[i % n_poses]
            seed_i = int(rng.randint(1 << 30))
            junk = rng.rand(H, W)
            col = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
            if skip:
                continue
            dep = sc.depth(tr, noise_seed=seed_i, noise_sigma=sigma)
            dep[junk < 0.02] = np.nan
            rv.integrate(dep, col, tr)
            T = synth.cam_from_vol_f32(tr)
            if by_var:
                ov.integrate_variance(dep, col if color else None, T, weight_by_depth=bool(by_depth))
            else:
                ov.integrate(dep, col if color else None, T, weight_by_depth=True)
        if skip:
            continue
        d, w, rgb, _, _ = rv.dump_dense()
        what = []
        if not (same(d, ov.d) and same(w, ov.w)):
            what.append("voxels")
            if case == a.detail:
                ne = (d.view(np.uint32) != ov.d.view(np.uint32)) | (w.view(np.uint32) != ov.w.view(np.uint32))
                ne &= ~((np.isnan(d) & np.isnan(ov.d)) & (np.isnan(w) & np.isnan(ov.w)))
                idx = np.argwhere(ne)
                print("differing voxels:", len(idx))
                import math
                print("noop predicate:", capi.load().tsdf_hip_reference_cull_is_noop(C.byref(p)), "fx fy cx cy", fx, fy, cx, cy, "W H", W, H,
                      "(cy+1)/fy", (cy + 1) / fy, "tv", math.tan(1.1 * math.atan(0.5 * H / fy)), "(cx+1)/fx", (cx + 1) / fx, "(W-cx)/fx", (W - cx) / fx,
                      "th", math.tan(1.1 * math.atan(0.5 * W / fx)), "(H-cy)/fy", (H - cy) / fy)
                oc = OracleVolume(p)
                on = OracleVolume(p)
                dep0 = np.full((H, W), 1.0 * size, np.float32)
                print("oracle: observed without / with the cull on a flat far frame:", on.integrate(dep0, None, synth.cam_from_vol_f32(poses[0])),
                      oc.integrate_culled(dep0, None, poses[0], synth.cam_from_vol_f32(poses[0])))
                planes = ov.reference_cull_planes(poses[0]).reshape(6, 4)
                vs = size / res
                for z, y, x in idx[:14]:
                    c = np.array([(x + 0.5) * vs - size / 2, (y + 0.5) * vs - size / 2, (z + 0.5) * vs - size / 2, 1.0])
                    g = np.linalg.inv(poses[0]) @ c
                    print("   cam", g[:3], "pixel", g[0] / g[2] * fx + cx, g[1] / g[2] * fy + cy, "plane dots", (planes @ c.astype(np.float32)).tolist())
                for z, y, x in idx[:12]:
                    print((z, y, x), "ref d,w", d[z, y, x], w[z, y, x], d[z, y, x].view(np.uint32), w[z, y, x].view(np.uint32),
                          "oracle d,w", ov.d[z, y, x], ov.w[z, y, x], ov.d[z, y, x].view(np.uint32), ov.w[z, y, x].view(np.uint32),
                          "M,ns", getattr(ov, "M", np.zeros_like(d))[z, y, x], getattr(ov, "nsample", np.zeros(d.shape, np.int32))[z, y, x])
        if color and not np.array_equal(rgb, ov.rgb):
            what.append("rgb")
        frac = float(((w % 1) != 0).mean())
        acted += frac > 0.01
        rv.close()
        print(f"case {case:4d}: res {res:3d} size {size:6.3f} {W}x{H} f {fx:6.1f} trunc {pos / size:.2f}/{neg / size:.2f} wmax {wmax} colour {int(color)} "
              f"by_depth {by_depth} by_variance {by_var} poses {n_poses} frames {n_frames} sigma {sigma / size:.3f} fractional weights {frac:.2f} "
              f"observed {int((ov.w > 0).sum()) if not np.isnan(ov.w).any() else -1:6d}  {'DIFF ' + ','.join(what) if what else 'ok'}", flush=True)
        if what:
            bad.append((case, what))
    print(f"{a.cases} cases, seed {a.seed}: {len(bad)} with differences {bad[:20]}; the weighting produced fractional weights in {acted} cases")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
