#!/usr/bin/env python3
"""The GPU twin of fuzz_oracle_vs_reference.py: the PRODUCT (HIP kernels through the C ABI) against the oracle over the
same random space of volumes, cameras and depth images -- voxels after fusion (both layouts), renderView, marching cubes
and getFxn / gradient / Hessian.  Needs a GPU.  One line per case, exit code 1 if anything differs.
usage: python tests/evidence/fuzz_product_vs_oracle.py [--cases 100] [--seed 1]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cpu_tsdf_amd import capi, synth  # noqa: E402
from cpu_tsdf_amd.volume import MarchingCubesTSDFOctree, TSDFVolumeOctree  # noqa: E402
from oracle.oracle import OracleVolume  # noqa: E402
from tests.evidence.fuzz_oracle_vs_reference import same  # noqa: E402

capi.use_test_library()  # the launch knobs (capi.set_tuning) are hooks of libtsdf_hip_test.so: same kernels, same sources


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ref-cull", type=float, default=0.0, help="fraction of the cases whose principal point is pushed up to 40 %% "
                    "off centre.  Every case runs the product's DEFAULT path (round 4: the reference's frustum cull applied per frame) "
                    "against the oracle's culled integrate -- the reference's own behaviour in every regime")
    a = ap.parse_args()
    rng = np.random.RandomState(a.seed)
    bad = []
    for case in range(a.cases):
        res = int(rng.choice([16, 32, 64, 64, 128, 136]))   # 136: not a power of two (no octree levels), pitch padding
        size = float(rng.choice([0.125, 0.3, 1.0, 3.0]))
        W, H = [(48, 36), (64, 48), (80, 60), (160, 120)][rng.randint(4)]
        f = float(rng.uniform(0.5, 1.6)) * W
        fx, fy = f, f * float(rng.uniform(0.9, 1.1))
        cx, cy = W / 2 - 0.5 + float(rng.uniform(-0.3, 0.3)) * W / 2, H / 2 - 0.5 + float(rng.uniform(-0.3, 0.3)) * H / 2
        zmin, zmax = float(rng.choice([0.0, 0.05 * size, 0.4 * size])), float(rng.uniform(1.5, 4.0)) * size
        pos, neg = float(rng.uniform(0.03, 0.25)) * size, float(rng.uniform(0.03, 0.25)) * size
        wmax = float(rng.choice([100.0, 2.0, 3.5, 1.0, 255.0, 300.0, 0.5]))
        color = bool(rng.randint(2))
        layout = int(rng.choice([capi.LAYOUT_AUTO, capi.LAYOUT_F32W]))
        order = int(rng.randint(2))
        res3 = (res, res, res)
        if rng.rand() < 0.3:   # a box of cubic voxels: other counts along y and z (the bench's slabs are such grids)
            res3 = (res, int(rng.choice([res // 2, res, res + 8])), int(rng.choice([res // 2, res + 24])))
        size3 = tuple(size * r / res for r in res3)
        # round 6: about a third of the cases sit where the software-pipelined row loop applies (k_integrate_p / _pc: the
        # PACKED layout with an integer weight limit, equal truncation limits, every voxel in range and in the image)
        friendly = bool(rng.rand() < 0.35)
        if friendly:
            f = float(rng.uniform(0.45, 0.55)) * W
            fx, fy = f, f
            cx, cy = W / 2 - 0.5, H / 2 - 0.5
            zmin, zmax = 0.0, 4.5 * max(size3)
            neg = pos
            wmax = float(rng.choice([100.0, 2.0, 1.0, 255.0]))
            layout = capi.LAYOUT_AUTO
        n_dev = int(rng.choice([1, 1, 2, 3]))    # one volume over several slab handles (all on GPU 0 here)
        if friendly and rng.rand() < 0.7:
            n_dev = 1
        ref_cull = bool(a.ref_cull > 0 and rng.rand() < a.ref_cull) and not friendly
        v = TSDFVolumeOctree()
        v.setResolution(*res3)
        v.setGridSize(*size3)
        if n_dev > 1:
            v.setDevices([0] * n_dev)
        v.setImageSize(W, H)
        v.setCameraIntrinsics(fx, fy, cx, cy)
        v.setSensorDistanceBounds(zmin, zmax)
        v.setDepthTruncationLimits(pos, neg)
        v.setWeightTruncationLimit(wmax)
        v.setIntegrateColor(color)
        v.setTransformOrder(order)
        v.setLayout(layout)
        if ref_cull:  # push the principal point further out: the regime where the cull decides voxels at the image border
            cx, cy = W / 2 - 0.5 + float(rng.uniform(-0.4, 0.4)) * W / 2, H / 2 - 0.5 + float(rng.uniform(-0.4, 0.4)) * H / 2
            v.setCameraIntrinsics(fx, fy, cx, cy)
        # round 6: the launch knobs that pick between the plain and the software-pipelined row loop and the block height, and
        # frame pairing through the pinned ring (two frames per sweep where both poses see the whole slab)
        knobs = {"pipe": int(rng.choice([0, 1, 3])), "rows_per_block": int(rng.choice([8, 16, 32, 64])), "fuse2": int(rng.choice([1, 2]))}
        pairing = bool(rng.rand() < 0.35)
        if friendly and rng.rand() < 0.8:   # most of the friendly cases really take the pipelined loop, colour included (bit 1)
            knobs["pipe"], pairing = 3, False
        for k, val in knobs.items():
            capi.set_tuning(k, val)
        v.reset()
        if pairing:
            v.setFramePairing(True)
        ref_cull = not v.referenceCullIsNoop()  # (for the log line only)
        ov = OracleVolume(v._p)
        sc = synth.Scene(size, W, H, sphere=float(rng.uniform(0.15, 0.35)), box=float(rng.uniform(0.35, 0.49)))
        sc.fx, sc.fy, sc.cx, sc.cy = fx, fy, cx, cy
        sc.h = np.array([0.47 * s3 for s3 in size3]) * float(rng.uniform(0.8, 1.0))
        what, n_pipelined = [], 0
        for i in range(int(rng.randint(2, 6))):
            r = float(rng.uniform(0.1, 2.4)) * size
            far = friendly and rng.rand() < 0.8
            if far:
                r = 2.4 * max(size3)   # far enough for the wide camera to see the whole grid
            eye = rng.normal(size=3)
            eye *= r / np.linalg.norm(eye)
            tr = synth.look_at_pose(eye, target=rng.uniform(-0.2, 0.2, 3) * size * (0.1 if far else 1.0))
            dep = sc.depth(tr, noise_seed=int(rng.randint(1 << 30)), noise_sigma=0.01 * size)
            junk = rng.rand(H, W)
            dep[junk < 0.03] = np.nan
            dep[(junk >= 0.03) & (junk < 0.04)] = 0.0
            dep[(junk >= 0.04) & (junk < 0.05)] = np.inf
            col = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
            n_cpu = ov.integrate_culled(dep, col if color else None, tr, synth.cam_from_vol_f32(tr))
            if pairing:   # (no count through the ring: the voxels below are the check)
                v.integrateCloud(dep, col if color else None, tr, pipelined=True)
                continue
            n_gpu = v.integrateCloud(dep, col if color else None, tr, count=True)
            if n_gpu != n_cpu:
                what.append(f"count{i}")
            if n_dev == 1:   # which launches took the software-pipelined row loop (tsdf_hip_last_launch_info bit 8)
                info = (C.c_int32 * 4)()
                if capi.load().tsdf_hip_last_launch_info(v._need(), info) == 0 and info[0] & 0x100:
                    n_pipelined += 1
        d, w, rgb = v.download()
        if not (same(d, ov.d) and same(w, ov.w)):
            what.append("voxels")
        if color and not np.array_equal(rgb, ov.rgb):
            what.append("rgb")
        for k in range(2):
            r = float(rng.uniform(0.05, 2.0)) * size
            eye = rng.normal(size=3)
            eye *= r / np.linalg.norm(eye)
            tr = synth.look_at_pose(eye, target=rng.uniform(-0.3, 0.3, 3) * size)
            try:
                if not same(v.renderView(tr, 1 + k, camera_frame=False), ov.raycast(tr, 1 + k)):
                    what.append(f"renderView{k}")
            except capi.TsdfHipError as e:
                what.append(f"renderView{k} raised: {e}")
        for wmin in (0.0, 1.0, 1.5, float(rng.choice([0.5, 2.5]))):   # (<= 1: the weight test may be elided, tsdf_march.hip)
            mc = MarchingCubesTSDFOctree()
            mc.setInputTSDF(v)
            mc.setMinWeight(wmin)
            mc.setColorByRGB(color)
            mesh = mc.reconstruct()
            v_m, c_m, _ = ov.march(wmin, 1 if color else 0)
            if not same(mesh["vertices"], v_m) or (color and not np.array_equal(mesh["rgb"], c_m)):
                what.append(f"mesh(w>={wmin})")
        pts = (rng.uniform(-0.55, 0.55, (400, 3)) * size).astype(np.float32)
        ok, val, grad, hess = v.sample(pts)
        ook, oval, ograd, ohess = ov.sample(pts)
        if not (np.array_equal(ok, ook) and same(val[ok], oval[ok]) and same(grad[ok], ograd[ok]) and same(hess[ok], ohess[ok])):
            what.append("getFxn")
        packed = v.getLayout() == capi.LAYOUT_PACKED
        v.close()
        print(f"case {case:4d}: res {'x'.join(map(str, res3)):>11s} slabs {n_dev} size {size:5.3f} {W}x{H} f {fx:6.1f} c ({cx - (W / 2 - 0.5):+5.1f},{cy - (H / 2 - 0.5):+5.1f}) "
              f"z [{zmin:.3f},{zmax:.2f}] trunc {pos / size:.2f}/{neg / size:.2f} wmax {wmax} colour {int(color)} "
              f"{'packed' if packed else 'f32w'} order {order}{' refcull' if ref_cull else ''}{' paired' if pairing else ''} "
              f"pipe {knobs['pipe']} rpb {knobs['rows_per_block']} fuse2 {knobs['fuse2']} kp {n_pipelined} observed {int((ov.w > 0).sum()):8d}  "
              f"{'DIFF ' + ','.join(what) if what else 'ok'}", flush=True)
        if what:
            bad.append((case, what))
    print(f"{a.cases} cases, seed {a.seed}: {len(bad)} with differences {bad[:20]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
