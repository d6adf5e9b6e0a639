#!/usr/bin/env python3
"""The drop-in claim itself, hunted at random: ONE driver (oracle/ref_capi.cpp, the reference's public C++ API only)
compiled twice -- against the reference's own sources (oracle/_ref/libcpu_tsdf_ref.so, CPU) and against this repo's
headers + libcpu_tsdf_hip.so (the MI355X drop-in) -- is fed the same random volumes, cameras and depth images through
cpu_tsdf::TSDFVolumeOctree::integrateCloud, and every output of the public API is compared bit for bit: voxels,
renderView (camera frame), renderColoredView colours, MarchingCubesTSDFOctree::reconstruct, getFxn / gradient / Hessian,
and save() on one side followed by load() on the OTHER.  Needs a GPU and oracle/_ref.
usage: python tests/evidence/fuzz_dropin_vs_reference.py [--cases 60] [--seed 1]"""
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
from tests.evidence.fuzz_oracle_vs_reference import same  # noqa: E402
from tests.test_oracle_golden import params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ref-cull", type=float, default=0.0, help="fraction of the cases whose principal point is pushed up to 40 %% "
                    "off centre (the reference's frustum cull then drops voxels that project into the image).  The drop-in is "
                    "driven through the reference's own API only -- no call to setReferenceCull or any other method the "
                    "reference lacks (VERDICT r03 #1): its default integrateCloud must follow the reference in every regime")
    a = ap.parse_args()
    assert refbind.available(), "oracle/_ref missing"
    dropin = refbind.DROPIN_LIB if os.path.exists(refbind.DROPIN_LIB) else refbind.build_dropin()
    rng = np.random.RandomState(a.seed)
    bad = []
    tmp = tempfile.mkdtemp()
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
        p.fx, p.fy, p.cx, p.cy, p.max_sensor_dist = fx, fy, cx, cy, zmax
        ref_cull = bool(a.ref_cull > 0 and rng.rand() < a.ref_cull)
        if ref_cull:  # push the principal point further out: the regime the mode exists for
            cx, cy = W / 2 - 0.5 + float(rng.uniform(-0.4, 0.4)) * W / 2, H / 2 - 0.5 + float(rng.uniform(-0.4, 0.4)) * H / 2
            p.cx, p.cy = cx, cy
        devices = [0] * int(rng.choice([1, 1, 2, 3]))
        kw = dict(trunc=(pos, neg), max_weight=wmax, color=color)
        cull_active = not capi.load().tsdf_hip_reference_cull_is_noop(C.byref(p))
        pairing = bool(rng.rand() < 0.4)   # round 6: TSDFVolumeOctree::setFramePairing on the drop-in side only (same voxels)
        gv = refbind.RefVolume(res, size, W, H, fx, fy, cx, cy, zmin, zmax, lib_path=dropin,
                               devices=devices if len(devices) > 1 else None, frame_pairing=pairing, **kw)
        rv = refbind.RefVolume(res, size, W, H, fx, fy, cx, cy, zmin, zmax, **kw)
        sc = synth.Scene(size, W, H, sphere=float(rng.uniform(0.15, 0.35)), box=float(rng.uniform(0.35, 0.49)))
        sc.fx, sc.fy, sc.cx, sc.cy = fx, fy, cx, cy
        for i in range(int(rng.randint(2, 6))):
            r = float(rng.uniform(0.1, 2.4)) * size
            eye = rng.normal(size=3)
            eye *= r / np.linalg.norm(eye)
            tr = synth.look_at_pose(eye, target=rng.uniform(-0.2, 0.2, 3) * size)
            dep = sc.depth(tr, noise_seed=int(rng.randint(1 << 30)), noise_sigma=0.01 * size)
            junk = rng.rand(H, W)
            dep[junk < 0.03] = np.nan
            dep[(junk >= 0.03) & (junk < 0.04)] = 0.0
            col = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
            gv.integrate(dep, col, tr)
            rv.integrate(dep, col, tr)
        what = []
        d, w, rgb = gv.download()
        rd, rw, rrgb, _, _ = rv.dump_dense()
        if not (same(d, rd) and same(w, rw)):
            what.append("voxels")
        if color and not np.array_equal(rgb, rrgb):
            what.append("rgb")
        for k in range(2):
            r = float(rng.uniform(0.05, 2.0)) * size
            eye = rng.normal(size=3)
            eye *= r / np.linalg.norm(eye)
            tr = synth.look_at_pose(eye, target=rng.uniform(-0.3, 0.3, 3) * size)
            if color and k == 0:
                (gc, grgb), (rc, rrgb2) = gv.render_colored_view(tr, 1), rv.render_colored_view(tr, 1)
                if not same(gc[..., :6], rc[..., :6]) or not np.array_equal(grgb, rrgb2):
                    what.append("renderColoredView")
            elif not same(gv.render_view(tr, 1 + k)[0][..., :6], rv.render_view(tr, 1 + k)[0][..., :6]):
                what.append(f"renderView{k}")
        for wmin in (0.0, 1.5):
            mode = 1 if color else 0
            v_g, c_g, _, _ = gv.march(wmin, mode)
            v_r, c_r, _, _ = rv.march(wmin, mode)
            if not same(v_g, v_r) or (color and not np.array_equal(c_g, c_r)):
                what.append(f"mesh(w>={wmin})")
        pts = (rng.uniform(-0.55, 0.55, (300, 3)) * size).astype(np.float32)
        gok, gval, ggrad, ghess = gv.sample(pts)
        rok, rval, rgrad, rhess = rv.sample(pts)
        gok, rok = gok.astype(bool), rok.astype(bool)
        if not (np.array_equal(gok, rok) and same(gval[gok], rval[gok]) and same(ggrad[gok], rgrad[gok]) and same(ghess[gok], rhess[gok])):
            what.append("getFxn")
        # checkpoints cross both ways: the drop-in's file into the reference, the reference's file into the drop-in
        fg, fr = os.path.join(tmp, "g.vol"), os.path.join(tmp, "r.vol")
        gv.save(fg)
        rv.save(fr)
        g2 = refbind.RefVolume(res, size, W, H, fx, fy, cx, cy, zmin, zmax, lib_path=dropin, **kw)
        r2 = refbind.RefVolume(res, size, W, H, fx, fy, cx, cy, zmin, zmax, **kw)
        g2.load(fr)
        r2.load(fg)
        d2, w2, rgb2 = g2.download()
        rd2, rw2, rrgb3, _, _ = r2.dump_dense()
        if not (same(d2, rd) and same(w2, rw) and same(rd2, rd) and same(rw2, rw)):
            what.append("save/load")
        if color and not (np.array_equal(rgb2, rrgb) and np.array_equal(rrgb3, rrgb)):
            what.append("save/load rgb")
        for v in (gv, rv, g2, r2):
            v.close()
        print(f"case {case:4d}: res {res:3d} slabs {len(devices)} size {size:5.3f} {W}x{H} f {fx:6.1f} c ({cx - (W / 2 - 0.5):+5.1f},{cy - (H / 2 - 0.5):+5.1f}) "
              f"z [{zmin:.3f},{zmax:.2f}] trunc {pos / size:.2f}/{neg / size:.2f} wmax {wmax} colour {int(color)} "
              f"{'refcull ' if cull_active else ''}{'paired ' if pairing else ''}observed {int((rw > 0).sum()):7d}  {'DIFF ' + ','.join(what) if what else 'ok'}", flush=True)
        if what:
            bad.append((case, what))
    print(f"{a.cases} cases, seed {a.seed}: {len(bad)} with differences {bad[:20]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
