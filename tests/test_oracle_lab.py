"""CPU tier: pin the oracle's LAB restatement (oracle_rgb2lab / oracle_lab2rgb / oracle_integrate_lab: RGB2LAB,
LAB2RGB and LABNode, src/lib/octree.cpp:436-551) against tests/golden/reference_lab_32.npz, which
tests/golden/make_golden_lab.py generated from the reference's own code with setColorMode("LAB"), and live against
oracle/_ref -- every one of the 2^24 colours through RGB2LAB, millions of means through LAB2RGB, and a saturating
random-colour fusion -- when that library is present.  Bar: bit equality (the oracle calls the same libm pow)."""
import ctypes as C
import os

import numpy as np
import pytest

from cpu_tsdf_amd import capi, synth
from oracle import oracle, refbind
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32
from tests.golden.make_golden_lab import colour_image
from tests.test_oracle_golden import params

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_lab_32.npz")


def test_conversions_match_the_reference_golden():
    gold = np.load(GOLD)
    assert_same_f32(oracle.rgb2lab(gold["probe_rgb"]), gold["probe_lab"], "RGB2LAB")
    assert np.array_equal(oracle.lab2rgb(gold["probe_mix"]), gold["probe_mix_rgb"])
    lab = oracle.rgb2lab(np.uint8([[255, 255, 255], [0, 0, 0], [255, 0, 0]]))
    assert abs(lab[0, 0] - 100) < 0.01 and abs(lab[0, 1]) < 0.01 and abs(lab[1]).max() < 1e-5 and lab[2, 1] > 70


def test_lab_matches_reference_golden():
    gold = np.load(GOLD)
    res, W, H, size = int(gold["res"]), int(gold["width"]), int(gold["height"]), float(gold["size"])
    sc = synth.scene_a(res, W, H)
    ov = OracleVolume(params(res, W, H, size))
    for i in range(int(gold["n_frames"])):
        tr = synth.turntable_pose(i, int(gold["total"]), size)
        ov.integrate_lab(sc.depth(tr), colour_image(sc, i), synth.cam_from_vol_f32(tr))
        assert_same_f32(ov.d, gold[f"d{i}"], f"d after frame {i}")
        assert np.array_equal(ov.w, gold[f"w{i}"].astype(np.float32))
        assert np.array_equal(ov.rgb, gold[f"rgb{i}"]), f"getRGB after frame {i}"
    assert (ov.rgb[ov.w > 0].max(axis=-1) > 0).mean() > 0.5
    # averaging in LAB is not averaging in RGB: the colours differ from the RGB voxel's
    plain = OracleVolume(params(res, W, H, size))
    for i in range(int(gold["n_frames"])):
        tr = synth.turntable_pose(i, int(gold["total"]), size)
        plain.integrate(sc.depth(tr), colour_image(sc, i), synth.cam_from_vol_f32(tr))
    assert np.array_equal(plain.d.view(np.uint32), ov.d.view(np.uint32)) and (plain.rgb != ov.rgb).mean() > 0.05
    v, c, _ = ov.march(0.0, 1)
    assert_same_f32(v, gold["mc_verts"], "mesh")
    assert np.array_equal(c, gold["mc_rgb"])


def test_every_colour_matches_the_reference_live():
    if not refbind.available():
        pytest.skip("oracle/_ref not built")
    i = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([(i >> 16) & 255, (i >> 8) & 255, i & 255], 1).astype(np.uint8)
    lab = oracle.rgb2lab(rgb)
    assert_same_f32(lab, refbind.ref_rgb2lab(rgb), "RGB2LAB of all 2^24 colours")
    assert np.array_equal(oracle.lab2rgb(lab[::5]), refbind.ref_lab2rgb(lab[::5]))
    rng = np.random.RandomState(0)   # out-of-gamut and negative means too
    wild = (rng.rand(2_000_000, 3) * [140, 400, 400] - [20, 200, 200]).astype(np.float32)
    assert np.array_equal(oracle.lab2rgb(wild), refbind.ref_lab2rgb(wild))


def test_lab_matches_reference_live():
    if not refbind.available():
        pytest.skip("oracle/_ref not built")
    res, W, H = 64, 96, 72
    sc = synth.scene_a(res, W, H)
    rv = refbind.RefVolume(res, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=True,
                           color_mode="LAB", max_weight=3.0)
    ov = OracleVolume(params(res, W, H, sc.size))
    ov.p.max_weight = 3.0   # saturating weights: the colour means keep moving with the clamped w
    rng = np.random.RandomState(4)
    for i in range(6):
        tr = synth.turntable_pose(i, 6, sc.size, tilt=0.1 * i)
        col = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
        col[rng.rand(H, W) < 0.02, :3] = 0
        dep = sc.depth(tr, noise_seed=50 + i)
        rv.integrate(dep, col, tr)
        ov.integrate_lab(dep, col, synth.cam_from_vol_f32(tr))
    d, w, rgb, _, _ = rv.dump_dense()
    assert_same_f32(ov.d, d, "d")
    assert np.array_equal(ov.w, w) and w.max() == 3.0
    assert np.array_equal(ov.rgb, rgb)
    rv.close()


@pytest.mark.parametrize("mode", ["LAB", "RGBNormalized"])
def test_colour_mode_fuzz_equals_compiled_reference(mode):
    """The non-default colour voxels far from Scene A: random grids, intrinsics, asymmetric truncation, small weight
    limits, cameras anywhere, junk depth values, random colours with black pixels.  d, w and getRGB() of every voxel equal
    the reference's own after every sequence."""
    if not refbind.available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.RandomState(77 if mode == "LAB" else 78)
    observed = 0
    for case in range(10):
        res = int(rng.choice([16, 32]))
        size = float(rng.choice([0.125, 0.3, 1.0, 3.0]))
        W, H = (48, 36) if case % 2 else (64, 48)
        f = float(rng.uniform(20.0, 80.0))
        fx, fy, cx, cy = f, f * float(rng.uniform(0.9, 1.1)), W / 2 - 0.5 + float(rng.uniform(-3, 3)), H / 2 - 0.5
        zmin, zmax = float(rng.choice([0.0, 0.05 * size])), float(rng.uniform(0.8, 3.5)) * size
        pos, neg = float(rng.uniform(0.02, 0.3)) * size, float(rng.uniform(0.02, 0.3)) * size
        wmax = float(rng.choice([100.0, 2.0, 3.5]))
        p = params(res, W, H, size, True)
        p.fx, p.fy, p.cx, p.cy = fx, fy, cx, cy
        p.min_sensor_dist, p.max_sensor_dist = zmin, zmax
        while not capi.load().tsdf_hip_reference_cull_is_noop(C.byref(p)):
            # a narrow camera with the principal point this far off centre: the reference's frustum cull would drop voxels
            # at one image border (tests/test_oracle_golden.py::test_reference_frustum_cull_regimes); stay where it cannot
            cx = W / 2 - 0.5 + 0.5 * (cx - (W / 2 - 0.5))
            p.cx = cx
        rv = refbind.RefVolume(res, size, W, H, fx, fy, cx, cy, zmin, zmax, trunc=(pos, neg), max_weight=wmax, color=True,
                               color_mode=mode)
        p.max_dist_pos, p.max_dist_neg, p.max_weight = pos, neg, wmax
        ov = OracleVolume(p)
        step = ov.integrate_lab if mode == "LAB" else ov.integrate_rgbn
        for i in range(5):
            eye = rng.uniform(-1.6, 1.6, 3) * size * (1.0 if rng.rand() < 0.7 else 0.2)
            tr = synth.look_at_pose(eye, target=rng.uniform(-0.3, 0.3, 3) * size)
            dep = (rng.uniform(0.2, 2.5, (H, W)) * size).astype(np.float32)
            junk = rng.rand(H, W)
            dep[junk < 0.05] = np.nan
            dep[(junk >= 0.05) & (junk < 0.07)] = np.inf
            dep[(junk >= 0.07) & (junk < 0.09)] = 0.0
            col = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
            col[rng.rand(H, W) < 0.03, :3] = 0
            col[rng.rand(H, W) < 0.03, :3] = 255
            rv.integrate(dep, col, tr)
            observed += step(dep, col, synth.cam_from_vol_f32(tr))
        d, w, rgb, _, _ = rv.dump_dense()
        assert_same_f32(d, ov.d, f"case {case}: d")
        assert_same_f32(w, ov.w, f"case {case}: w")
        assert np.array_equal(rgb, ov.rgb), f"case {case}: getRGB"
        rv.close()
    assert observed > 20000
