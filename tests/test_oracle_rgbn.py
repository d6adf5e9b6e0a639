"""CPU tier: pin the oracle's RGBNormalized restatement (oracle_integrate_rgbn: RGBNormalized::addObservation
and getRGB, src/lib/octree.cpp:380-402) against tests/golden/reference_rgbn_32.npz, which
tests/golden/make_golden_rgbn.py generated from the reference's own code with setColorMode("RGBNormalized"),
and live against oracle/_ref at another size when that library is present.  Bar: bit equality."""
import os

import numpy as np
import pytest

from cpu_tsdf_amd import synth
from oracle import refbind
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32
from tests.golden.make_golden_rgbn import colour_image
from tests.test_oracle_golden import params

GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_rgbn_32.npz")


def test_rgbn_matches_reference_golden():
    gold = np.load(GOLD)
    res, W, H, size = int(gold["res"]), int(gold["width"]), int(gold["height"]), float(gold["size"])
    sc = synth.scene_a(res, W, H)
    ov = OracleVolume(params(res, W, H, size))
    for i in range(int(gold["n_frames"])):
        tr = synth.turntable_pose(i, int(gold["total"]), size)
        ov.integrate_rgbn(sc.depth(tr), colour_image(sc, i), synth.cam_from_vol_f32(tr))
        assert_same_f32(ov.d, gold[f"d{i}"], f"d after frame {i}")
        assert np.array_equal(ov.w, gold[f"w{i}"].astype(np.float32))
        assert np.array_equal(ov.rgb, gold[f"rgb{i}"]), f"getRGB after frame {i}"
    nan_state = np.isnan(ov.cn[0])
    assert 50 < nan_state.sum() < 0.5 * nan_state.size          # the black block poisoned some voxels for good
    assert (ov.rgb[nan_state] == 0).all()
    assert (ov.rgb[(ov.w > 0) & ~nan_state].max(axis=-1) > 0).mean() > 0.9
    # averaging r/i and i separately is NOT averaging r: the colours differ from the RGB voxel's
    plain = OracleVolume(params(res, W, H, size))
    for i in range(int(gold["n_frames"])):
        tr = synth.turntable_pose(i, int(gold["total"]), size)
        plain.integrate(sc.depth(tr), colour_image(sc, i), synth.cam_from_vol_f32(tr))
    assert np.array_equal(plain.d.view(np.uint32), ov.d.view(np.uint32)) and (plain.rgb != ov.rgb).mean() > 0.05
    # the mesh coloured by getRGB
    v, c, _ = ov.march(0.0, 1)
    assert_same_f32(v, gold["mc_verts"], "mesh")
    assert np.array_equal(c, gold["mc_rgb"])


@pytest.mark.parametrize("order_name", ["default"])
def test_rgbn_matches_reference_live(order_name):
    if not refbind.available():
        pytest.skip("oracle/_ref not built")
    res, W, H = 64, 96, 72
    sc = synth.scene_a(res, W, H)
    rv = refbind.RefVolume(res, sc.size, W, H, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3 * sc.size, color=True,
                           color_mode="RGBNormalized", max_weight=3.0)
    ov = OracleVolume(params(res, W, H, sc.size))
    ov.p.max_weight = 3.0   # saturating weights: the colour means keep moving with the clamped w
    rng = np.random.RandomState(4)
    for i in range(6):
        tr = synth.turntable_pose(i, 6, sc.size, tilt=0.1 * i)
        col = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
        col[rng.rand(H, W) < 0.02, :3] = 0
        dep = sc.depth(tr, noise_seed=50 + i)
        rv.integrate(dep, col, tr)
        ov.integrate_rgbn(dep, col, synth.cam_from_vol_f32(tr))
    d, w, rgb, _, _ = rv.dump_dense()
    assert_same_f32(ov.d, d, "d")
    assert np.array_equal(ov.w, w) and w.max() == 3.0
    assert np.array_equal(ov.rgb, rgb)
    rv.close()
