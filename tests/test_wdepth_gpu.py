"""GPU tier: the two dead weightings of updateVoxel (hpp:200-204) on the product side.

weight_by_depth_ has no setter in the reference; a volume only gets it from a .vol whose header says so.  The
product follows: load() (Python binding and the C++ class alike) switches the handle to the depth-weighted plain
kernel and float weights; weight_by_variance_ (round 3: M_ / nsample_ planes, tests/test_wvar_gpu.py) likewise."""
import os

import numpy as np
import pytest

from cpu_tsdf_amd import capi, synth
from cpu_tsdf_amd.volume import TSDFVolumeOctree
from oracle import refbind
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32, frames, make_volume
from tests.golden.make_golden_wdepth import H, NF, RES, W, frame, patch_weighting, weighted_reference
from tests.test_oracle_golden import params

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_wdepth_32.npz")


def weighted_product(sc, tmp_path, color=True, by_depth=1, by_variance=0, layout=capi.LAYOUT_AUTO, order=0):
    v = TSDFVolumeOctree()
    v.setResolution(RES, RES, RES)
    v.setGridSize(sc.size, sc.size, sc.size)
    v.setImageSize(W, H)
    v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    v.setSensorDistanceBounds(0.0, 3 * sc.size)
    v.setIntegrateColor(color)
    v.reset()
    path = str(tmp_path / "empty.vol")
    v.save(path)
    patch_weighting(path, by_depth, by_variance)
    v.setLayout(layout)
    v.setTransformOrder(order)  # load() builds the new handle with this object's device / layout / transform order
    v.load(path)
    return v


@pytest.mark.parametrize("order", [0, 1])
def test_weight_by_depth_matches_reference_golden_and_oracle(gpu, tmp_path, order):
    gold = np.load(GOLD)
    sc = synth.scene_a(RES, W, H)
    v = weighted_product(sc, tmp_path, order=order)
    assert v.getLayout() == capi.LAYOUT_F32W  # AUTO: weights stop being counts
    pp = params(RES, W, H, sc.size)
    pp.xform_order = order
    ov = OracleVolume(pp)
    for i in range(NF):
        tr, dep, col = frame(sc, i)
        n = v.integrateCloud(dep, col, tr, count=True)
        assert n == ov.integrate(dep, col, synth.cam_from_vol_f32(tr), weight_by_depth=True)
        d, w, rgb = v.download()
        assert_same_f32(d, ov.d, f"d vs oracle, frame {i}")
        assert_same_f32(w, ov.w, f"w vs oracle, frame {i}")
        assert np.array_equal(rgb, ov.rgb)
        if order == 0:  # the golden was made with the oracle build's transform order
            assert_same_f32(d, gold[f"d{i}"], f"d vs reference golden, frame {i}")
            assert_same_f32(w, gold[f"w{i}"], f"w vs reference golden, frame {i}")
            assert np.array_equal(rgb, gold[f"rgb{i}"])
    # the flags survive a save / load round trip of the fused volume (tsdf_volume_octree.cpp:240-241)
    path = str(tmp_path / "fused.vol")
    v.save(path)
    assert open(path, "rb").read().split(b"\n", 14)[12:14] == [b"1", b"0"]
    v.close()


def test_weight_by_depth_without_colour_and_mid_range_weights(gpu, tmp_path):
    sc = synth.scene_a(RES, W, H)
    v = weighted_product(sc, tmp_path, color=False)
    ov = OracleVolume(params(RES, W, H, sc.size, False))
    for i in range(3):
        tr, dep, _ = frame(sc, i)
        dep[dep > 5] = 3.0 + i
        v.integrateCloud(dep, None, tr)
        ov.integrate(dep, None, synth.cam_from_vol_f32(tr), weight_by_depth=True)
    d, w, _ = v.download()
    assert_same_f32(d, ov.d, "d")
    assert_same_f32(w, ov.w, "w")
    v.close()


def test_weight_by_variance_integrates_and_keeps_its_flag(gpu, tmp_path):
    """Round 3: a header with weight_by_variance_ no longer makes integrate refuse (tests/test_wvar_gpu.py pins the
    results); the flag still survives save."""
    sc = synth.scene_a(RES, W, H)
    v = weighted_product(sc, tmp_path, by_depth=0, by_variance=1)
    tr, dep, col = frame(sc, 0)
    assert v.integrateCloud(dep, col, tr, count=True) > 1000
    M, ns = v.downloadVarianceState()
    assert ns.max() == 1 and (ns > 0).sum() > 1000
    path = str(tmp_path / "again.vol")
    v.save(path)
    assert open(path, "rb").read().split(b"\n", 14)[12:14] == [b"0", b"1"]
    v.close()


def test_weight_by_depth_needs_float_weights(gpu, tmp_path):
    sc = synth.scene_a(RES, W, H)
    with pytest.raises(capi.TsdfHipError) as e:
        weighted_product(sc, tmp_path, layout=capi.LAYOUT_PACKED)
    assert e.value.code == capi.E_UNSUPPORTED
    plain, _ = make_volume(32)
    plain.setLayout(capi.LAYOUT_PACKED)
    plain.reset()
    assert capi.load().tsdf_hip_set_weighting(plain._need(), 1, 0) == capi.E_UNSUPPORTED
    plain.close()


@pytest.mark.parametrize("color", [False, True])
def test_plain_kernel_equals_fast_kernel(gpu, color):
    """The plain per-voxel kernel with w_new = 1 (tuning knob "plain_kernel") against the quad kernel and the oracle:
    pins the plain kernel's own arithmetic independently of the weighting."""
    outs = []
    try:
        for knob in (1, 0):
            capi.set_tuning("plain_kernel", knob)
            vol, sc = make_volume(64, color=color, max_weight=3.0)
            vol.setLayout(capi.LAYOUT_F32W)
            vol.reset()
            ov = OracleVolume(vol._p)
            for i, tr, dep, col in frames(sc, 6, 8, noise=True):
                n = vol.integrateCloud(dep, col if color else None, tr, count=True)
                assert n == ov.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr))
            d, w, rgb = vol.download()
            assert_same_f32(d, ov.d, "d")
            assert_same_f32(w, ov.w, "w")
            if color:
                assert np.array_equal(rgb, ov.rgb)
            outs.append((d, w))
            vol.close()
    finally:
        capi.set_tuning("plain_kernel", 0)
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))


def test_weight_by_depth_through_the_cpp_class(gpu):
    """The same through cpu_tsdf::TSDFVolumeOctree (save, patched header, load, templated integrateCloud), driven
    by the C driver that also wraps the reference; and, where oracle/_ref is present, directly against it."""
    dropin = refbind.DROPIN_LIB if os.path.exists(refbind.DROPIN_LIB) else refbind.build_dropin()
    gold = np.load(GOLD)
    sc = synth.scene_a(RES, W, H)
    dv = weighted_reference(sc, lib_path=dropin)
    rv = weighted_reference(sc) if refbind.available() else None
    for i in range(NF):
        tr, dep, col = frame(sc, i)
        dv.integrate(dep, col, tr)
        if rv is not None:
            rv.integrate(dep, col, tr)
    d, w, rgb = dv.download()
    assert_same_f32(d, gold[f"d{NF - 1}"], "d vs reference golden")
    assert_same_f32(w, gold[f"w{NF - 1}"], "w vs reference golden")
    assert np.array_equal(rgb, gold[f"rgb{NF - 1}"])
    if rv is not None:
        d2, w2, rgb2, _, _ = rv.dump_dense()
        assert_same_f32(d, d2, "d vs live reference")
        assert_same_f32(w, w2, "w vs live reference")
        assert np.array_equal(rgb, rgb2)
        rv.close()
    dv.close()
