"""GPU tier: setColorMode("RGBNormalized") -- k_integrate_rgbn (RGBNormalized::addObservation / getRGB,
src/lib/octree.cpp:380-402) vs the reference's own outputs (tests/golden/reference_rgbn_32.npz) and vs the oracle
on seeded inputs: d, w bit for bit, the colours every reader sees (download, marching cubes, renderColoredView)
byte for byte, including voxels whose colour state went NaN on a black pixel."""
import os

import numpy as np
import pytest

from cpu_tsdf_amd import capi, synth
from cpu_tsdf_amd.volume import MarchingCubesTSDFOctree
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32, make_volume
from tests.golden.make_golden_rgbn import colour_image

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_rgbn_32.npz")


def test_rgbn_matches_the_reference_golden(gpu):
    gold = np.load(GOLD)
    res, W, H = int(gold["res"]), int(gold["width"]), int(gold["height"])
    vol, sc = make_volume(res, W, H, color=True)
    vol.setColorMode("RGBNormalized")
    vol.reset()
    assert vol.getLayout() == capi.LAYOUT_F32W
    for i in range(int(gold["n_frames"])):
        tr = synth.turntable_pose(i, int(gold["total"]), sc.size)
        vol.integrateCloud(sc.depth(tr), colour_image(sc, i), tr)
        d, w, rgb = vol.download()
        assert_same_f32(d, gold[f"d{i}"], f"d after frame {i}")
        assert np.array_equal(w, gold[f"w{i}"].astype(np.float32))
        assert np.array_equal(rgb, gold[f"rgb{i}"]), f"colours after frame {i}"
    mc = MarchingCubesTSDFOctree()
    mc.setInputTSDF(vol)
    mc.setMinWeight(0.0)
    mc.setColorByRGB(True)
    mesh = mc.reconstruct()
    assert_same_f32(mesh["vertices"], gold["mc_verts"], "mesh")
    assert np.array_equal(mesh["rgb"], gold["mc_rgb"])
    cloud, rgb = vol.renderColoredView(gold["view_pose"], 1)
    assert_same_f32(cloud[..., :6], gold["view"], "renderColoredView cloud")
    assert np.array_equal(rgb, gold["view_rgb"]) and (gold["view_rgb"] > 0).sum() > 100
    vol.close()


@pytest.mark.parametrize("order", [0, 1])
def test_rgbn_matches_the_oracle_on_random_colours(gpu, order):
    vol, sc = make_volume(64, 96, 72, color=True, order=order, max_weight=3.0)
    vol.setColorMode("RGBNormalized")
    vol.reset()
    ov = OracleVolume(vol._p)
    rng = np.random.RandomState(11 + order)
    for i in range(6):
        tr = synth.turntable_pose(i, 6, sc.size, tilt=0.1 * i)
        col = rng.randint(0, 256, (72, 96, 4)).astype(np.uint8)
        col[rng.rand(72, 96) < 0.02, :3] = 0
        dep = sc.depth(tr, noise_seed=50 + i)
        n_gpu = vol.integrateCloud(dep, col, tr, count=True)
        n_cpu = ov.integrate_rgbn(dep, col, synth.cam_from_vol_f32(tr))
        assert n_gpu == n_cpu
    d, w, rgb = vol.download()
    assert_same_f32(d, ov.d, "d")
    assert np.array_equal(w, ov.w) and w.max() == 3.0
    assert np.array_equal(rgb, ov.rgb)
    assert np.isnan(ov.cn[0]).sum() > 100
    vol.close()


def test_rgbn_slab_handle_and_pipelined_entry_points(gpu):
    """A Z-slab handle and the host entry points (sync, pipelined) reach the same kernel."""
    full, sc = make_volume(64, 96, 72, color=True)
    full.setColorMode("RGBNormalized")
    full.reset()
    slab, _ = make_volume(64, 96, 72, color=True)
    slab.setColorMode("RGBNormalized")
    slab.setZSlab(20, 41, halo=2)
    slab.reset()
    for i in range(4):
        tr = synth.turntable_pose(i, 8, sc.size)
        full.integrateCloud(sc.depth(tr), sc.bgra(i), tr)
        slab.integrateCloud(sc.depth(tr), sc.bgra(i), tr, pipelined=True)
    slab.synchronize()
    d, w, rgb = full.download()
    ds, ws, cs = slab.download(z0=20, nz=21)
    assert_same_f32(ds, d[20:41], "slab d")
    assert np.array_equal(ws, w[20:41]) and np.array_equal(cs, rgb[20:41]) and cs.max() > 0
    halo = slab.download(z0=18, nz=2)
    assert (halo[1] == 0).all()      # halo planes are not integrated
    full.close()
    slab.close()


def test_rgbn_refusals(gpu, tmp_path):
    vol, sc = make_volume(32, 80, 60, color=True)
    vol.setColorMode("RGBNormalized")
    vol.setLayout(capi.LAYOUT_PACKED)
    with pytest.raises(capi.TsdfHipError) as e:
        vol.reset()
    assert e.value.code == capi.E_UNSUPPORTED
    vol.setLayout(capi.LAYOUT_AUTO)
    vol.reset()
    d, w, rgb = vol.download()
    with pytest.raises(capi.TsdfHipError) as e:     # the colour STATE is four floats: r,g,b bytes cannot set it
        vol.upload(d, w, rgb)
    assert e.value.code == capi.E_UNSUPPORTED
    vol.upload(d, w)
    with pytest.raises(capi.TsdfHipError) as e:     # nor does the reference have a usable file form for it
        vol.save(str(tmp_path / "x.vol"))
    assert e.value.code == capi.E_UNSUPPORTED
    with pytest.raises(ValueError):
        vol.setColorMode("HSV")
    vol.close()
