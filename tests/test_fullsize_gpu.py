"""GPU tier: BASELINE.json's full sizes.  A 2048^3 (69 GB) volume cannot be compared voxel by voxel on the host,
but the oracle can own any few planes of it (oracle.SlabOracle): the full-size GPU volume is integrated, sampled
plane groups are downloaded and must equal the oracle bit for bit -- this is real parity at the headline size
(64-bit plane offsets, 2048-wide rows, 2048 x 64 x 8 blocks), not a scaled-down stand-in.  Likewise one Z-slab of
BASELINE configs[4] (4096^3 grid, 1280x960 frames): 4096-wide rows, four x-chunks per row."""
import numpy as np
import pytest
import torch

from cpu_tsdf_amd import capi, synth
from cpu_tsdf_amd.volume import MarchingCubesTSDFOctree, TSDFVolumeOctree
from oracle.oracle import SlabOracle
from tests.common import assert_same_f32

pytestmark = pytest.mark.gpu


def free_gb():
    return torch.cuda.mem_get_info()[0] / 2 ** 30


def configure(vol, res3, size3, W, H, color, zb=0, ze=0):
    sc = synth.Scene(size3[0], W, H)
    vol.setResolution(*res3)
    vol.setGridSize(*size3)
    vol.setImageSize(W, H)
    vol.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    vol.setSensorDistanceBounds(0.0, 3.0 * max(size3))
    vol.setIntegrateColor(color)
    if ze:
        vol.setZSlab(zb, ze)
    return sc


@pytest.mark.parametrize("color", [True, False])
def test_2048_cubed_sampled_planes_match_oracle(gpu, color):
    if free_gb() < 80:
        pytest.skip("needs ~70 GB of free HBM")
    res, W, H = 2048, 640, 480
    vol = TSDFVolumeOctree()
    sc = configure(vol, (res,) * 3, (res * 2.0 ** -8,) * 3, W, H, color)
    vol.reset()
    assert vol.getLayout() == capi.LAYOUT_PACKED
    groups = [(0, 2), (777, 779), (1023, 1025), (2046, 2048)]
    oracles = [SlabOracle(vol._p, a, b) for a, b in groups]
    n_gpu = []
    for i in range(3):
        tr = synth.turntable_pose(i, 44, sc.size)
        dep, col = sc.depth(tr), sc.bgra(i)
        n_gpu.append(vol.integrateCloud(dep, col if color else None, tr, count=True))
        for o in oracles:
            o.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr))
    assert min(n_gpu) > 5e9  # ~72 % of 8.6 G voxels observed per frame
    for (a, b), o in zip(groups, oracles):
        d, w, rgb = vol.download(z0=a, nz=b - a)
        assert_same_f32(d, o.d, f"d planes {a}:{b}")
        assert np.array_equal(w, o.w), f"w planes {a}:{b}"
        if color:
            assert np.array_equal(rgb, o.rgb), f"rgb planes {a}:{b}"
        if 100 < a < 1900:  # the outermost planes lie outside the scene's box: nothing is observed there
            assert (w > 0).mean() > 0.3
    # the mesh of the full grid lies on the analytic surfaces (sphere r = S/4, box faces 0.47 S) to within the
    # depth image's resolution: one 640x480 pixel is ~3 cm (8 voxels) wide at the 17 m viewing distance, so the
    # fused surface is a staircase of that pitch
    mc = MarchingCubesTSDFOctree()
    mc.setInputTSDF(vol)
    mc.setMinWeight(2.0)
    mesh = mc.reconstruct()
    v = mesh["vertices"]
    assert len(v) > 3 * 10 ** 7
    r = np.linalg.norm(v[::97].astype(np.float64), axis=1)
    box = np.abs(np.abs(v[::97]).max(1) - sc.h)
    resid = np.minimum(np.abs(r - sc.r), box)
    assert np.median(resid) < 0.03 and np.quantile(resid, 0.99) < 0.25, (np.median(resid), np.quantile(resid, 0.99))
    vol.close()


def test_4096_grid_slab_with_1280x960_frames_matches_oracle(gpu):
    """BASELINE configs[4]: one rank's view of the 4096^3 grid (a Z-slab; 6 planes here) with 1280x960 frames."""
    res3, W, H = (4096, 4096, 4096), 1280, 960
    size3 = tuple(r * 2.0 ** -8 for r in res3)
    zb, ze = 2045, 2051
    vol = TSDFVolumeOctree()
    sc = configure(vol, res3, size3, W, H, True, zb, ze)
    vol.reset()
    o = SlabOracle(vol._p, zb, ze)
    for i in range(2):
        tr = synth.turntable_pose(i, 16, sc.size)
        dep, col = sc.depth(tr), sc.bgra(i)
        n = vol.integrateCloud(dep, col, tr, count=True)
        assert n == o.integrate(dep, col, synth.cam_from_vol_f32(tr))
    d, w, rgb = vol.download()
    assert_same_f32(d, o.d, "d")
    assert np.array_equal(w, o.w) and np.array_equal(rgb, o.rgb)
    assert (w > 0).mean() > 0.3
    vol.close()


def test_plane_placement_is_probed_for_large_volumes_only(gpu):
    """tsdf_hip_create keeps the fastest of up to `alloc_tries` placements of a >= 4 GiB volume's planes and says what it
    did (tsdf_hip_alloc_probe); small volumes and alloc_tries = 1 allocate once."""
    import ctypes as C

    from cpu_tsdf_amd import capi
    from tests.common import make_volume

    def probe(vol):
        ms, chosen = (C.c_float * 4)(), C.c_int32(-1)
        n = capi.load().tsdf_hip_alloc_probe(vol._need(), ms, C.byref(chosen))
        return n, list(ms), chosen.value
    big, _ = make_volume(1024, color=True)      # 8 GiB of planes
    big.reset()
    n, ms, chosen = probe(big)
    assert n == 3 and 0 <= chosen < 3 and all(m > 0 for m in ms[:3]) and ms[chosen] == min(ms[:3])
    big.close()
    capi.check(capi.load().tsdf_hip_set_tuning(b"alloc_tries", 1), "tuning")
    try:
        once, _ = make_volume(1024, color=True)
        once.reset()
        assert probe(once)[0] == 1
        once.close()
    finally:
        capi.check(capi.load().tsdf_hip_set_tuning(b"alloc_tries", 3), "tuning")
    small, _ = make_volume(128, color=True)
    small.reset()
    assert probe(small)[0] == 1
    small.close()
