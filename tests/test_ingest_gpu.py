"""GPU tier: frame ingest (tsdf_hip_organize = the `integrate` program's units / zero->NaN / world->camera /
z-buffer reprojection, src/prog/integrate.cpp:559-618) vs the serial restatement in the oracle, then
integrateCloud on the staged frame vs the oracle fed with the oracle's organised frame."""
import numpy as np
import pytest

from cpu_tsdf_amd import synth
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32, make_volume

pytestmark = pytest.mark.gpu


def unorganised_cloud(sc, trans, seed, units=1.0, world=None, zeros=False):
    """Back-project a synthetic depth image to a shuffled point list with everything the loop has to cope with:
    several points per pixel (incl. exact z ties with different colours), NaN / zero / negative / infinite
    coordinates, points outside the image."""
    rng = np.random.RandomState(seed)
    dep = sc.depth(trans).astype(np.float64)
    v, u = np.nonzero(np.isfinite(dep))
    z = dep[v, u]
    # jitter inside the pixel so that truncation still lands in (u, v)
    uu = u + rng.uniform(0.05, 0.95, u.size)
    vv = v + rng.uniform(0.05, 0.95, v.size)
    pts = np.stack([(uu - sc.cx) / sc.fx * z, (vv - sc.cy) / sc.fy * z, z], 1)
    farther = pts[rng.choice(len(pts), len(pts) // 3)] * rng.uniform(1.0, 1.5, (len(pts) // 3, 1))  # same rays, hidden
    ties = pts[rng.choice(len(pts), len(pts) // 10)].copy()                                           # exact duplicates
    junk = np.array([[np.nan, 0, 1], [0, 0, 0], [0.1, 0.1, -1.0], [np.inf, 0, 2], [5, 5, 0.1], [0, 0, np.nan],
                     [0, 0, np.inf], [1e30, 1e30, 1e-30]], np.float64)
    if zeros:
        junk = np.concatenate([junk, np.zeros((50, 3))])
    allp = np.concatenate([pts, farther, ties, junk])
    if world is not None:  # express the same points in the world frame
        allp = np.where(np.isfinite(allp).all(1, keepdims=True), allp @ world[:3, :3].T + world[:3, 3], allp)
    allp = allp / units
    order = rng.permutation(len(allp))
    xyz = np.zeros((len(allp), 8), np.float32)  # PCL PointXYZRGBA: 8 floats per point
    xyz[:, :3] = allp[order]
    bgra = rng.randint(0, 256, (len(allp), 32)).astype(np.uint8)  # colour bytes every 32 bytes
    return xyz, bgra


@pytest.mark.parametrize("case", ["camera", "units", "world", "zero_nans"])
def test_organize_matches_serial_loop_and_feeds_integrate(gpu, case):
    vol, sc = make_volume(64, 160, 120, color=True)
    vol.reset()
    ov = OracleVolume(vol._p)
    for i in range(3):
        tr = synth.turntable_pose(i, 8, sc.size)
        units = 0.001 if case == "units" else 1.0
        world = synth.look_at_pose((0.3, -0.2, 0.9), target=(0.05, 0.0, 0.1)) if case == "world" else None
        xyz, bgra = unorganised_cloud(sc, tr, 10 + i, units=units, world=world, zeros=(case == "zero_nans"))
        w2c = np.linalg.inv(world) if world is not None else None
        kw = dict(units=units, zero_nans=(case == "zero_nans"), world_to_cam=w2c)
        d_ref, c_ref, n_ref = ov.organize(xyz, bgra[:, :4].copy(), **kw)
        d_gpu, c_gpu, n_gpu = vol.organize(xyz, bgra, cloud_units=units, zero_nans=(case == "zero_nans"), world_to_cam=w2c)
        assert n_ref == n_gpu and n_ref > 5000
        assert_same_f32(d_gpu, d_ref, f"organised depth, frame {i}")
        filled = np.isfinite(d_ref)
        assert np.array_equal(c_gpu[filled], c_ref[filled]), "colour of the z-buffer winners (incl. ties -> first point)"
        assert (c_gpu[~filled] == [0, 0, 0, 255]).all()
        n_obs = vol.integrateStaged(tr, count=True)
        assert n_obs == ov.integrate(d_ref, c_ref, synth.cam_from_vol_f32(tr))
    d, w, rgb = vol.download()
    assert_same_f32(d, ov.d, "d")
    assert np.array_equal(w, ov.w) and np.array_equal(rgb, ov.rgb)


def test_organize_empty_cloud_and_stride3(gpu):
    vol, sc = make_volume(32, 80, 60)
    vol.reset()
    with pytest.raises(Exception):
        vol.integrateStaged(np.eye(4))  # nothing staged yet
    d, c, n = vol.organize(np.zeros((0, 3), np.float32))
    assert n == 0 and np.isnan(d).all()
    pts = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 0.5], [0.0, 0.0, 0.5]], np.float32)  # tightly packed xyz, no colour
    d, c, n = vol.organize(pts)
    assert n == 1 and np.nanmin(d) == np.float32(0.5)
    ov = OracleVolume(vol._p)
    d2, _, n2 = ov.organize(pts)
    assert n2 == 1
    assert_same_f32(d, d2, "depth")
