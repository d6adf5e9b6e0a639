"""GPU tier: renderView, getFxn/getGradient/getHessian and marching cubes through the C ABI vs the CPU
oracle and vs the committed golden outputs of the reference (tests/golden/reference_32.npz).

Bar: marching-cubes case indices / topology / triangle order exact (north_star); positions, normals,
sampled values: bit equality with the oracle (the kernels replay the same fp32/fp64 operation order)."""
import ctypes as C
import os

import numpy as np
import pytest

from cpu_tsdf_amd import capi, synth
from cpu_tsdf_amd.volume import MarchingCubesTSDFOctree, TSDFVolumeOctree
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32, frames, make_volume

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "reference_32.npz")


@pytest.fixture(scope="module")
def fused64(gpu):
    vol, sc = make_volume(64, color=True)
    vol.reset()
    ov = OracleVolume(vol._p)
    for i, tr, dep, col in frames(sc, 6, 8):
        vol.integrateCloud(dep, col, tr)
        ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
    return vol, ov, sc


VIEWS = [lambda s: synth.turntable_pose(1, 8, s), lambda s: synth.turntable_pose(3, 16, s, tilt=0.5),
         lambda s: synth.look_at_pose((0.05, 0.02, -0.3)), lambda s: synth.look_at_pose((0.0, 0.0, -0.05), target=(0.01, 0.0, 0.2))]


@pytest.mark.parametrize("view", range(len(VIEWS)))
@pytest.mark.parametrize("ds", [1, 2])
def test_raycast_matches_oracle(fused64, view, ds):
    vol, ov, sc = fused64
    tr = VIEWS[view](sc.size)
    got = vol.renderView(tr, ds, camera_frame=False)
    want = ov.raycast(tr, ds)
    assert got.shape == want.shape
    assert np.isfinite(want[..., 0]).sum() > 100
    assert_same_f32(got, want, "renderView xyz/normal/t/iterations")


@pytest.mark.parametrize("size3,res3", [((0.25, 0.25, 0.25), (64, 64, 64)), ((3.3, 3.3, 3.3), (128, 128, 128)),
                                         ((3.0, 12.0, 0.7), (256, 64, 32)), ((1.0, 1.0, 1.0), (50, 37, 41))])
def test_containing_voxel_lookup_equals_octree_descent(gpu, size3, res3):
    """The raycast kernel's cell lookup == OctreeNode::getContainingVoxel's level-by-level walk (oracle
    restatement, octree.cpp:112-121) on random points, on every node centre and its float neighbours (the strict
    `(x - c) > 0` tie rule), on the volume's faces, and on NaN / infinities.  Includes non-dyadic sizes, whose
    centres carry rounding, and non-power-of-two grids (closed-form fallback)."""
    import ctypes as C
    from cpu_tsdf_amd import capi
    from oracle import oracle as O
    vol, _ = make_volume(64, res3=res3, size3=size3)
    vol.reset()
    p = O.params_from(vol._p)
    rng = np.random.RandomState(3)
    pts = [rng.uniform(-0.55, 0.55, (200000, 3)) * np.array(size3)]
    for a in range(3):  # every centre of every level (they are the decision boundaries) +- 1, 2 ulp
        dyadic = [k * size3[a] / 2 ** l - size3[a] / 2 for l in range(1, 9) for k in range(2 ** l + 1)]
        c = np.unique(np.concatenate([vol.centers(a), np.array(dyadic, np.float32)]).astype(np.float32))
        near = np.concatenate([c, np.nextafter(c, np.float32(np.inf)), np.nextafter(c, np.float32(-np.inf)),
                               np.nextafter(np.nextafter(c, np.float32(np.inf)), np.float32(np.inf)),
                               np.float32(size3[a] / 2) * np.array([1, -1], np.float32)])
        q = rng.uniform(-0.5, 0.5, (len(near), 3)) * np.array(size3)
        q[:, a] = near
        pts.append(q)
    pts.append(np.array([[np.nan, 0, 0], [0, np.nan, 0], [0, 0, np.nan], [np.inf, 0, 0], [0, -np.inf, 0], [0, 0, 0],
                         [-0.0, 0.0, -0.0]]))
    xyz = np.ascontiguousarray(np.concatenate(pts), dtype=np.float32)
    n = len(xyz)
    got = np.empty((n, 3), np.int32)
    capi.check(capi.load().tsdf_hip_selftest_containing(vol._need(), capi.as_f32p(xyz), n,
                                                        got.ctypes.data_as(C.POINTER(C.c_int32))), "selftest_containing")
    want = np.full((n, 3), -1, np.int32)
    idx = (C.c_int * 3)()
    L = O.lib()
    for t in range(n):
        if L.oracle_containing(C.byref(p), float(xyz[t, 0]), float(xyz[t, 1]), float(xyz[t, 2]), idx):
            want[t] = idx[0], idx[1], idx[2]
    bad = (got != want).any(1)
    assert not bad.any(), (int(bad.sum()), xyz[bad][:5], got[bad][:5], want[bad][:5])
    assert (want >= 0).all(1).mean() > 0.7
    vol.close()


def test_raycast_non_dyadic_volume(gpu):
    """renderView on a 3.3 m volume (octree centres are rounded sums, not exact dyadics)."""
    vol, sc = make_volume(64, color=True, size=3.3, zmax=12.0, trunc=(0.2, 0.2))
    vol.reset()
    ov = OracleVolume(vol._p)
    for i, tr, dep, col in frames(sc, 4, 8):
        vol.integrateCloud(dep, col, tr)
        ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
    for tr in (synth.turntable_pose(1, 8, sc.size), synth.look_at_pose((0.5, 0.2, -3.0)), synth.look_at_pose((0.0, 0.0, -0.6))):
        got = vol.renderView(tr, 1, camera_frame=False)
        want = ov.raycast(tr, 1)
        assert_same_f32(got, want, "renderView, 3.3 m volume")
        assert np.isfinite(want[..., 0]).sum() > 300


def test_raycast_camera_frame_and_misses(fused64):
    vol, ov, sc = fused64
    tr = synth.turntable_pose(2, 8, sc.size)
    cam = vol.renderView(tr, 1)  # the reference returns camera-frame points (:422); transform done in the kernel
    from cpu_tsdf_amd.volume import transform_cloud_with_normals
    host = transform_cloud_with_normals(vol.renderView(tr, 1, camera_frame=False), synth.eigen_affine_inverse(tr))
    assert_same_f32(cam, host, "in-kernel transformPointCloudWithNormals == the host restatement")
    hit = np.isfinite(cam[..., 0])
    assert 0 < hit.sum() < hit.size
    # a hit's camera-frame z is the depth the sensor would have seen; compare with the analytic scene
    dep = sc.depth(tr)
    both = hit & np.isfinite(dep)  # rays the sensor model also returns a depth for
    err = np.abs(cam[..., 2][both] - dep[both])
    assert both.sum() > 500 and np.median(err) < 0.05  # within ~the truncation band of the true surface
    # a miss keeps PointNormal's default normal (0,0,0); a degenerate crossing (t* = NaN) has NaN normals
    nm = cam[..., 3:6][~hit]
    assert ((nm == 0).all(1) | np.isnan(nm).all(1)).all() and (nm == 0).all(1).sum() > 1000


def test_sample_matches_oracle(fused64):
    vol, ov, sc = fused64
    rng = np.random.RandomState(3)
    pts = rng.uniform(-0.14, 0.14, (20000, 3)).astype(np.float32)
    pts[:8] = [[0, 0, 0], [0.125, 0, 0], [-0.125, 0, 0], [0.1230469, 0, 0], [5, 0, 0], [np.nan, 0, 0],
               [-0.1249, -0.1249, -0.1249], [0.00195312, 0.00195312, 0.00195312]]
    ok, val, grad, hess = vol.sample(pts)
    ok2, val2, grad2, hess2 = ov.sample(pts)
    assert np.array_equal(ok, ok2) and 0 < ok.sum() < len(ok)
    assert_same_f32(val, val2, "getFxn")
    assert_same_f32(grad, grad2, "getGradient")
    assert_same_f32(hess, hess2, "getHessian")


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("wmin", [0.0, 2.0, 2.5])
def test_marching_cubes_matches_oracle(fused64, mode, wmin):
    vol, ov, sc = fused64
    mc = MarchingCubesTSDFOctree()
    mc.setInputTSDF(vol)
    mc.setMinWeight(wmin)
    mc.setColorByRGB(mode == 1)
    mc.setColorByConfidence(mode == 2)
    mesh = mc.reconstruct(want_cells=True)
    verts, rgb, cells = ov.march(wmin, mode)
    assert len(verts) > 30000
    assert np.array_equal(mesh["cells"], cells), "cell sequence (topology + reference triangle order)"
    assert_same_f32(mesh["vertices"], verts, "triangle vertices")
    assert np.array_equal(mesh["polygons"].ravel(), np.arange(len(verts)))
    if mode:
        assert np.array_equal(mesh["rgb"], rgb)


def _mesh_vs_oracle(vol, ov, wmin, mode):
    mc = MarchingCubesTSDFOctree()
    mc.setInputTSDF(vol)
    mc.setMinWeight(wmin)
    mc.setColorByRGB(mode == 1)
    mc.setColorByConfidence(mode == 2)
    mesh = mc.reconstruct(want_cells=True)
    verts, rgb, cells = ov.march(wmin, mode)
    assert np.array_equal(mesh["cells"], cells), "cell sequence (topology + reference triangle order)"
    assert_same_f32(mesh["vertices"], verts, "triangle vertices")
    if mode:
        assert np.array_equal(mesh["rgb"], rgb)
    return len(verts)


def test_marching_cubes_odd_resolutions(gpu):
    # nx not a multiple of 4 (pitch > nx), three different non-power-of-two axes: the quad classify must
    # not emit cells in the padding nor lose the last cells of a row
    for res3 in [(50, 37, 41), (33, 64, 23), (7, 5, 6), (3, 3, 3), (4, 3, 9)]:
        vol, sc = make_volume(64, color=True, res3=res3, size3=(0.25, 0.25, 0.25))
        vol.reset()
        ov = OracleVolume(vol._p)
        for i, tr, dep, col in frames(sc, 3, 8):
            vol.integrateCloud(dep, col, tr)
            ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
        n = _mesh_vs_oracle(vol, ov, 0.0, 1)
        if min(res3) > 20:
            assert n > 1000


def test_marching_cubes_wide_rows(gpu):
    """Rows of >= 1024 voxels make a block one row high (TY = 1, several x-chunks per row), the shape of the
    BASELINE grids.  Exact parity on such grids, including a row count that is not a multiple of 4 and nx that is
    not a multiple of 1024.  (A variant that fetched the bundle rows of four consecutive cell rows together and
    took x+1 neighbours from the next lane passed this test too but was slower -- 14.0 vs 12.0 ms at 2048^3, 128
    VGPRs -- and was dropped.)"""
    for res3, size3 in [((1024, 24, 12), (4.0, 0.09375, 0.046875)), ((1100, 10, 7), (1.1, 0.01, 0.007))]:
        vol, sc = make_volume(64, 160, 120, color=True, res3=res3, size3=size3, zmax=30.0, trunc=(0.01, 0.01))
        vol.reset()
        ov = OracleVolume(vol._p)
        D = 1.2 * size3[0]  # far enough for the 62 deg FOV to span the slab's width
        tr = synth.look_at_pose((0.0, 0.0, -D))
        u = np.arange(160, dtype=np.float32)[None, :]
        v = np.arange(120, dtype=np.float32)[:, None]
        for k in range(3):  # a gently tilted, rippled sheet through the slab: a surface cell in every column
            dep = (np.float32(D) + np.float32(0.00004 * (k + 1)) * (u - 80) + np.float32(0.0003) * np.sin(v * 0.7 + k))
            dep = np.ascontiguousarray(np.broadcast_to(dep.astype(np.float32), (120, 160)))
            col = sc.bgra(k)
            vol.integrateCloud(dep, col, tr)
            ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
        n = _mesh_vs_oracle(vol, ov, 0.0, 1)
        assert n > 30000, n
        _mesh_vs_oracle(vol, ov, 2.0, 0)
        vol.close()


def test_marching_cubes_wave_list_flush_paths(gpu):
    """The classify kernel's wave-private LDS list flushes mid-block only on dense surfaces; force a flush after
    every append (tuning knob mc_flush_at = 0) and taller blocks, and compare with the oracle."""
    from cpu_tsdf_amd import capi
    try:
        capi.set_tuning("mc_flush_at", 0)
        capi.set_tuning("rows_per_block", 64)
        vol, sc = make_volume(64, color=True)
        vol.reset()
        ov = OracleVolume(vol._p)
        for i, tr, dep, col in frames(sc, 4, 8, noise=True):
            vol.integrateCloud(dep, col, tr)
            ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
        assert _mesh_vs_oracle(vol, ov, 0.0, 1) > 30000
        _mesh_vs_oracle(vol, ov, 2.0, 0)
    finally:
        capi.set_tuning("mc_flush_at", 512)
        capi.set_tuning("rows_per_block", 64)


def test_marching_cubes_empty_and_global_transform(gpu):
    vol, sc = make_volume(32)
    vol.reset()
    mc = MarchingCubesTSDFOctree()
    mc.setInputTSDF(vol)
    assert len(mc.reconstruct()["vertices"]) == 0  # nothing integrated yet
    tr = synth.turntable_pose(0, 8, sc.size)
    vol.integrateCloud(sc.depth(tr), None, tr)
    mc.setMinWeight(0)
    base = mc.reconstruct()["vertices"]
    assert len(base) > 100
    g = synth.look_at_pose((0.3, 0.1, -0.2))
    vol.setGlobalTransform(g)
    moved = mc.reconstruct()["vertices"]
    want = (base.astype(np.float64) @ g[:3, :3].T + g[:3, 3]).astype(np.float32)
    assert np.max(np.abs(moved - want)) < 1e-6


def test_everything_matches_reference_golden(gpu):
    """The HIP path reproduces outputs of the reference's own code (generated by
    tests/golden/make_golden.py from oracle/_ref) -- no oracle in the loop."""
    from cpu_tsdf_amd.volume import transform_cloud_with_normals
    g = np.load(GOLD)
    res, W, H, size = int(g["res"]), int(g["width"]), int(g["height"]), float(g["size"])
    vol, sc = make_volume(res, W, H, color=True)
    assert sc.size == size
    vol.reset()
    for i in range(int(g["n_frames"])):
        tr = synth.turntable_pose(i, int(g["total"]), size)
        vol.integrateCloud(sc.depth(tr), sc.bgra(i), tr)
        d, w, rgb = vol.download()
        assert_same_f32(d, g[f"d{i}"], f"d after frame {i}")
        assert np.array_equal(w, g[f"w{i}"].astype(np.float32)) and np.array_equal(rgb, g[f"rgb{i}"])
    for k, tr in enumerate(g["view_poses"]):
        assert_same_f32(vol.renderView(tr, 1)[..., :6], g[f"view{k}"], f"renderView pose {k}")
    assert_same_f32(vol.renderView(g["view_poses"][0], 2)[..., :6], g["view0_ds2"], "renderView ds 2")
    ok, val, grad, hess = vol.sample(g["sample_pts"])
    gok = g["sample_ok"] == 7
    assert np.array_equal(ok, gok)
    assert_same_f32(val[ok], g["sample_val"][ok], "getFxn")
    assert_same_f32(grad[ok], g["sample_grad"][ok], "getGradient")
    assert_same_f32(hess[ok], g["sample_hess"][ok], "getHessian")
    for mode in (0, 1, 2):
        for wmin in (0.0, 2.0):
            mc = MarchingCubesTSDFOctree()
            mc.setInputTSDF(vol)
            mc.setMinWeight(wmin)
            mc.setColorByRGB(mode == 1)
            mc.setColorByConfidence(mode == 2)
            mesh = mc.reconstruct()
            key = f"mc_m{mode}_w{int(wmin)}"
            assert_same_f32(mesh["vertices"], g[key + "_verts"], key)
            if mode:
                assert np.array_equal(mesh["rgb"], g[key + "_rgb"])


def test_raycast_and_mesh_at_512_properties(gpu):
    """Size-independent checks at a grid too big for the CPU oracle to be quick: every hit of a fused
    analytic scene lies within a voxel of the true surface; the mesh is closed around the sphere."""
    vol, sc = make_volume(512, width=320, height=240)
    vol.setZSlab(128, 384, halo=0)
    vol.reset()
    for i in range(4):
        tr = synth.turntable_pose(i, 4, sc.size)
        vol.integrateCloud(sc.depth(tr), None, tr)
    mc = MarchingCubesTSDFOctree()
    mc.setInputTSDF(vol)
    mc.setMinWeight(1.0)
    with pytest.raises(Exception):
        mc.reconstruct()  # slab without halo cannot mesh its top cells: must refuse loudly
    full, _ = make_volume(256, width=320, height=240)
    full.reset()
    sc = synth.scene_a(256, 320, 240)
    for i in range(6):
        tr = synth.turntable_pose(i, 6, sc.size)
        full.integrateCloud(sc.depth(tr), None, tr)
    tr = synth.turntable_pose(0, 6, sc.size)
    cam = full.renderView(tr, 1)
    hit = np.isfinite(cam[..., 0])
    dep = sc.depth(tr)
    both = hit & np.isfinite(dep)
    assert both.sum() > 5000
    assert np.percentile(np.abs(cam[..., 2][both] - dep[both]), 90) < 0.06  # 2x the truncation distance
    mc.setInputTSDF(full)
    mesh = mc.reconstruct()
    v = mesh["vertices"]
    r = np.linalg.norm(v, axis=1)
    on_sphere = np.abs(r - sc.r) < 3 * sc.size / 256
    assert on_sphere.sum() > 10000


@pytest.mark.parametrize("color", [False, True])
def test_marching_cubes_on_a_random_volume(gpu, color):
    """Every cube configuration: random distances (3 % beyond +-1), random weights (some zero or under the threshold)
    and colours uploaded into the HIP volume; mesh == the oracle's (which tests/test_oracle_golden.py pins to the
    reference on the same kind of volume), triangle for triangle."""
    res = 48
    vol, sc = make_volume(res, 80, 60, color=color)
    vol.reset()
    rng = np.random.RandomState(40 + color)
    ov = OracleVolume(vol._p)
    ov.d[:] = rng.uniform(-1.03, 1.03, ov.d.shape).astype(np.float32)
    ov.w[:] = np.where(rng.rand(*ov.w.shape) < 0.04, rng.randint(0, 2, ov.w.shape), rng.randint(2, 4, ov.w.shape)).astype(np.float32)
    if color:
        ov.rgb[:] = rng.randint(0, 256, ov.rgb.shape)
    vol.upload(ov.d, ov.w, ov.rgb if color else None)
    for w_min, mode in [(1.5, 1 if color else 0), (0.5, 2)]:
        mc = MarchingCubesTSDFOctree()
        mc.setInputTSDF(vol)
        mc.setMinWeight(w_min)
        mc.setColorByRGB(mode == 1)
        mc.setColorByConfidence(mode == 2)
        mesh = mc.reconstruct(want_cells=True)
        v, c, cells = ov.march(w_min, mode)
        assert len(cells) > 20000
        assert np.array_equal(mesh["cells"], cells)
        assert_same_f32(mesh["vertices"], v, f"random volume, w_min {w_min}")
        if mode:
            assert np.array_equal(mesh["rgb"], c)
    vol.close()


def test_raycast_and_sampling_on_a_random_volume(gpu):
    """k_raycast / k_sample on a volume of random distances and weights (uploaded): erratic steps, sign changes
    everywhere, refinement walks, t_star extrapolation on nearly equal samples, mostly invalid normals -- == the oracle
    (pinned to the reference on the same kind of volume in tests/test_oracle_golden.py), float for float."""
    res, W, H = 32, 80, 60
    vol, sc = make_volume(res, W, H)
    vol.reset()
    rng = np.random.RandomState(99)
    ov = OracleVolume(vol._p)
    z, y, x = np.meshgrid(*[np.linspace(-1, 1, res)] * 3, indexing="ij")
    ov.d[:] = np.clip(0.9 * np.sin(3 * x + 1) * np.cos(2 * y) + 0.5 * z + rng.normal(0, 0.15, ov.d.shape), -1, 1).astype(np.float32)
    ov.w[:] = np.where(rng.rand(*ov.w.shape) < 0.1, 0, rng.randint(1, 5, ov.w.shape)).astype(np.float32)
    vol.upload(ov.d, ov.w)
    hits = 0
    for k in range(6):
        eye = rng.uniform(-2.2, 2.2, 3) * sc.size * (1.0 if k % 3 else 0.25)
        tr = synth.look_at_pose(eye, target=rng.uniform(-0.2, 0.2, 3) * sc.size)
        ds = 1 + (k == 4)
        got = vol.renderView(tr, ds, camera_frame=False)
        assert_same_f32(got, ov.raycast(tr, ds), f"renderView pose {k}")
        hits += int(np.isfinite(got[..., 0]).sum())
    assert hits > 3000
    pts = rng.uniform(-0.07, 0.07, (2000, 3)).astype(np.float32)
    ok, val, grad, hess = vol.sample(pts)
    ok2, val2, grad2, hess2 = ov.sample(pts)
    assert np.array_equal(ok, ok2) and ok.sum() > 200
    assert_same_f32(val[ok], val2[ok], "getFxn")
    assert_same_f32(grad[ok], grad2[ok], "getGradient")
    assert_same_f32(hess[ok], hess2[ok], "getHessian")
    vol.close()


def test_pinned_caller_memory_takes_the_direct_dma_path(gpu):
    """Host buffers from tsdf_hip_host_alloc are detected (hipPointerGetAttributes) and written by DMA without the
    bounce buffer: same bytes as the pageable path, for renderView, block download and the mesh."""
    vol, sc = make_volume(64, color=True)
    vol.reset()
    ov = OracleVolume(vol._p)
    for i, tr, dep, col in frames(sc, 4, 8):
        vol.integrateCloud(dep, col, tr)
        ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
    tr = synth.turntable_pose(1, 8, sc.size)
    a = vol.renderView(tr, 1)
    b = vol.renderView(tr, 1, pinned=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.isfinite(a[..., 0]).sum() > 100
    pd = capi.PinnedArray((64, 64, 64), np.float32)
    pw = capi.PinnedArray((64, 64, 64), np.float32)
    capi.check(capi.load().tsdf_hip_download(vol._need(), 0, 0, 0, 64, 64, 64, capi.as_f32p(pd.array), capi.as_f32p(pw.array), None),
               "download into pinned memory")
    assert np.array_equal(pd.array.view(np.uint32), ov.d.view(np.uint32)) and np.array_equal(pw.array, ov.w)
    up = capi.PinnedArray((64, 64, 64), np.float32)  # and the other direction: upload FROM pinned memory
    up.array[:] = ov.d[::-1]
    capi.check(capi.load().tsdf_hip_upload(vol._need(), 0, 0, 0, 64, 64, 64, capi.as_f32p(up.array), None, None), "upload")
    assert np.array_equal(vol.download()[0], ov.d[::-1])
    for x in (pd, pw, up):
        x.free()
    vol.close()


def _mesh(vol, wmin=2.0, mode=1):
    mc = MarchingCubesTSDFOctree()
    mc.setInputTSDF(vol)
    mc.setMinWeight(wmin)
    mc.setColorByRGB(mode == 1)
    mc.setColorByConfidence(mode == 2)
    return mc.reconstruct(want_cells=True)


def _same_mesh(a, b, what):
    assert np.array_equal(a["cells"], b["cells"]), what
    assert_same_f32(a["vertices"], b["vertices"], what)
    if a["rgb"] is not None:
        assert np.array_equal(a["rgb"], b["rgb"]), what


@pytest.mark.parametrize("res3,trunc,pose_kind", [((128, 128, 128), (0.03, 0.03), "turntable"), ((200, 72, 150), (0.02, 0.05), "turntable"),
                                                  ((130, 128, 128), (0.03, 0.03), "inside"), ((320, 96, 64), (0.06, 0.03), "turntable"),
                                                  ((1024, 44, 48), (0.03, 0.04), "turntable"), ((2048, 24, 20), (0.03, 0.03), "inside")])
def test_marching_cubes_skips_what_no_band_observation_is_near(gpu, res3, trunc, pose_kind):
    """k_mc_classify reads only what the integrate kernels' "band seen" flags are near (cells of 64 x 4 x 1 voxels, grown
    by one voxel: a triangle needs a negative corner, a negative distance needs an observation inside the truncation
    band).  With the skip on and off the mesh is the same and equals the oracle's: cubic / flat / wide grids (several
    x-chunks, rows that are no multiple of 4 or 64), a hinge value below 1 (pos < neg: free space is INSIDE the band),
    launches restricted to a sub-box of the grid (camera inside the volume: flag coordinates offset), noise; rows of 1024
    and 2048 voxels, whose flag rows are whole 16-byte groups (k_mc_need_rows instead of k_mc_need)."""
    S = max(res3) * 2.0 ** -8
    vol, sc = make_volume(64, color=True, trunc=trunc, size=S, zmax=4 * S, res3=res3, size3=tuple(r * 2.0 ** -8 for r in res3))
    sc.h = np.array([0.47 * r * 2.0 ** -8 for r in res3])
    vol.reset()
    ov = OracleVolume(vol._p)
    for i in range(5):
        tr = synth.turntable_pose(i, 7, S, radius_factor=1.6) if pose_kind == "turntable" else \
            synth.look_at_pose((0.02 * i - 0.04, 0.01, -0.12), target=(0.0, 0.0, 0.3))
        dep, col = sc.depth(tr, noise_seed=31 + i), sc.bgra(i)
        vol.integrateCloud(dep, col, tr)
        ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
    meshes, stats = {}, {}
    try:
        for skip in (1, 0):
            capi.set_tuning("mc_skip", skip)
            meshes[skip] = _mesh(vol)
            st = (C.c_uint64 * 4)()
            capi.check(capi.load().tsdf_hip_march_stats(vol._need(), st), "march_stats")
            stats[skip] = [int(v) for v in st]
    finally:
        capi.set_tuning("mc_skip", 1)
    _same_mesh(meshes[1], meshes[0], "skip on vs off")
    # tsdf_hip_march_stats: cells / triangles as fetched; without the flags every plane is requested (the plane a block
    # of up to 32 cell planes ends on is read again by the next one), with them strictly less
    assert stats[1][3] == 1 and stats[0][3] == 0
    assert stats[1][:2] == stats[0][:2] == [len(np.unique(meshes[1]["cells"])), len(meshes[1]["cells"])]
    nx, ny, nz = res3
    pitch = (nx + 3) // 4 * 4
    planes_read = (nz - 2) + -(-(nz - 2) // 32)  # cell planes 1 .. nz - 2, + one plane per block
    assert 0.9 * ny * pitch * 4 * planes_read < stats[0][2] < 2.0 * ny * pitch * 4 * planes_read  # (+ halo rows / columns, whole 64-voxel groups)
    assert 0 < stats[1][2] <= stats[0][2]
    v2, c2, cells2 = ov.march(2.0, 1)
    assert len(cells2) > 300
    assert np.array_equal(meshes[1]["cells"], cells2)
    assert_same_f32(meshes[1]["vertices"], v2, "mesh vs oracle")
    assert np.array_equal(meshes[1]["rgb"], c2)
    vol.close()


@pytest.mark.parametrize("color", [True, False])
def test_weight_test_is_elided_only_where_no_count_can_fail_it(gpu, color):
    """marching_cubes_tsdf_octree.cpp:98 drops a cell with a corner of w < w_min.  In the PACKED layout, while the planes
    hold only what integrateCloud wrote since the reset, a corner inside the band has been observed at least once, so the
    test cannot fail for w_min <= min(1, max_weight) and the 8 gathers per listed cell are not made (tsdf_hip_march_stats
    out[3] bit 1).  The mesh equals the oracle's -- which evaluates the test on every corner -- with the elision on
    (w_min 0.5, 1), where it must stay off (w_min 1.5, 2: first-seen voxels at the rim of every view DO fail), with the
    flags knob off, after an upload that plants in-band distances with zero weight, and with max_weight below w_min."""
    lib = capi.load()

    def stats(v):
        st = (C.c_uint64 * 4)()
        capi.check(lib.tsdf_hip_march_stats(v._need(), st), "march_stats")
        return [int(x) for x in st]
    vol, sc = make_volume(64, color=color)
    vol.reset()
    assert vol.getLayout() == capi.LAYOUT_PACKED
    ov = OracleVolume(vol._p)
    for i, tr, dep, col in frames(sc, 4, 8, noise=True):
        vol.integrateCloud(dep, col if color else None, tr)
        ov.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr))
    mode = 1 if color else 0
    n = {}
    for wmin, elided in ((0.5, True), (1.0, True), (1.5, False), (2.0, False)):
        n[wmin] = _mesh_vs_oracle(vol, ov, wmin, mode)
        assert bool(stats(vol)[3] & 2) == elided, (wmin, stats(vol))
    assert n[0.5] == n[1.0] > n[2.0] > 1000   # the weight test does bite on this volume where it is evaluated
    try:
        capi.set_tuning("mc_skip", 0)
        assert _mesh_vs_oracle(vol, ov, 1.0, mode) == n[1.0] and stats(vol)[3] == 0
    finally:
        capi.set_tuning("mc_skip", 1)
    # in-band distances nobody observed (weight 0) arrive from outside: the counts prove nothing any more
    d, w, rgb = vol.download()
    z, y, x = np.mgrid[0:64, 0:64, 0:64]
    blob = (np.sqrt((x - 12.0) ** 2 + (y - 50.0) ** 2 + (z - 9.0) ** 2) - 4.0) / 8.0
    region = np.abs(blob) < 0.9
    d2, w2 = d.copy(), w.copy()
    d2[region] = blob[region].astype(np.float32)
    w2[region] = 0.0
    vol.upload(d2, w2, rgb)
    ov2 = OracleVolume(vol._p, adopt=(d2, w2, rgb))
    n_up = _mesh_vs_oracle(vol, ov2, 1.0, mode)   # the sphere's cells fail the test, in the product as in the oracle
    assert stats(vol)[3] == 0
    assert _mesh_vs_oracle(vol, ov2, 0.0, mode) > n_up + 100   # ... and are there without it
    vol.close()
    # max_weight 0.5: every observed weight is 0.5 -- below w_min 1 (test evaluated, nothing left), not below 0.5 (elided)
    vol, sc = make_volume(64, color=color, max_weight=0.5)
    vol.reset()
    if vol.getLayout() == capi.LAYOUT_PACKED:
        ov = OracleVolume(vol._p)
        for i, tr, dep, col in frames(sc, 3, 8):
            vol.integrateCloud(dep, col if color else None, tr)
            ov.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr))
        assert _mesh_vs_oracle(vol, ov, 1.0, mode) == 0 and not stats(vol)[3] & 2
        assert _mesh_vs_oracle(vol, ov, 0.5, mode) > 1000 and stats(vol)[3] & 2
    vol.close()


def test_band_flags_are_dropped_when_voxels_arrive_from_outside(gpu):
    """An upload can put a surface where nothing was ever observed: the flags stop describing the planes and marching
    cubes must read everything again (until the next reset)."""
    vol, sc = make_volume(64, color=False)
    vol.reset()
    for i, tr, dep, col in frames(sc, 3, 8):
        vol.integrateCloud(dep, None, tr)
    before = _mesh(vol, 1.0, 0)
    d, w, _ = vol.download()
    z, y, x = np.mgrid[0:64, 0:64, 0:64]
    blob = (np.sqrt((x - 12.0) ** 2 + (y - 50.0) ** 2 + (z - 9.0) ** 2) - 4.0) / 8.0  # a small sphere in a never-observed corner
    region = np.abs(blob) < 0.9
    d2, w2 = d.copy(), w.copy()
    d2[region] = blob[region].astype(np.float32)
    w2[region] = 3.0
    vol.upload(d2, w2, None)
    ov = OracleVolume(vol._p, adopt=(d2, w2, None))
    after = _mesh(vol, 1.0, 0)
    v2, _, cells2 = ov.march(1.0, 0)
    assert len(cells2) > len(before["cells"]) + 50
    assert np.array_equal(after["cells"], cells2)
    assert_same_f32(after["vertices"], v2, "mesh after upload")
    # ... and exact again after a reset
    vol.reset()
    assert len(_mesh(vol, 1.0, 0)["cells"]) == 0
    for i, tr, dep, col in frames(sc, 3, 8):
        vol.integrateCloud(dep, None, tr)
    _same_mesh(_mesh(vol, 1.0, 0), before, "after reset + the same frames")
    vol.close()
