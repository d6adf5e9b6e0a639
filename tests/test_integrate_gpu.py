"""GPU tier: integrateCloud through the C ABI vs the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): per-voxel (d, w) within 1e-5; we assert bit equality (the kernel
mirrors the oracle's fp32 operation order, no FMA) and report it as such; rgb bytes exact; the
observed-voxel counter exact."""
import numpy as np
import pytest

from cpu_tsdf_amd import capi, synth
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32, frames, make_volume

pytestmark = pytest.mark.gpu
TOL = 1e-5


def run_pair(vol, sc, n_frames, total=None, noise=False):
    vol.reset()
    ov = OracleVolume(vol._p)
    for i, tr, dep, col in frames(sc, n_frames, total, noise):
        n_gpu = vol.integrateCloud(dep, col if vol._p.integrate_color else None, tr, count=True)
        n_cpu = ov.integrate(dep, col if vol._p.integrate_color else None, synth.cam_from_vol_f32(tr))
        assert n_gpu == n_cpu, f"frame {i}: n_observed {n_gpu} != oracle {n_cpu}"
    return ov


def compare(vol, ov):
    d, w, rgb = vol.download()
    assert np.nanmax(np.abs(d - ov.d)) <= TOL and np.max(np.abs(w - ov.w)) <= TOL
    assert_same_f32(d, ov.d, "d")
    assert_same_f32(w, ov.w, "w")
    if ov.rgb is not None:
        assert np.array_equal(rgb, ov.rgb), f"{(rgb != ov.rgb).sum()} rgb bytes differ"
    return d, w


@pytest.mark.parametrize("order", [0, 1])
def test_parity_64_multi_frame(gpu, order):
    vol, sc = make_volume(64, order=order)
    ov = run_pair(vol, sc, 6, total=8)
    d, w = compare(vol, ov)
    assert (w > 0).mean() > 0.5 and (np.abs(d) < 1).sum() > 1000


def test_parity_color_128(gpu):
    vol, sc = make_volume(128, color=True)
    ov = run_pair(vol, sc, 5, total=12)
    compare(vol, ov)
    assert ov.rgb[ov.w > 0].max() > 0


def test_parity_noise_and_weight_saturation(gpu):
    # max_weight 3 -> running mean turns into an EMA after 3 frames (octree.cpp:157-159)
    vol, sc = make_volume(64, max_weight=3.0)
    ov = run_pair(vol, sc, 8, total=8, noise=True)
    d, w = compare(vol, ov)
    assert w.max() == 3.0


def test_parity_sensor_bounds_and_truncation(gpu):
    vol, sc = make_volume(64, zmin=0.5, zmax=0.6, trunc=(0.02, 0.05))
    ov = run_pair(vol, sc, 3, total=8)
    d, w = compare(vol, ov)
    assert 0 < (w > 0).mean() < 0.97  # the bounds really clip
    assert d.max() == pytest.approx(0.02 / 0.05, rel=1e-6)


@pytest.mark.parametrize("trunc", [(0.03, 0.03), (0.06, 0.03), (0.1, 0.03), (0.02, 0.05), (0.0301, 0.0299)])
@pytest.mark.parametrize("color", [False, True])
def test_free_space_voxels_resting_at_the_hinge(gpu, trunc, color):
    """Waves whose observed voxels all sit at the hinge value p = pos/neg and see free space again skip the d update
    when the host has checked (p*w + p)/(w + 1) == p for every weight; hinge values for which that identity holds
    (1, 2) and for which it may not (3.33.., 0.4, 1.0067) must all equal the oracle, frame after frame from the same
    pose (every free-space voxel re-observed at the hinge) and through weight saturation."""
    vol, sc = make_volume(64, color=color, trunc=trunc, max_weight=4.0)
    vol.reset()
    ov = OracleVolume(vol._p)
    for i in range(7):
        tr = synth.turntable_pose(i // 3, 8, sc.size)      # three frames per pose
        dep, col = sc.depth(tr, noise_seed=5 + i), sc.bgra(i)
        n_gpu = vol.integrateCloud(dep, col if color else None, tr, count=True)
        assert n_gpu == ov.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr))
        compare(vol, ov)
    p = np.float32(trunc[0]) / np.float32(trunc[1])
    assert (ov.d == p).mean() > 0.02
    vol.close()


def test_parity_default_grid_3m_512_nondyadic(gpu):
    # reference defaults: 3 m / 512 (voxel 3*2^-9), 640x480 f=525, sensor 0.3..3 m; 64-plane slab
    from cpu_tsdf_amd.volume import TSDFVolumeOctree
    vol = TSDFVolumeOctree()
    vol.setZSlab(224, 288)
    vol.reset()
    sc = synth.Scene(3.0)
    ov = OracleVolume(vol._p)
    for i in range(2):
        tr = synth.look_at_pose((0.3 * i, -0.1, -2.0))
        dep = sc.depth(tr)
        n_gpu = vol.integrateCloud(dep, None, tr, count=True)
        n_cpu = ov.integrate(dep, None, synth.cam_from_vol_f32(tr), 224, 288)
        assert n_gpu == n_cpu and n_gpu > 0
    d, w, _ = vol.download()
    assert_same_f32(d, ov.d[224:288], "d")
    assert_same_f32(w, ov.w[224:288], "w")


def test_parity_odd_resolutions(gpu):
    # nx not a multiple of 4, non power-of-two axes: closed-form centres (tsdf_volume_octree.cpp:553-560)
    vol, sc = make_volume(64, res3=(50, 37, 41), size3=(0.25, 0.25, 0.25))
    ov = run_pair(vol, sc, 3, total=8)
    compare(vol, ov)


def test_parity_non_cubic_grid_size(gpu):
    """setGridSize with three different extents on a power-of-two grid: node centres descend from size_x on every
    axis, as in the reference's octree (tests/test_oracle_golden.py pins the oracle on this against oracle/_ref)."""
    vol, sc = make_volume(64, size3=(0.25, 0.2, 0.3), color=True)
    ov = run_pair(vol, sc, 3, total=8)
    compare(vol, ov)
    for a in range(3):
        assert np.array_equal(vol.centers(a), ov.centers(a)) and np.array_equal(vol.centers(a), vol.centers(0))


def test_depth_edge_cases_nan_inf_zero(gpu):
    # only NaN is "no return" (hpp:152); +Inf clamps to max_dist_pos, 0 is far behind every voxel
    vol, sc = make_volume(64)
    vol.reset()
    ov = OracleVolume(vol._p)
    tr = synth.turntable_pose(0, 8, sc.size)
    dep = sc.depth(tr)
    dep[40:60, 60:100] = np.inf
    dep[60:80, 60:100] = 0.0
    dep[80:90, :] = np.nan
    dep[30, 30] = -1.0
    assert vol.integrateCloud(dep, None, tr, count=True) == ov.integrate(dep, None, synth.cam_from_vol_f32(tr))
    compare(vol, ov)


def test_empty_frame_and_reset(gpu):
    vol, sc = make_volume(64)
    vol.reset()
    dep = np.full((120, 160), np.nan, np.float32)
    assert vol.integrateCloud(dep, None, np.eye(4), count=True) == 0
    d, w, _ = vol.download()
    assert (d == -1).all() and (w == 0).all()  # tsdf_volume_octree.cpp:217
    tr = synth.turntable_pose(0, 8, sc.size)
    assert vol.integrateCloud(sc.depth(tr), None, tr, count=True) > 0
    vol.reset()
    d, w, _ = vol.download()
    assert (d == -1).all() and (w == 0).all()


def test_z_slabs_equal_whole_grid(gpu):
    """K virtual Z-slabs on one device == the single-volume result, bit for bit (multi-GPU partition)."""
    vol, sc = make_volume(64, color=True)
    ov = run_pair(vol, sc, 3, total=8)
    d_all, w_all, rgb_all = vol.download()
    total = 0
    for zb, ze in [(0, 16), (16, 21), (21, 64)]:
        part, _ = make_volume(64, color=True)
        part.setZSlab(zb, ze)
        part.reset()
        for i, tr, dep, col in frames(sc, 3, 8):
            total += part.integrateCloud(dep, col, tr, count=True)
        d, w, rgb = part.download()
        assert_same_f32(d, d_all[zb:ze], "slab d")
        assert_same_f32(w, w_all[zb:ze], "slab w")
        assert np.array_equal(rgb, rgb_all[zb:ze])
    assert total == sum(ov.integrate(dep, col, synth.cam_from_vol_f32(tr)) for _, tr, dep, col in frames(sc, 3, 8)) \
        or total > 0


def test_upload_download_roundtrip(gpu):
    vol, sc = make_volume(64, color=True, res3=(30, 20, 10), size3=(0.25, 0.25, 0.25))
    vol.setLayout(capi.LAYOUT_F32W)  # arbitrary float weights need the float weight plane
    vol.reset()
    rng = np.random.RandomState(0)
    d = rng.randn(10, 20, 30).astype(np.float32)
    w = rng.rand(10, 20, 30).astype(np.float32)
    rgb = rng.randint(0, 256, (10, 20, 30, 3)).astype(np.uint8)
    vol.upload(d, w, rgb)
    d2, w2, rgb2 = vol.download()
    assert np.array_equal(d, d2) and np.array_equal(w, w2) and np.array_equal(rgb, rgb2)
    d3, w3, _ = vol.download(x0=3, y0=2, z0=1, nx=7, ny=5, nz=4)
    assert np.array_equal(d3, d[1:5, 2:7, 3:10]) and np.array_equal(w3, w[1:5, 2:7, 3:10])


def test_block_transfers_across_bounce_chunks(gpu):
    """Host transfers move through a pinned two-slot bounce buffer in 2 MiB chunks (tsdf_to_host / tsdf_to_device):
    blocks of 7.9 chunks, exactly 4 chunks, a few bytes and 14+ chunks, back to back and in both directions, must
    arrive intact."""
    vol, sc = make_volume(64, color=True, res3=(176, 168, 160), size3=(0.25, 0.25, 0.25))
    vol.setLayout(capi.LAYOUT_F32W)
    vol.reset()
    rng = np.random.RandomState(3)
    for (x0, y0, z0, nx, ny, nz) in [(5, 3, 1, 163, 161, 157), (0, 0, 0, 128, 128, 128), (7, 9, 11, 3, 1, 2), (0, 0, 0, 176, 168, 160)]:
        d = rng.randn(nz, ny, nx).astype(np.float32)
        w = rng.rand(nz, ny, nx).astype(np.float32)
        rgb = rng.randint(0, 256, (nz, ny, nx, 3)).astype(np.uint8)
        vol.upload(d, w, rgb, x0, y0, z0)
        d2, w2, rgb2 = vol.download(x0, y0, z0, nx, ny, nz)
        assert np.array_equal(d.view(np.uint32), d2.view(np.uint32)), (nx, ny, nz)
        assert np.array_equal(w, w2) and np.array_equal(rgb, rgb2), (nx, ny, nz)
    vol.close()


@pytest.mark.parametrize("color", [False, True])
@pytest.mark.parametrize("wmax", [100.0, 2.5, 0.0, 255.0])
def test_packed_layout_roundtrip_and_refusal(gpu, color, wmax):
    """PACKED stores the observation count k (w = min(k, max_weight)): every weight of that form survives
    upload/download bit for bit next to arbitrary d and rgb; anything else is refused, not rounded."""
    vol, sc = make_volume(64, color=color, res3=(30, 20, 10), size3=(0.25, 0.25, 0.25), max_weight=wmax)
    vol.reset()
    assert vol.getLayout() == capi.LAYOUT_PACKED
    rng = np.random.RandomState(1)
    d = rng.randn(10, 20, 30).astype(np.float32)
    k = rng.randint(0, int(np.ceil(wmax)) + 1, (10, 20, 30))
    w = np.minimum(k.astype(np.float32), np.float32(wmax))
    rgb = rng.randint(0, 256, (10, 20, 30, 3)).astype(np.uint8) if color else None
    vol.upload(d, w, rgb)
    d2, w2, rgb2 = vol.download()
    assert np.array_equal(d, d2) and np.array_equal(w, w2)
    if color:
        assert np.array_equal(rgb, rgb2)
        vol.upload(None, w[::-1].copy(), None)  # the count shares a word with the colour: neither disturbs the other
        assert np.array_equal(vol.download()[2], rgb) and np.array_equal(vol.download()[1], w[::-1])
    bad = w.copy()
    bad[3, 4, 5] = 0.5 if wmax != 0.5 else 0.25
    with pytest.raises(capi.TsdfHipError):
        vol.upload(None, bad, None)
    big, _ = make_volume(64, max_weight=300.0)
    big.reset()
    assert big.getLayout() == capi.LAYOUT_F32W  # AUTO falls back when one byte cannot hold the count
    big.setLayout(capi.LAYOUT_PACKED)
    with pytest.raises(capi.TsdfHipError):
        big.reset()


@pytest.mark.parametrize("layout", [capi.LAYOUT_F32W, capi.LAYOUT_PACKED])
@pytest.mark.parametrize("color", [False, True])
@pytest.mark.parametrize("wmax", [100.0, 3.0, 2.5])
def test_parity_both_layouts_through_weight_saturation(gpu, layout, color, wmax):
    """Same frames through both HBM layouts vs the oracle, long enough for the weight to saturate (also at a
    non-integer max_weight, where the saturated weight is not an integer)."""
    vol, sc = make_volume(64, color=color, max_weight=wmax)
    vol.setLayout(layout)
    ov = run_pair(vol, sc, 6, total=8)
    assert vol.getLayout() == layout
    compare(vol, ov)


@pytest.mark.parametrize("wmax", [255.0, 254.5])
def test_packed_count_saturates_at_one_byte(gpu, wmax):
    """max_weight in (254, 255] makes kmax == 255, the largest count byte: the 256th and later observations must
    leave the count (and so the weight) saturated instead of wrapping it to zero.  270 frames on a
    small grid, the same 8 noisy turntable frames cycled; every plane vs the oracle at the end and at frame 257."""
    vol, sc = make_volume(32, color=True, max_weight=wmax)
    vol.setLayout(capi.LAYOUT_PACKED)
    vol.reset()
    ov = OracleVolume(vol._p)
    fr = [(tr, dep, col) for _, tr, dep, col in frames(sc, 8, 8, noise=True)]
    for i in range(270):
        tr, dep, col = fr[i % 8]
        vol.integrateCloud(dep, col, tr)
        ov.integrate(dep, col, synth.cam_from_vol_f32(tr))
        if i in (255, 256, 269):
            _, w = compare(vol, ov)
    assert w.max() == wmax


def test_brick_cull_is_conservative_camera_inside_volume(gpu):
    """The brick-level frustum cull (dense counterpart of getFrustumCulledVoxels) may only drop blocks no voxel
    of which can be observed.  Camera INSIDE the volume with a short sensor range, as in the reference's README
    use: vs the oracle, and cull forced on == cull off bit for bit, with the counter proving blocks were cut."""
    sc = synth.scene_b(160, 120)
    try:
        outs = []
        for cull in (2, 0):
            capi.set_tuning("cull", cull)
            vol, _ = make_volume(128, 160, 120, color=True, size=10.0, zmin=0.0, zmax=3.0)
            vol.reset()
            ov = OracleVolume(vol._p) if cull == 2 else None
            tot = 0
            for i in range(5):
                tr = synth.scene_b_pose(i, 5)
                dep, col = sc.depth(tr), sc.bgra(i)
                n = vol.integrateCloud(dep, col, tr, count=True)
                tot += n
                if ov is not None:
                    assert n == ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr))
            d, w, rgb = vol.download()
            if ov is not None:
                assert_same_f32(d, ov.d, "d")
                assert np.array_equal(w, ov.w) and np.array_equal(rgb, ov.rgb)
            outs.append((d, w, rgb, tot))
            vol.close()
        assert 0 < outs[0][3] == outs[1][3] < 0.1 * 128 ** 3 * 5  # a small part of the grid is ever observed
        for a, b in zip(outs[0][:3], outs[1][:3]):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    finally:
        capi.set_tuning("cull", 1)


def test_brick_cull_random_poses_equal_no_cull(gpu):
    """Random poses (inside, outside, grazing, behind) on a non-cubic odd grid: forced cull == no cull."""
    rng = np.random.RandomState(5)
    sc = synth.Scene(1.0, 160, 120)
    poses = []
    for _ in range(12):
        eye = rng.uniform(-1.2, 1.2, 3)
        tgt = rng.uniform(-0.4, 0.4, 3)
        poses.append(synth.look_at_pose(eye, target=tgt))
    try:
        res = []
        for cull in (2, 0):
            capi.set_tuning("cull", cull)
            vol, _ = make_volume(64, 160, 120, res3=(70, 45, 33), size3=(1.0, 0.8, 0.6), zmin=0.05, zmax=0.9)
            vol.reset()
            counts = [vol.integrateCloud(sc.depth(tr), None, tr, count=True) for tr in poses]
            res.append((vol.download()[:2], counts))
            vol.close()
        assert res[0][1] == res[1][1] and sum(res[0][1]) > 1000
        assert np.array_equal(res[0][0][0].view(np.uint32), res[1][0][0].view(np.uint32))
        assert np.array_equal(res[0][0][1], res[1][0][1])
    finally:
        capi.set_tuning("cull", 1)


@pytest.mark.parametrize("slab", [None, (7, 31)])
def test_cull_sub_grid_offsets_on_a_wide_grid(gpu, slab):
    """The launch shrinks to the index box of the observable pyramid (observable_index_box): pointers, centre tables
    and limits are offset on the host.  A grid wide enough for several blocks along x (1024 voxels each) and y,
    whole and as a Z-slab handle, with cameras inside / outside / grazing / looking away, a sheared (non-rigid)
    pose and a short sensor range: forced cull == no cull, bit for bit, voxel counts included."""
    rng = np.random.RandomState(17)
    res3, size3 = (2304, 72, 40), (9.0, 0.3, 0.16)
    sc = synth.Scene(1.0, 160, 120)
    poses = []
    for k in range(14):
        eye = np.array([rng.uniform(-5.0, 5.0), rng.uniform(-0.3, 0.3), rng.uniform(-0.4, 0.4)])
        tgt = np.array([rng.uniform(-4.5, 4.5), rng.uniform(-0.1, 0.1), rng.uniform(-0.05, 0.05)])
        tr = synth.look_at_pose(eye, target=tgt)
        if k == 5:   # a general affine pose: cam_from_vol is not a rotation
            tr = tr.copy()
            tr[:3, :3] = tr[:3, :3] @ np.array([[1.1, 0.05, 0.0], [0.0, 0.9, 0.02], [0.03, 0.0, 1.0]])
        poses.append(tr)
    poses.append(synth.look_at_pose((6.0, 0.0, 0.0), target=(12.0, 0.0, 0.0)))    # outside, looking away: nothing
    poses.append(synth.look_at_pose((6.5, 0.01, 0.0), target=(0.0, 0.0, 0.0)))    # outside, grid beyond the sensor range
    frames = [np.full((120, 160), d, np.float32) for d in rng.uniform(0.2, 1.4, len(poses))]
    try:
        res = []
        for cull in (2, 0):
            capi.set_tuning("cull", cull)
            vol, _ = make_volume(64, 160, 120, color=True, res3=res3, size3=size3, zmin=0.05, zmax=1.5)
            if slab:
                vol.setZSlab(*slab)
            vol.reset()
            counts = [vol.integrateCloud(dep, sc.bgra(i), tr, count=True) for i, (tr, dep) in enumerate(zip(poses, frames))]
            res.append((vol.download(), counts))
            vol.close()
        assert res[0][1] == res[1][1] and sum(res[0][1]) > 10000 and min(res[0][1]) == 0 < max(res[0][1])
        for a, b in zip(res[0][0], res[1][0]):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
        w = res[0][0][1]
        assert 0.0 < (w > 0).mean() < 0.95   # the frames covered only part of the grid
    finally:
        capi.set_tuning("cull", 1)


@pytest.mark.parametrize("color", [False, True])
def test_pipelined_host_frames_equal_synchronous(gpu, color):
    """tsdf_hip_integrate_async: frames handed over back to back from ONE reused host buffer (the call must have
    copied it before returning), more frames than ring slots, a synchronous call and a download in between --
    same volume as the oracle, bit for bit."""
    vol, sc = make_volume(64, color=color)
    vol.reset()
    ov = OracleVolume(vol._p)
    dep_buf = np.empty((sc.height, sc.width), np.float32)
    col_buf = np.empty((sc.height, sc.width, 4), np.uint8)
    for i in range(9):
        tr = synth.turntable_pose(i, 12, sc.size)
        dep_buf[...] = sc.depth(tr, noise_seed=7 + i)
        col_buf[...] = sc.bgra(i)
        ov.integrate(dep_buf, col_buf if color else None, synth.cam_from_vol_f32(tr))
        if i == 4:
            vol.integrateCloud(dep_buf, col_buf if color else None, tr)  # synchronous call in the middle of the stream
        else:
            vol.integrateCloud(dep_buf, col_buf if color else None, tr, pipelined=True)
        dep_buf.fill(np.nan)  # the caller may scribble over its buffers at once
        col_buf.fill(0)
        if i == 6:
            assert_same_f32(vol.download()[0], ov.d, "d after 7 frames")
    compare(vol, ov)
    vol.close()


def test_center_tables_match_oracle(gpu):
    vol, sc = make_volume(64, size=3.3)  # non-dyadic size: octree-descent sums, not the closed form
    vol.reset()
    ov = OracleVolume(vol._p)
    for a in range(3):
        assert_same_f32(vol.centers(a), ov.centers(a), f"centres axis {a}")


def test_linearity_property_512(gpu):
    """Size-independent property at a roofline-sized grid: integrating the same frame k times gives
    d == d_1 wherever observed, and w == min(k, max_weight); the counter is frame-invariant."""
    vol, sc = make_volume(512, width=640, height=480)
    vol.setZSlab(192, 320)  # 128 planes = 268 MB of d+w, above the caches' comfort zone
    vol.reset()
    tr = synth.turntable_pose(1, 8, sc.size)
    dep = sc.depth(tr)
    n1 = vol.integrateCloud(dep, None, tr, count=True)
    d1, w1, _ = vol.download()
    for _ in range(3):
        assert vol.integrateCloud(dep, None, tr, count=True) == n1
    d4, w4, _ = vol.download()
    obs = w1 > 0
    assert int(obs.sum()) == n1
    assert np.array_equal(w4[obs], np.full(n1, 4, np.float32)) and (w4[~obs] == 0).all()
    assert np.max(np.abs(d4[obs] - d1[obs])) <= 2e-7  # (d*w + d)/(w+1) re-rounds in fp32
    assert (d4[~obs] == -1).all()


def test_integrate_fuzz_equals_the_oracle(gpu):
    """The HIP path on the adversarial inputs of tests/test_oracle_golden.py::test_integrate_fuzz_equals_compiled_reference
    (where the oracle is pinned to the reference): random non-dyadic grid sizes, intrinsics, sensor bounds, asymmetric
    truncation, small weight limits, cameras anywhere incl. sheared poses, depth images with NaN / inf / 0 / negative /
    huge values -- d, w, rgb and the observed-voxel counts equal the oracle's, in both layouts."""
    rng = np.random.RandomState(2024)
    for case in range(24):
        res = int(rng.choice([16, 32]))
        size = float(rng.choice([0.125, 0.3, 1.0, 3.0]))
        W, H = (48, 36) if case % 2 else (64, 48)
        f = float(rng.uniform(20.0, 80.0))
        fx, fy, cx, cy = f, f * float(rng.uniform(0.9, 1.1)), W / 2 - 0.5 + float(rng.uniform(-3, 3)), H / 2 - 0.5
        zmin, zmax = float(rng.choice([0.0, 0.05 * size])), float(rng.uniform(0.8, 3.5)) * size
        pos, neg = float(rng.uniform(0.02, 0.3)) * size, float(rng.uniform(0.02, 0.3)) * size
        wmax = float(rng.choice([100.0, 2.0, 3.5]))
        color = bool(case % 3)
        vol, _ = make_volume(res, W, H, color=color, size=size, zmin=zmin, zmax=zmax, trunc=(pos, neg), max_weight=wmax)
        vol.setCameraIntrinsics(fx, fy, cx, cy)
        if case % 4 == 3:
            vol.setLayout(capi.LAYOUT_F32W)
        vol.reset()
        ov = OracleVolume(vol._p)
        for i in range(5):
            eye = rng.uniform(-1.6, 1.6, 3) * size * (1.0 if rng.rand() < 0.7 else 0.2)
            tr = synth.look_at_pose(eye, target=rng.uniform(-0.3, 0.3, 3) * size)
            if rng.rand() < 0.25:
                tr = tr.copy()
                tr[:3, :3] = tr[:3, :3] @ (np.eye(3) + rng.uniform(-0.2, 0.2, (3, 3)))
            dep = (rng.uniform(0.2, 2.5, (H, W)) * size).astype(np.float32)
            junk = rng.rand(H, W)
            dep[junk < 0.05] = np.nan
            dep[(junk >= 0.05) & (junk < 0.07)] = np.inf
            dep[(junk >= 0.07) & (junk < 0.09)] = 0.0
            dep[(junk >= 0.09) & (junk < 0.10)] = -1.0
            dep[(junk >= 0.10) & (junk < 0.11)] = 3.0e38
            col = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
            n_gpu = vol.integrateCloud(dep, col if color else None, tr, count=True)
            n_cpu = ov.integrate_culled(dep, col if color else None, tr, synth.cam_from_vol_f32(tr))
            assert n_gpu == n_cpu, (case, i, n_gpu, n_cpu)
        d, w, rgb = vol.download()
        assert_same_f32(d, ov.d, f"case {case}: d")
        assert_same_f32(w, ov.w, f"case {case}: w")
        if color:
            assert np.array_equal(rgb, ov.rgb), f"case {case}: rgb"
        vol.close()


def launch_info(vol):
    import ctypes as C
    out = (C.c_int32 * 4)()
    capi.check(capi.load().tsdf_hip_last_launch_info(vol._need(), out), "last_launch_info")
    return [int(out[0]) & 0xff, int(out[1]), int(out[2]), int(out[3]), bool(int(out[0]) & 0x100)]  # [4]: the pipelined row loop (k_integrate_p)


@pytest.mark.parametrize("color,layout,wmax", [(True, capi.LAYOUT_AUTO, 100.0), (False, capi.LAYOUT_AUTO, 4.0),
                                               (True, capi.LAYOUT_F32W, 100.0), (True, capi.LAYOUT_AUTO, 2.5)])
def test_all_inside_instance_equals_general_instance_and_oracle(gpu, color, layout, wmax):
    """k_integrate's ALLIN instance (every voxel provably in range and in the image: no per-voxel range / bounds tests,
    one certificate compare per quad, indexed frame gather) against the general instance and the oracle: same frames,
    knob "allin" on and off, noisy depth with NaN holes so that unobserved / in-band / free-space quads all occur, enough
    frames to pass the weight limit.  A non-integer max_weight in the PACKED layout must fall back to the general one."""
    vols = []
    try:
        for allin in (1, 0):
            capi.set_tuning("allin", allin)
            vol, sc = make_volume(96, color=color, max_weight=wmax)
            vol.setLayout(layout)
            vol.reset()
            ov = OracleVolume(vol._p)
            used = []
            for i, tr, dep, col in frames(sc, 7, 9, noise=True):
                dep = dep.copy()
                dep[(i * 7) % 50::53, ::3] = np.nan
                n_gpu = vol.integrateCloud(dep, col if color else None, tr, count=(i % 2 == 0))
                used.append(launch_info(vol)[0])
                n_cpu = ov.integrate(dep, col if color else None, synth.cam_from_vol_f32(tr))
                assert n_gpu is True or n_gpu == n_cpu
            packed = vol.getLayout() == capi.LAYOUT_PACKED
            expect = int(allin == 1 and not (packed and wmax != int(wmax)))
            assert used == [expect] * 7, (used, expect)
            compare(vol, ov)
            vols.append(vol.download())
            vol.close()
    finally:
        capi.set_tuning("allin", 1)
    assert_same_f32(vols[0][0], vols[1][0], "d: ALLIN vs general")
    assert np.array_equal(vols[0][1], vols[1][1])


def test_all_inside_instance_is_not_chosen_when_a_voxel_may_leave_the_image_or_the_range(gpu):
    """The host's proof obligation: eight corner voxels inside {sensor range, image minus a one-pixel border}.  A camera
    close enough that a corner leaves the image, a sensor range that cuts the volume, a camera inside the volume and a
    grid whose x resolution is not a multiple of 4 must all take the general instance -- and still equal the oracle."""
    cases = []
    vol, sc = make_volume(64)
    cases.append(("turntable", vol, sc, synth.turntable_pose(1, 8, sc.size), 1))
    vol, sc = make_volume(64)
    cases.append(("close: corners outside the image", vol, sc, synth.turntable_pose(1, 8, sc.size, radius_factor=0.9), 0))
    vol, sc = make_volume(64, zmax=0.5)
    cases.append(("far plane cuts the volume", vol, sc, synth.turntable_pose(1, 8, sc.size), 0))
    vol, sc = make_volume(64)
    cases.append(("camera inside", vol, sc, synth.look_at_pose((0.01, 0.0, -0.02), target=(0.0, 0.0, 1.0)), 0))
    vol, sc = make_volume(64, res3=(66, 64, 64))
    cases.append(("nx not a multiple of 4", vol, sc, synth.turntable_pose(1, 8, sc.size), 0))
    for name, vol, sc, tr, want in cases:
        vol.reset()
        ov = OracleVolume(vol._p)
        dep = sc.depth(tr)
        assert vol.integrateCloud(dep, None, tr, count=True) == ov.integrate_culled(dep, None, tr, synth.cam_from_vol_f32(tr)), name
        assert launch_info(vol)[0] == want, (name, launch_info(vol))
        compare(vol, ov)
        vol.close()


@pytest.mark.parametrize("color,layout", [(True, capi.LAYOUT_AUTO), (False, capi.LAYOUT_AUTO), (True, capi.LAYOUT_F32W)])
def test_reference_cull_replication_mode(gpu, color, layout):
    """setReferenceCull(True) (tsdf_hip_set_reference_cull): where the reference's frustum cull is NOT a no-op -- a
    principal point far off centre, so that the 1.1 x FOV pyramid around the optical axis cuts the image -- the product
    drops exactly the voxels the reference drops: equal to the oracle's restatement of the cull (which
    tests/test_oracle_golden.py pins to the compiled reference) and, where oracle/_ref is present, to the compiled
    reference itself.  Without the mode the product integrates the superset (the default, unchanged).  Both layouts,
    with and without colour, counts included; a centred camera in the same mode keeps the fast kernel."""
    res, W, H, size = 64, 64, 48, 1.0  # (a power of two: the compiled reference is an octree)
    fx = fy = 110.0
    cy = H / 2 - 0.5
    rng = np.random.RandomState(4)
    for cx, off_centre in ((W / 2 - 0.5 + 5.5, True), (W / 2 - 0.5, False)):
        vols = {}
        for mode in (True, False):
            v, _ = make_volume(res, W, H, color=color, size=size)
            v.setCameraIntrinsics(fx, fy, cx, cy)
            v.setLayout(layout)
            v.setReferenceCull(mode)
            v.reset()
            vols[mode] = v
        assert vols[True].referenceCullIsNoop() == (not off_centre)
        oc, ov = OracleVolume(vols[True]._p), OracleVolume(vols[True]._p)
        ref = None
        try:
            from oracle import refbind
            if refbind.available():
                ref = refbind.RefVolume(res, size, W, H, fx, fy, cx, cy, 0.0, 3 * size, color=color)
        except ImportError:
            pass
        for i in range(4):
            tr = synth.look_at_pose(np.array([1.9, 0.25 * i - 0.3, 0.3]) * size, target=np.zeros(3))
            dep = (rng.uniform(1.2, 2.6, (H, W)) * size).astype(np.float32)
            col = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
            c = col if color else None
            n_cull = oc.integrate_culled(dep, c, tr, synth.cam_from_vol_f32(tr))
            n_all = ov.integrate(dep, c, synth.cam_from_vol_f32(tr))
            assert vols[True].integrateCloud(dep, c, tr, count=True) == n_cull
            assert vols[False].integrateCloud(dep, c, tr, count=True) == n_all
            assert (n_cull < n_all) == off_centre
            if ref is not None:
                ref.integrate(dep, c, tr)
        compare(vols[True], oc)
        compare(vols[False], ov)
        if ref is not None:
            d, w, rgb, _, _ = ref.dump_dense()
            gd, gw, grgb = vols[True].download()
            assert_same_f32(gd, d, "d vs the compiled reference")
            assert np.array_equal(gw, w) and (not color or np.array_equal(grgb, rgb))
            ref.close()
        for v in vols.values():
            v.close()


@pytest.mark.parametrize("color,layout,res3", [(True, capi.LAYOUT_AUTO, (64, 64, 64)), (False, capi.LAYOUT_AUTO, (70, 45, 33)),
                                               (True, capi.LAYOUT_F32W, (32, 32, 32)), (False, capi.LAYOUT_F32W, (384, 40, 24))])
def test_default_path_is_the_reference_cull_through_the_row_intervals(gpu, color, layout, res3):
    """a caller who never heard of setReferenceCull gets the reference's voxels where its frustum cull
    decides them (tsdf_volume_octree.cpp:619-652, hpp:93-94) -- principal point up to 40 % off centre, a far plane through
    the volume, cameras inside, sheared poses -- AND through the fast kernel: the LIVE instance of k_integrate masked by
    k_rows' row intervals (last_launch_info[2] == 2), bit-identical to the plain per-voxel kernel applying the six planes
    itself (knob refcull_plain) and to the culled oracle; counts included; a centred camera that sees the whole volume
    keeps the ALLIN instance although the planes are set."""
    rng = np.random.RandomState(hash((color, layout, res3)) % 1000)
    W, H = 64, 48
    size = 1.0
    outs = {}
    try:
        for plain in (0, 1):
            capi.set_tuning("refcull_plain", plain)
            rng = np.random.RandomState(77)
            seen_modes = set()
            for case in range(6):
                f = float(rng.uniform(40.0, 110.0))
                cx = W / 2 - 0.5 + float(rng.uniform(-0.4, 0.4)) * W / 2 * (case % 3 != 2)
                cy = H / 2 - 0.5 + float(rng.uniform(-0.4, 0.4)) * H / 2 * (case % 3 == 1)
                zmax = float(rng.choice([0.9, 3.0]))
                cubic = res3[0] == res3[1] == res3[2]
                vol, _ = make_volume(res3[0], W, H, color=color, size=size, zmin=float(rng.choice([0.0, 0.1])), zmax=zmax,
                                     max_weight=float(rng.choice([100.0, 2.0])), res3=res3,
                                     size3=None if cubic else (size * res3[0] / 64, size * res3[1] / 64, size * res3[2] / 64))
                vol.setCameraIntrinsics(f, f * float(rng.uniform(0.9, 1.1)), cx, cy)
                vol.setLayout(layout)
                vol.reset()
                oc = OracleVolume(vol._p)
                ext = max(vol.getGridSize())
                for i in range(5):
                    eye = rng.uniform(-1.5, 1.5, 3) * ext * (1.0 if i % 2 else 0.25)
                    tr = synth.look_at_pose(eye, target=rng.uniform(-0.3, 0.3, 3) * ext)
                    if i == 3:
                        tr = tr.copy()
                        tr[:3, :3] = tr[:3, :3] @ (np.eye(3) + rng.uniform(-0.15, 0.15, (3, 3)))
                    dep = (rng.uniform(0.2, 2.5, (H, W)) * ext).astype(np.float32)
                    dep[rng.rand(H, W) < 0.05] = np.nan
                    col = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
                    c = col if color else None
                    want = oc.integrate_culled(dep, c, tr, synth.cam_from_vol_f32(tr))
                    got = vol.integrateCloud(dep, c, tr, count=(i % 2 == 0))
                    assert got is True or got == want, (case, i, got, want)
                    seen_modes.add(launch_info(vol)[2] if not plain else -1)
                compare(vol, oc)
                outs.setdefault(case, []).append(vol.download())
                vol.close()
            if not plain:
                assert 2 in seen_modes, seen_modes  # the row intervals carried the cull at least once
    finally:
        capi.set_tuning("refcull_plain", 0)
    for case, (a, b) in outs.items():
        for x, y in zip(a, b):
            assert (x is None and y is None) or np.array_equal(x.view(np.uint8), y.view(np.uint8)), case
    # the headline shape: planes set (default), camera outside, whole volume in view and in range -> ALLIN, no flags
    vol, sc = make_volume(64, color=color)
    vol.setLayout(layout)
    vol.reset()
    ov = OracleVolume(vol._p)
    for i, tr, dep, col in frames(sc, 3, 8):
        c = col if color else None
        assert vol.integrateCloud(dep, c, tr, count=True) == ov.integrate_culled(dep, c, tr, synth.cam_from_vol_f32(tr))
        assert launch_info(vol)[0] == 1 and launch_info(vol)[2] == 0, launch_info(vol)
    compare(vol, ov)
    vol.close()


@pytest.mark.parametrize("color,layout", [(True, capi.LAYOUT_PACKED), (False, capi.LAYOUT_PACKED), (True, capi.LAYOUT_F32W)])
def test_calibration_sweeps_leave_every_bit_alone(gpu, color, layout):
    """The PMC calibration sweeps (tsdf_hip_selftest_sweep: read-modify-write of every plane; tsdf_hip_selftest_read_sweep:
    one word per 4 / 64 / 128 bytes of the distance plane) run inside bench.py's profiled process: they must report the
    bytes they move and change nothing -- also not a -0.0 or a NaN payload."""
    import ctypes as C
    vol, sc = make_volume(64, color=color)
    vol.setLayout(layout)
    vol.reset()
    for i, tr, dep, col in frames(sc, 3, 8):
        vol.integrateCloud(dep, col if color else None, tr)
    d, w, rgb = vol.download()
    d[3, 4, 5:9] = np.array([-0.0, np.nan, np.inf, 1e-42], dtype=np.float32)  # a denormal too
    w[3, 4, 5:9] = 1.0
    vol.upload(d, w, rgb)
    lib, h = capi.load(), vol._need()
    br, bw = C.c_uint64(), C.c_uint64()
    capi.check(lib.tsdf_hip_selftest_sweep(h, C.byref(br), C.byref(bw)), "sweep")
    n = 64 ** 3
    per_voxel = 4 + (0 if layout == capi.LAYOUT_PACKED else 4) + (4 if color else (1 if layout == capi.LAYOUT_PACKED else 0))
    assert br.value == bw.value == n * per_voxel
    for stride in (4, 64, 128):
        span, words = C.c_uint64(), C.c_uint64()
        capi.check(lib.tsdf_hip_selftest_read_sweep(h, stride, C.byref(span), C.byref(words)), "read_sweep")
        assert span.value == n * 4 and words.value == n * 4 // stride
    assert lib.tsdf_hip_selftest_read_sweep(h, 32, None, None) == capi.E_INVALID
    d2, w2, rgb2 = vol.download()
    assert np.array_equal(d2.view(np.uint32), d.view(np.uint32))
    assert np.array_equal(w2.view(np.uint32), w.view(np.uint32))
    if color:
        assert np.array_equal(rgb2, rgb)
    vol.close()


def test_every_reachable_k_integrate_instance_equals_the_oracle(gpu):
    """k_integrate is compiled in 80 instances (transform order x colour x certified
    projection x counting x layout x {general, ALLIN (certified projection only), row intervals}) and k_integrate2 in 8; this
    test drives the public entry points into EVERY one of them -- knobs and poses choose, tsdf_hip_last_launch_info
    confirms which one ran -- on noisy frames with NaN holes, twice per volume so that the second update works on real
    state, and compares every voxel with the culled oracle (the reference's integrateCloud incl. its frustum cull)."""
    import torch
    W, H, res = 160, 120, 32
    hit = set()
    try:
        for order in (0, 1):
            for color in (False, True):
                for layout in (capi.LAYOUT_PACKED, capi.LAYOUT_F32W):
                    for fp in (1, 0):
                        for kind in ("general", "allin", "rows", "cull"):
                            if kind == "allin" and not fp:
                                continue  # the ALLIN instances exist only with the certified projection
                            capi.set_tuning("fast_projection", fp)
                            capi.set_tuning("allin", 0 if kind == "general" else 1)
                            capi.set_tuning("pipe", 0)  # k_integrate's own instances here; k_integrate_p has its block below
                            vol, sc = make_volume(res, W, H, color=color, order=order, max_weight=100.0)
                            if kind == "cull":  # whole grid in view, principal point 60 % off centre: the reference's cull cuts the grid
                                sc.cx += 0.6 * W / 2
                                vol.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
                            vol.setLayout(layout)
                            vol.reset()
                            ov = OracleVolume(vol._p)
                            for i in range(4):
                                if kind == "rows":  # camera inside the grid
                                    tr = synth.look_at_pose((0.02 * i, 0.01, -0.03), target=(0.05, 0.0, 1.0))
                                elif kind == "cull":
                                    psi = float(np.arctan(0.6 * (W / 2) / sc.fx))
                                    yaw = np.eye(4)
                                    yaw[0, 0], yaw[0, 2], yaw[2, 0], yaw[2, 2] = np.cos(psi), np.sin(psi), -np.sin(psi), np.cos(psi)
                                    tr = synth.turntable_pose(2 * i + 1, 8, sc.size) @ yaw  # the grid seen edge-on: its corners leave the pyramid
                                else:
                                    tr = synth.turntable_pose(i, 9, sc.size)
                                dep = sc.depth(tr, noise_seed=40 + i)
                                dep[(i * 5) % 30::31, ::3] = np.nan
                                col = sc.bgra(i) if color else None
                                count = i % 2 == 0
                                want = ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr))
                                got = vol.integrateCloud(dep, col, tr, count=count)
                                assert got is True or got == want, (order, color, layout, fp, kind, i, got, want)
                                info = launch_info(vol)
                                expect = {"general": (0, 0), "allin": (1, 0), "rows": (0, None), "cull": (0, 2)}[kind]
                                ok = info[0] == expect[0] and (expect[1] is None or info[2] == expect[1]) and info[1] == fp and not info[4]
                                assert ok and (kind != "rows" or info[2] in (1, 2)), (order, color, layout, fp, kind, info)
                                hit.add((order, color, layout, fp, count, kind))
                            compare(vol, ov)
                            vol.close()
        # k_integrate_p (the software-pipelined row loop: ALLIN, PACKED, no colour): transform order x counting
        capi.set_tuning("fast_projection", 1)
        capi.set_tuning("allin", 1)
        capi.set_tuning("pipe", 3)
        for order in (0, 1):
          for color in (False, True):  # k_integrate_p / k_integrate_pc
            vol, sc = make_volume(res, W, H, color=color, order=order, max_weight=100.0)
            vol.reset()
            ov = OracleVolume(vol._p)
            for i in range(4):
                tr = synth.turntable_pose(i, 9, sc.size)
                dep = sc.depth(tr, noise_seed=40 + i)
                dep[(i * 5) % 30::31, ::3] = np.nan
                col = sc.bgra(i) if color else None
                count = i % 2 == 0
                want = ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr))
                got = vol.integrateCloud(dep, col, tr, count=count)
                assert got is True or got == want, (order, color, i, got, want)
                info = launch_info(vol)
                assert info[0] == 1 and info[1] == 1 and info[2] == 0 and info[4], info
                hit.add((order, color, "kp", count))
            compare(vol, ov)
            vol.close()
        # k_integrate2: transform order x colour x counting (PACKED, certified projection, both poses ALLIN; knob fuse2 = 2: without
        # colour the default leaves pairs to the pipelined single-frame kernel)
        capi.set_tuning("fuse2", 2)
        for order in (0, 1):
            for color in (False, True):
                vol, sc = make_volume(res, W, H, color=color, order=order)
                vol.reset()
                ov = OracleVolume(vol._p)
                keep = []
                for k in range(2):
                    pair, want = [], []
                    for i in (2 * k, 2 * k + 1):
                        tr = synth.turntable_pose(i, 9, sc.size)
                        dep = sc.depth(tr, noise_seed=70 + i)
                        dep[(i * 5) % 30::31, ::3] = np.nan
                        col = sc.bgra(i) if color else None
                        t = torch.empty((2, H, W), dtype=torch.float32, device="cuda")
                        t[0].copy_(torch.from_numpy(dep))
                        if color:
                            t[1].view(torch.uint8).view(H, W, 4).copy_(torch.from_numpy(col))
                        keep.append(t)
                        pair.append((t[0].data_ptr(), t[1].data_ptr() if color else 0, tr))
                        want.append(ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr)))
                    fused, counts = vol.integrateCloudDevice2(pair[0], pair[1], count=(k == 0))
                    assert fused and launch_info(vol)[0] == 2 and (counts is None or counts == want)
                    hit.add((order, color, "k2", k == 0))
                compare(vol, ov)
                vol.close()
    finally:
        capi.set_tuning("fast_projection", -1)
        capi.set_tuning("allin", 1)
        capi.set_tuning("pipe", 1)
        capi.set_tuning("fuse2", 1)
    # 8 x {certified: general, ALLIN, row intervals (camera inside), row intervals (cull); exact projection: the same without
    # ALLIN} x counting or not: all 80 k_integrate instances (the row-interval one twice) + the 4 + 4 of k_integrate_p /
    # k_integrate_pc + the 8 of k_integrate2
    assert len(hit) == 2 * 2 * 2 * 2 * (4 + 3) + 8 + 8, len(hit)


@pytest.mark.parametrize("color", [False, True], ids=["k_integrate_p", "k_integrate_pc"])
@pytest.mark.parametrize("rows_per_block", [8, 16, 24, 32, 40, 48, 64, 96])
def test_pipelined_row_loop_equals_the_oracle_and_the_plain_row_loop(gpu, rows_per_block, color):
    """k_integrate_p / k_integrate_pc (round 6: two rows in flight per wave, stage A = projection + every load of a row, stage
    B = update + stores; with colour the voxel words are asked by a predictor two rows back, an observed quad it missed asks
    late) on a 96^3 grid, whose blocks walk 1, 2, 3, 4, 5, 6, 8 or 12 row steps (TY = 8 rows per step; with 5 and 8 the grid's last
    block is SHORTER than the others: 16 of 40 rows, 32 of 64): the tail of
    one row, of a pair, the odd tail behind the steady-state loop and the loop itself -- noisy depth with NaN holes, frames
    past the weight limit (max_weight 4), counting on alternate frames -- against the oracle voxel for voxel, against
    k_integrate's own instance (knob pipe = 0) plane for plane, observed-voxel counts included; and the row count the
    pipelined kernel needs (ny a multiple of TY, the rows of one step) falls back by itself when it does not hold."""
    outs = []
    try:
        capi.set_tuning("rows_per_block", rows_per_block)
        for pipe in (3, 0):
            capi.set_tuning("pipe", pipe)
            vol, sc = make_volume(96, color=color, max_weight=4.0)
            vol.reset()
            ov = OracleVolume(vol._p)
            counts = []
            for i, tr, dep, col in frames(sc, 7, 9, noise=True):
                dep = dep.copy()
                dep[(i * 7) % 50::53, ::3] = np.nan
                c = col if color else None
                n_gpu = vol.integrateCloud(dep, c, tr, count=(i % 2 == 0))
                info = launch_info(vol)
                assert info[0] == 1 and info[4] == bool(pipe), (pipe, info)
                n_cpu = ov.integrate(dep, c, synth.cam_from_vol_f32(tr))
                assert n_gpu is True or n_gpu == n_cpu, (pipe, i, n_gpu, n_cpu)
                counts.append(n_gpu)
            compare(vol, ov)
            outs.append((vol.download(), counts))
            vol.close()
        assert_same_f32(outs[0][0][0], outs[1][0][0], "d: pipelined vs plain row loop")
        assert np.array_equal(outs[0][0][1], outs[1][0][1]) and outs[0][1] == outs[1][1]
        assert (outs[0][0][2] is None and outs[1][0][2] is None) or np.array_equal(outs[0][0][2], outs[1][0][2])
        # 100 rows are no multiple of the 8 rows of a step: the launch takes k_integrate's instance by itself
        capi.set_tuning("pipe", 3)
        vol, sc = make_volume(96, color=color, res3=(96, 100, 96))
        vol.reset()
        ov = OracleVolume(vol._p)
        tr = synth.turntable_pose(1, 8, sc.size)
        dep, c = sc.depth(tr), (sc.bgra(1) if color else None)
        assert vol.integrateCloud(dep, c, tr, count=True) == ov.integrate(dep, c, synth.cam_from_vol_f32(tr))
        assert launch_info(vol)[0] == 1 and not launch_info(vol)[4], launch_info(vol)
        compare(vol, ov)
        vol.close()
    finally:
        capi.set_tuning("rows_per_block", 64)
        capi.set_tuning("pipe", 1)


def test_planes_fastest_block_order_changes_nothing(gpu):
    """Knob zfast (tsdf_block_coords): the hardware grid handed out planes-first -- what a launch uses by itself when the
    frame outgrows an XCD's L2 (1280x960 + colour) -- against the default order and the oracle: the whole grid in view
    (ALLIN), a camera inside (row intervals, block flags indexed by logical coordinates), a Z-slab handle on a grid several
    x chunks wide, counting or not; and a frame big enough to switch it on by itself."""
    outs = {}
    try:
        for zfast in (0, 1):
            capi.set_tuning("zfast", zfast)
            res = []
            for name, kw, slab, poses in (
                    ("allin", dict(res=64, color=True), None, [synth.turntable_pose(i, 8, 0.25) for i in range(3)]),
                    ("inside", dict(res=64, color=False), None, [synth.look_at_pose((0.01 * i, 0.0, -0.02), target=(0.0, 0.01, 1.0)) for i in range(3)]),
                    ("wide slab", dict(res=64, color=True, res3=(2304, 40, 48), size3=(9.0, 0.15625, 0.1875), zmax=20.0), (7, 31),
                     [synth.look_at_pose((0.5 * i - 0.5, 0.02, -11.0), target=(0.0, 0.0, 0.0)) for i in range(3)])):
                vol, sc = make_volume(kw.pop("res"), 160, 120, **kw)
                if slab:
                    vol.setZSlab(*slab)
                vol.reset()
                ov = OracleVolume(vol._p)
                color = bool(vol._p.integrate_color)
                rng = np.random.RandomState(11)
                for i, tr in enumerate(poses):
                    dep = sc.depth(tr, noise_seed=9 + i) if name != "wide slab" else rng.uniform(10.8, 11.2, (120, 160)).astype(np.float32)
                    col = sc.bgra(i) if color else None
                    zb, ze = slab if slab else (0, 0)
                    want = ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr), zb, ze)
                    got = vol.integrateCloud(dep, col, tr, count=(i != 1))
                    assert got is True or got == want, (zfast, name, i, got, want)
                d, w, rgb = vol.download()
                zs = slice(*slab) if slab else slice(None)
                assert_same_f32(d, ov.d[zs], f"{name}: d")
                assert np.array_equal(w, ov.w[zs]) and (rgb is None or np.array_equal(rgb, ov.rgb[zs]))
                assert (w > 0).mean() > 0.05, name
                res.append((d, w))
                vol.close()
            outs[zfast] = res
    finally:
        capi.set_tuning("zfast", -1)
    for (d0, w0), (d1, w1) in zip(outs[0], outs[1]):
        assert np.array_equal(d0.view(np.uint32), d1.view(np.uint32)) and np.array_equal(w0, w1)
    # a 1024x768 colour frame (6.3 MB) switches the order on by itself
    vol, sc = make_volume(32, 1024, 768, color=True)
    vol.reset()
    ov = OracleVolume(vol._p)
    tr = synth.turntable_pose(1, 8, sc.size)
    dep, col = sc.depth(tr), sc.bgra(1)
    assert vol.integrateCloud(dep, col, tr, count=True) == ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr))
    compare(vol, ov)
    vol.close()
