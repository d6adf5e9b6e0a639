"""CPU tier: the launch-shrinking index box of integrateCloud (observable_index_box, cpu_tsdf_amd/csrc/
tsdf_integrate.hip) must contain every voxel the reference's updateVoxel can possibly observe -- otherwise results
would silently change.  Property test on the host (tsdf_hip_selftest_index_box needs no device): random grids
(non-cubic, dyadic and not), intrinsics, sensor ranges and poses (inside, outside, looking away, sheared); the
oracle integrates a frame whose every pixel returns a far depth, so that EVERY voxel passing the range and
image tests (hpp:146, .cpp:611-617) is observed, and the observed set is compared with the box."""
import ctypes as C

import numpy as np

from cpu_tsdf_amd import capi, synth
from oracle.oracle import OracleVolume


def index_box(p, T):
    box = (C.c_int32 * 6)()
    state = C.c_int32(-1)
    capi.check(capi.load().tsdf_hip_selftest_index_box(C.byref(p), capi.as_f32p(T), box, C.byref(state)), "index_box")
    return int(state.value), np.array(box[:3]), np.array(box[3:])


def test_index_box_contains_every_observable_voxel():
    rng = np.random.RandomState(23)
    tight = empty = whole = 0
    for case in range(400):
        p = capi.default_params()
        res = [int(r) for r in rng.choice([16, 24, 32, 40, 56], 3)]
        size = [float(s) for s in rng.uniform(0.5, 4.0, 3)]
        p.res[:], p.size[:] = res, size
        W, H = (64, 48) if case % 3 else (40, 56)
        p.image_width, p.image_height = W, H
        f = float(rng.uniform(25.0, 90.0))
        p.fx, p.fy, p.cx, p.cy = f, f * float(rng.uniform(0.8, 1.2)), W / 2 - 0.5 + float(rng.uniform(-5, 5)), H / 2 - 0.5
        p.min_sensor_dist = float(rng.choice([0.0, 0.2]))
        p.max_sensor_dist = float(rng.uniform(0.3, 6.0))
        p.max_dist_pos = p.max_dist_neg = 0.03
        ext = max(size)
        eye = rng.uniform(-1.2 * ext, 1.2 * ext, 3) * (1.0 if case % 4 else 0.3)
        tgt = rng.uniform(-0.5 * ext, 0.5 * ext, 3) if case % 7 else eye + (eye - 0.0) + 1e-3   # sometimes looking away
        tr = synth.look_at_pose(eye, target=tgt)
        if case % 5 == 0:  # a sheared, scaled pose
            tr = tr.copy()
            tr[:3, :3] = tr[:3, :3] @ (np.eye(3) + rng.uniform(-0.15, 0.15, (3, 3)))
        T = synth.cam_from_vol_f32(tr)
        ov = OracleVolume(p)
        n = ov.integrate(np.full((H, W), 1.0e6, np.float32), None, T)
        state, lo, hi = index_box(p, np.ascontiguousarray(T, np.float32).reshape(12))
        seen = np.argwhere(ov.w > 0)[:, ::-1]  # (x, y, z)
        assert len(seen) == n
        if state == 1:
            assert n == 0, (case, n)
            empty += 1
        elif state == 0:
            if n:
                assert (seen >= lo).all() and (seen <= hi).all(), (case, lo, hi, seen.min(0), seen.max(0))
                # and not absurdly loose: within a few voxels + the pyramid's own slack of the true bounding box
            vol_box = np.prod(np.clip(hi, 0, np.array(res) - 1) - np.clip(lo, 0, np.array(res) - 1) + 1)
            tight += vol_box < 0.5 * np.prod(res)
        else:
            whole += 1
    assert tight > 60 and empty > 10 and whole == 0


def test_brick_cull_predicate_never_drops_an_observable_voxel():
    """k_cull's per-block test (box_may_be_observed, evaluated here on the host for every block of the grid): a
    block holding any voxel the oracle observes must be flagged live; and the flags do cut something."""
    rng = np.random.RandomState(31)
    dropped_total = live_total = 0
    for case in range(150):
        p = capi.default_params()
        res = [int(r) for r in rng.choice([24, 33, 40, 64], 3)]
        size = [float(s) for s in rng.uniform(0.5, 4.0, 3)]
        p.res[:], p.size[:] = res, size
        W, H = 64, 48
        p.image_width, p.image_height = W, H
        f = float(rng.uniform(25.0, 90.0))
        p.fx, p.fy, p.cx, p.cy = f, f, W / 2 - 0.5, H / 2 - 0.5 + float(rng.uniform(-4, 4))
        p.min_sensor_dist = float(rng.choice([0.0, 0.3]))
        p.max_sensor_dist = float(rng.uniform(0.4, 5.0))
        ext = max(size)
        eye = rng.uniform(-1.1 * ext, 1.1 * ext, 3) * (1.0 if case % 3 else 0.3)
        tr = synth.look_at_pose(eye, target=rng.uniform(-0.5 * ext, 0.5 * ext, 3))
        if case % 6 == 0:
            tr = tr.copy()
            tr[:3, :3] = tr[:3, :3] @ (np.eye(3) + rng.uniform(-0.1, 0.1, (3, 3)))
        T = np.ascontiguousarray(synth.cam_from_vol_f32(tr), np.float32).reshape(12)
        bx, by = int(rng.choice([4, 8, 16])), int(rng.choice([1, 4, 8]))
        gx, gy = -(-res[0] // bx), -(-res[1] // by)
        flags = np.zeros((res[2], gy, gx), np.uint8)
        capi.check(capi.load().tsdf_hip_selftest_block_flags(C.byref(p), capi.as_f32p(T), bx, by, capi.as_u8p(flags)), "flags")
        ov = OracleVolume(p)
        ov.integrate(np.full((H, W), 1.0e6, np.float32), None, T)
        z, y, x = np.nonzero(ov.w > 0)
        assert flags[z, y // by, x // bx].all(), (case, int((flags[z, y // by, x // bx] == 0).sum()))
        live_total += int(flags.sum())
        dropped_total += int(flags.size - flags.sum())
    assert dropped_total > 0.2 * (live_total + dropped_total)


def test_index_box_gives_up_on_degenerate_input():
    p = capi.default_params()
    T = np.zeros(12, np.float32)              # singular pose
    assert index_box(p, T)[0] == 2
    T = synth.cam_from_vol_f32(np.eye(4)).reshape(12).astype(np.float32)
    p.max_sensor_dist = float("inf")
    assert index_box(p, T)[0] == 2            # unbounded range: no claim
    p.max_sensor_dist = -1.0
    assert index_box(p, T)[0] == 1            # nothing can have 0 < z <= -1
    p.max_sensor_dist = 3.0
    p.fx = 0.0
    assert index_box(p, T)[0] == 2


def row_intervals(p, T, planes=None):
    words = np.zeros((p.res[2], p.res[1]), np.uint32)
    capi.check(capi.load().tsdf_hip_selftest_row_intervals(C.byref(p), capi.as_f32p(T), capi.as_f32p(planes) if planes is not None else None,
                                                           words.ctypes.data_as(C.POINTER(C.c_uint32))), "row_intervals")
    lo, ln = (words & 0xffff).astype(np.int64), (words >> 16).astype(np.int64)
    x = np.arange(p.res[0])[None, None, :]
    return (x >= lo[..., None]) & (x < (lo + ln)[..., None])  # [z, y, x] mask of the voxels a LIVE launch does not mask


def test_row_intervals_contain_every_observable_voxel_and_replicate_the_reference_cull_exactly():
    """k_rows (cpu_tsdf_amd/csrc/tsdf_integrate.hip: row_interval, evaluated here on the host for every voxel row): (1)
    without planes the interval of a row is a superset of what the oracle observes with an all-valid far frame; (2) with
    the reference cull's six planes the voxels integrated -- interval AND updateVoxel's own tests -- are EXACTLY the culled
    oracle's (tsdf_volume_octree.cpp:619-652 restated in oracle/tsdf_oracle.c, itself pinned to the compiled reference),
    for principal points up to 40 % off centre, range planes through the volume, sheared poses, cameras inside."""
    rng = np.random.RandomState(47)
    cut = kept_all = bites = 0
    for case in range(260):
        p = capi.default_params()
        r = int(rng.choice([16, 32])) if case % 2 else None
        res = [r, r, r] if r else [int(v) for v in rng.choice([16, 24, 40], 3)]
        size = [float(rng.uniform(0.5, 4.0))] * 3 if r else [float(s) for s in rng.uniform(0.5, 4.0, 3)]
        p.res[:], p.size[:] = res, size
        W, H = 64, 48
        p.image_width, p.image_height = W, H
        f = float(rng.uniform(25.0, 90.0))
        off = 0.4 if case % 3 == 0 else 0.05
        p.fx, p.fy = f, f * float(rng.uniform(0.8, 1.2))
        p.cx, p.cy = W / 2 - 0.5 + float(rng.uniform(-off, off)) * W / 2, H / 2 - 0.5 + float(rng.uniform(-off, off)) * H / 2
        p.min_sensor_dist = float(rng.choice([0.0, 0.2]))
        p.max_sensor_dist = float(rng.uniform(0.3, 6.0))
        p.max_dist_pos = p.max_dist_neg = 0.03
        ext = max(size)
        eye = rng.uniform(-1.2 * ext, 1.2 * ext, 3) * (1.0 if case % 4 else 0.3)
        tr = synth.look_at_pose(eye, target=rng.uniform(-0.5 * ext, 0.5 * ext, 3))
        if case % 5 == 0:
            tr = tr.copy()
            tr[:3, :3] = tr[:3, :3] @ (np.eye(3) + rng.uniform(-0.15, 0.15, (3, 3)))
        T = np.ascontiguousarray(synth.cam_from_vol_f32(tr), np.float32).reshape(12)
        far = np.full((H, W), 1.0e6, np.float32)
        ov = OracleVolume(p)
        ov.integrate(far, None, T)
        seen = ov.w > 0
        keep = row_intervals(p, T)
        assert not (seen & ~keep).any(), (case, int((seen & ~keep).sum()))
        cut += int((~keep).sum())
        kept_all += int(keep.sum())
        # (2) the culled oracle: exactly interval-with-planes AND seen
        oc = OracleVolume(p)
        planes = oc.reference_cull_planes(tr)
        if not np.isfinite(planes).all():
            continue
        oc.integrate_culled(far, None, tr, T)
        keep_rc = row_intervals(p, T, planes)
        want = oc.w > 0
        got = keep_rc & seen
        assert np.array_equal(got, want), (case, int((got != want).sum()), int(want.sum()), int(seen.sum()))
        bites += int(want.sum() < seen.sum())
    assert cut > 0.3 * (cut + kept_all) and bites > 40, (cut, kept_all, bites)
