"""GPU tier: two frames per sweep (tsdf_hip_integrate_device2 -> k_integrate2, cpu_tsdf_amd/csrc/tsdf_integrate.hip).

The pair entry point must leave exactly the voxels of two integrateCloud calls in order -- updateVoxel applied frame by
frame (include/cpu_tsdf/impl/tsdf_volume_octree.hpp:113-218, addObservation src/lib/octree.cpp:152-163, 328-337) -- whether
one kernel sweep did both frames (both poses see the whole slab: the turntable) or it fell back to two launches.  Checked
against the CPU oracle bit for bit, counts per frame included, through weight saturation, with NaN holes and noise, and
against the band flags' consumer (marching cubes)."""
import numpy as np
import pytest
import torch

from cpu_tsdf_amd import capi, synth
from cpu_tsdf_amd.volume import MarchingCubesTSDFOctree
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32, frames, make_volume

pytestmark = pytest.mark.gpu


def device_frame(dep, col):
    """[depth | bgra] in one allocation, as tsdf_hip_integrate_device2 wants it."""
    H, W = dep.shape
    t = torch.empty((2, H, W), dtype=torch.float32, device="cuda")
    t[0].copy_(torch.from_numpy(dep))
    if col is not None:
        t[1].view(torch.uint8).view(H, W, 4).copy_(torch.from_numpy(col))
    return t


@pytest.fixture(autouse=True)
def _always_share_the_sweep(gpu):
    """These tests are about k_integrate2 itself: knob fuse2 = 2 shares the sweep wherever both poses qualify.  (The default,
    1, leaves COLOURLESS pairs to two launches of the pipelined single-frame kernel, which is faster there since round 6.)"""
    capi.set_tuning("fuse2", 2)
    yield
    capi.set_tuning("fuse2", 1)


def launch_info(vol):
    import ctypes as C
    out = (C.c_int32 * 4)()
    capi.check(capi.load().tsdf_hip_last_launch_info(vol._need(), out), "last_launch_info")
    return [int(out[0]) & 0xff, int(out[1]), int(out[2]), int(out[3]), bool(int(out[0]) & 0x100)]  # [4]: the pipelined row loop (k_integrate_p)


def run_pairs(vol, sc, n_pairs, color, total, expect_fused, count_every=2, poses=None):
    ov = OracleVolume(vol._p)
    keep = []
    fr = list(frames(sc, 2 * n_pairs, total, noise=True))
    for k in range(n_pairs):
        pair = []
        want = []
        for i, tr, dep, col in fr[2 * k:2 * k + 2]:
            if poses is not None:
                tr = poses[i]
                dep = sc.depth(tr, noise_seed=99 + i)
            dep = dep.copy()
            dep[(i * 7) % 50::53, ::3] = np.nan
            c = col if color else None
            t = device_frame(dep, c)
            keep.append(t)
            pair.append((t[0].data_ptr(), t[1].data_ptr() if color else 0, tr))
            want.append(ov.integrate_culled(dep, c, tr, synth.cam_from_vol_f32(tr)))
        fused, counts = vol.integrateCloudDevice2(pair[0], pair[1], count=(k % count_every == 0))
        assert fused == expect_fused, (k, fused, launch_info(vol))
        if counts is not None:
            assert counts == want, (k, counts, want)
    vol.synchronize()
    return ov


def compare(vol, ov):
    d, w, rgb = vol.download()
    assert_same_f32(d, ov.d, "d")
    assert_same_f32(w, ov.w, "w")
    if ov.rgb is not None:
        assert np.array_equal(rgb, ov.rgb)
    return d, w, rgb


@pytest.mark.parametrize("color,wmax,order", [(True, 100.0, 0), (False, 100.0, 0), (True, 3.0, 1), (False, 2.0, 1), (True, 255.0, 0)])
def test_two_frames_per_sweep_equal_two_integrate_calls_and_the_oracle(gpu, color, wmax, order):
    vol, sc = make_volume(96, color=color, max_weight=wmax, order=order)
    vol.reset()
    assert vol.getLayout() == capi.LAYOUT_PACKED
    ov = run_pairs(vol, sc, 5, color, total=11, expect_fused=True)
    assert launch_info(vol)[0] == 2
    d, w, _ = compare(vol, ov)
    assert (w > 0).mean() > 0.5 and (np.abs(d) < 1).sum() > 1000 and w.max() == min(wmax, 10.0)
    # the band flags k_integrate2 kept feed marching cubes' skip: the mesh equals the oracle's
    mc = MarchingCubesTSDFOctree()
    mc.setInputTSDF(vol)
    mc.setMinWeight(0.0)
    mc.setColorByRGB(color)
    mesh = mc.reconstruct()
    verts, cols = ov.march(0.0, 1 if color else 0)[:2]
    assert len(verts) > 3000
    assert_same_f32(mesh["vertices"], verts, "mesh after fused integration")
    if color:
        assert np.array_equal(mesh["rgb"], cols)
    vol.close()


def test_pairs_that_do_not_qualify_take_two_launches_with_the_same_result(gpu):
    """One pose of the pair inside the volume; the F32W layout; nx not a multiple of 4; the knob off: no fused sweep, same
    voxels, same per-frame counts."""
    cases = []
    vol, sc = make_volume(64, color=True)
    inside = [synth.turntable_pose(0, 8, sc.size), synth.look_at_pose((0.01, 0.0, -0.02), target=(0.0, 0.0, 1.0)),
              synth.turntable_pose(2, 8, sc.size), synth.turntable_pose(3, 8, sc.size)]
    cases.append(("a camera inside", vol, sc, inside, None, [False, True]))
    vol, sc = make_volume(64, color=True)
    vol.setLayout(capi.LAYOUT_F32W)
    cases.append(("F32W", vol, sc, None, None, [False, False]))
    vol, sc = make_volume(64, color=False, res3=(66, 64, 64))
    cases.append(("nx % 4", vol, sc, None, None, [False, False]))
    vol, sc = make_volume(64, color=True)
    cases.append(("knob off", vol, sc, None, ("fuse2", 0), [False, False]))  # (restored to 2 below: this module runs on fuse2 = 2)
    for name, vol, sc, poses, knob, expect in cases:
        try:
            if knob:
                capi.set_tuning(*knob)
            vol.reset()
            ov = OracleVolume(vol._p)
            color = bool(vol._p.integrate_color)
            keep = []
            for k in range(2):
                pair, want = [], []
                for i in (2 * k, 2 * k + 1):
                    tr = poses[i] if poses else synth.turntable_pose(i, 8, sc.size)
                    dep, col = sc.depth(tr, noise_seed=5 + i), sc.bgra(i) if color else None
                    t = device_frame(dep, col)
                    keep.append(t)
                    pair.append((t[0].data_ptr(), t[1].data_ptr() if color else 0, tr))
                    want.append(ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr)))
                fused, counts = vol.integrateCloudDevice2(pair[0], pair[1], count=True)
                assert fused == expect[k], (name, k, fused)
                assert counts == want, (name, k, counts, want)
            compare(vol, ov)
            vol.close()
        finally:
            if knob:
                capi.set_tuning(knob[0], 2)


def test_fused_sweep_on_a_z_slab_and_a_wide_grid(gpu):
    """A Z-slab handle (planes 5..29 of 64) on a grid several blocks wide: pointers and tables offset as in the
    single-frame launch."""
    vol, sc = make_volume(64, 160, 120, color=True, res3=(1280, 48, 64), size3=(5.0, 0.1875, 0.25), zmax=20.0)
    vol.setZSlab(5, 29)
    vol.reset()
    ov = OracleVolume(vol._p)
    keep = []
    rng = np.random.RandomState(3)
    for k in range(2):
        pair = []
        for i in (2 * k, 2 * k + 1):
            tr = synth.look_at_pose((0.3 * i - 0.4, 0.05 * i, -6.5), target=(0.0, 0.0, 0.0))
            dep = rng.uniform(6.3, 6.7, (120, 160)).astype(np.float32)
            col = rng.randint(0, 256, (120, 160, 4)).astype(np.uint8)
            t = device_frame(dep, col)
            keep.append(t)
            pair.append((t[0].data_ptr(), t[1].data_ptr(), tr))
            ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr), 5, 29)
        fused, _ = vol.integrateCloudDevice2(pair[0], pair[1])
        assert fused, launch_info(vol)
    d, w, rgb = vol.download(z0=5, nz=24)
    assert_same_f32(d, ov.d[5:29], "d")
    assert np.array_equal(w, ov.w[5:29]) and np.array_equal(rgb, ov.rgb[5:29])
    assert (w > 0).mean() > 0.5
    vol.close()


def test_frame_pairing_of_the_pipelined_host_path_changes_no_voxel(gpu):
    """tsdf_hip_set_frame_pairing (TSDFVolumeOctree.setFramePairing): pipelined integrateCloud calls are held back one
    frame and integrated two per sweep; any other call on the volume launches a waiting frame first.  An odd number of
    frames, a renderView and a counted integrateCloud in between, cameras that do and do not qualify for the fused sweep:
    every intermediate result equals the unpaired volume's and the oracle."""
    vols = []
    for pairing in (True, False):
        vol, sc = make_volume(64, color=True)
        vol.setFramePairing(pairing)
        vol.reset()
        vols.append(vol)
    ov = OracleVolume(vols[0]._p)
    poses = [synth.turntable_pose(i, 8, sc.size) for i in range(4)] + [synth.look_at_pose((0.01, 0.0, -0.02), target=(0.0, 0.0, 1.0))] + \
            [synth.turntable_pose(i, 8, sc.size) for i in (5, 6)]
    infos = []
    for i, tr in enumerate(poses):
        dep, col = sc.depth(tr, noise_seed=21 + i), sc.bgra(i)
        ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr))
        for v in vols:
            v.integrateCloud(dep, col, tr, pipelined=True)
        infos.append(launch_info(vols[0])[0])
        if i == 2:  # a query while frame 2 waits for its partner: it is launched on its own first
            a, b = (v.renderView(poses[1], 2, camera_frame=False) for v in vols)
            assert_same_f32(a, b, "renderView with a frame waiting")
            assert_same_f32(a[..., :6], ov.raycast(poses[1], 2)[..., :6], "renderView vs oracle")
    # frames 0+1 went through one sweep; frame 2 alone (the renderView); 3+4 as two launches (4 sits inside the grid); 5+6 fused
    assert infos[1] == 2 and infos[6] == 2 and infos[4] != 2, infos
    for v in vols:
        compare(v, ov)
    # one more frame: it waits ... until the download inside compare() launches it
    tr = synth.turntable_pose(7, 8, sc.size)
    dep, col = sc.depth(tr, noise_seed=99), sc.bgra(7)
    ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr))
    for v in vols:
        v.integrateCloud(dep, col, tr, pipelined=True)
    for v in vols:
        compare(v, ov)
        v.close()


def test_frame_pairing_when_the_reference_cull_decides_voxels(gpu):
    """with pairing on, frame B's cull planes reach tsdf_integrate_launch2 as the handle's OWN array, which the
    two-launch path used to overwrite with frame A's before launching B.  A principal point 60 % off centre makes the
    reference's frustum cull (1.1 x FOV about the optical axis) cut voxels, so such pairs cannot fuse and take that path, each
    frame with planes of its own (distinct poses); the paired volume must equal the unpaired one and the culled oracle."""
    vols = []
    for pairing in (True, False):
        vol, sc = make_volume(64, color=True)
        sc.cx += 0.6 * sc.width / 2
        vol.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
        vol.setFramePairing(pairing)
        vol.reset()
        vols.append(vol)
    ov = OracleVolume(vols[0]._p)
    plain = OracleVolume(vols[0]._p)   # the same frames WITHOUT the cull: shows that the cull decided voxels
    # yawed about the camera's y axis so that the grid's centre still projects to the image centre: the grid then sits OFF the
    # optical axis and one side of it leaves the cull's pyramid (bench.py --principal-offset does the same)
    psi = float(np.arctan(0.6 * (sc.width / 2) / sc.fx))
    yaw = np.eye(4)
    yaw[0, 0], yaw[0, 2], yaw[2, 0], yaw[2, 2] = np.cos(psi), np.sin(psi), -np.sin(psi), np.cos(psi)
    poses = [synth.turntable_pose(i, 7, sc.size, tilt=0.1 * i) @ yaw for i in range(6)]
    cut, infos = [], []
    for i, tr in enumerate(poses):
        dep, col = sc.depth(tr, noise_seed=31 + i), sc.bgra(i)
        n_cull = ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr))
        cut.append(plain.integrate(dep, col, synth.cam_from_vol_f32(tr)) - n_cull)
        for v in vols:
            v.integrateCloud(dep, col, tr, pipelined=True)
        infos.append(launch_info(vols[0])[0])
    assert min(cut[:5]) > 0, cut                   # the frames lost voxels to the cull (the last one none: its pair is mixed) ...
    assert all(i != 2 for i in infos), infos       # ... so no pair went through the fused sweep: two launches each
    for v in vols:
        compare(v, ov)
        v.close()
    assert not np.array_equal(ov.w, plain.w)


def test_colourless_pairs_take_the_pipelined_kernel_by_default(gpu):
    """Knob fuse2 = 1 (the default): without colour a pair is TWO launches of the software-pipelined single-frame kernel
    (k_integrate_p: faster per frame than the shared sweep since round 6), with colour ONE sweep; the voxels and the counts are
    the oracle's either way, and fuse2 = 2 (this module's setting) shares the sweep without colour too."""
    for color in (False, True):
        for knob, expect in ((1, color), (2, True)):
            capi.set_tuning("fuse2", knob)
            vol, sc = make_volume(96, color=color)
            vol.reset()
            ov = OracleVolume(vol._p)
            keep = []
            for k in range(2):
                pair, want = [], []
                for i in (2 * k, 2 * k + 1):
                    tr = synth.turntable_pose(i, 8, sc.size)
                    dep, col = sc.depth(tr, noise_seed=9 + i), sc.bgra(i) if color else None  # (96 rows: the last 64-row block is short)
                    t = device_frame(dep, col)
                    keep.append(t)
                    pair.append((t[0].data_ptr(), t[1].data_ptr() if color else 0, tr))
                    want.append(ov.integrate_culled(dep, col, tr, synth.cam_from_vol_f32(tr)))
                fused, counts = vol.integrateCloudDevice2(pair[0], pair[1], count=True)
                assert fused == expect and counts == want, (color, knob, fused, counts, want)
                info = launch_info(vol)
                assert info[0] == (2 if expect else 1) and (expect or info[4]), info  # else: the pipelined single-frame kernel
            compare(vol, ov)
            vol.close()
