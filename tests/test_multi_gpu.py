"""GPU tier: one volume over several "GPUs" behind ONE handle (tsdf_hip_create_multi / TSDFVolumeOctree.setDevices).

gpurun exposes one GPU, so the device list repeats ordinal 0: the Z-slab handles then live on the same device, but
every code path is the multi-GPU one -- frame fan-out into each slab's staging buffer, one integrate launch per slab,
halo planes by (peer) copy, concurrent per-slab meshing + Morton merge, ray hand-off with the records merged on the
first slab's device, ownership-routed sampling / block transfer / save / load.  Everything must equal a single handle
holding the whole grid, bit for bit (VERDICT r01 "Next round" #4)."""
import numpy as np
import pytest
import torch  # at collection time, i.e. BEFORE libtsdf_hip.so brings in /opt/rocm's HIP runtime (see tests/conftest.py)

from cpu_tsdf_amd import capi, synth
from cpu_tsdf_amd.volume import MarchingCubesTSDFOctree, TSDFVolumeOctree
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32

pytestmark = pytest.mark.gpu
RES, W, H, NF = 48, 160, 120, 5


def make(devices, color=True, layout=capi.LAYOUT_AUTO, max_weight=100.0):
    sc = synth.scene_a(64, W, H)  # the 64-voxel scene on a 48^3 grid of a different voxel size: slabs of 16 / 9.6 planes
    v = TSDFVolumeOctree()
    v.setResolution(RES, RES, RES)
    v.setGridSize(sc.size, sc.size, sc.size)
    v.setImageSize(W, H)
    v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    v.setSensorDistanceBounds(0.0, 3 * sc.size)
    v.setIntegrateColor(color)
    v.setWeightTruncationLimit(max_weight)
    v.setLayout(layout)
    v.setDevices(devices)
    v.reset()
    return v, sc


def fuse(vols, sc, n=NF, counts=True, pipelined=False):
    ov = OracleVolume(vols[0]._p)
    for i in range(n):
        tr = synth.turntable_pose(i, 8, sc.size)
        dep, col = sc.depth(tr, noise_seed=7 + i), sc.bgra(i)
        c = col if vols[0]._p.integrate_color else None
        want = ov.integrate(dep, c, synth.cam_from_vol_f32(tr))
        for v in vols:
            got = v.integrateCloud(dep, c, tr, count=counts, pipelined=pipelined)
            if counts:
                assert got == want
    return ov


@pytest.mark.parametrize("n_slabs", [2, 3, 5])
@pytest.mark.parametrize("layout", [capi.LAYOUT_AUTO, capi.LAYOUT_F32W])
def test_multi_handle_equals_one_handle(gpu, n_slabs, layout):
    multi, sc = make([0] * n_slabs, layout=layout)
    slabs = multi.slabs()
    assert len(slabs) == n_slabs and slabs[0][1] == 0 and slabs[-1][2] == RES and all(s[3] >= 1 for s in slabs)
    assert all(a[2] == b[1] for a, b in zip(slabs, slabs[1:]))
    ov = fuse([multi], sc)
    d, w, rgb = multi.download()
    assert_same_f32(d, ov.d, "d")
    assert_same_f32(w, ov.w, "w")
    assert np.array_equal(rgb, ov.rgb)
    # a block that straddles two slab boundaries
    z0, z1 = slabs[0][2] - 2, min(RES, slabs[1][2] + 3)
    bd, bw, brgb = multi.download(3, 5, z0, 20, 17, z1 - z0)
    assert np.array_equal(bd, ov.d[z0:z1, 5:22, 3:23]) and np.array_equal(bw, ov.w[z0:z1, 5:22, 3:23])
    # marching cubes: count, order, vertex bits, colours
    mc = MarchingCubesTSDFOctree()
    mc.setInputTSDF(multi)
    for wmin, by_rgb, by_conf, mode in [(2.0, True, False, 1), (0.0, False, True, 2), (2.5, False, False, 0)]:
        mc.setMinWeight(wmin)
        mc.setColorByRGB(by_rgb)
        mc.setColorByConfidence(by_conf)
        mesh = mc.reconstruct(want_cells=True)
        v2, c2, cells2 = ov.march(wmin, mode)
        assert len(cells2) > 500
        assert np.array_equal(mesh["cells"], cells2)
        assert_same_f32(mesh["vertices"], v2, "mesh vertices")
        if mode:
            assert np.array_equal(mesh["rgb"], c2)
    # renderView (ray hand-off between the slabs), both frames, with and without downsampling
    for tr, ds in [(synth.turntable_pose(1, 8, sc.size), 1), (synth.look_at_pose((0.05, -0.3, -0.2)), 2),
                   (synth.look_at_pose((0.0, 0.02, -0.6 * sc.size), target=(0.0, 0.0, 1.0)), 1)]:
        got = multi.renderView(tr, ds, camera_frame=False)
        want = ov.raycast(tr, ds)
        assert np.isfinite(want[..., 0]).sum() > 50
        assert_same_f32(got, want, f"renderView ds={ds}")
    # getFxn / gradient / Hessian, crowded around the slab seams
    rng = np.random.RandomState(3)
    pts = rng.uniform(-0.5 * sc.size, 0.5 * sc.size, (3000, 3)).astype(np.float32)
    ok, val, grad, hess = multi.sample(pts)
    ok2, val2, grad2, hess2 = ov.sample(pts)
    assert np.array_equal(ok, ok2) and ok.sum() > 2000
    assert_same_f32(val[ok], val2[ok], "getFxn")
    assert_same_f32(grad[ok], grad2[ok], "gradient")
    assert_same_f32(hess[ok], hess2[ok], "Hessian")
    multi.close()


def test_multi_handle_more_frames_after_queries_and_upload(gpu):
    """Halo freshness: integrate -> mesh -> integrate -> render -> upload -> sample must each see current planes."""
    multi, sc = make([0, 0, 0], color=False)
    one, _ = make(None, color=False)
    mc = MarchingCubesTSDFOctree()
    for rnd in range(3):
        for i in range(2):
            tr = synth.turntable_pose(2 * rnd + i, 8, sc.size)
            dep = sc.depth(tr)
            multi.integrateCloud(dep, None, tr)
            one.integrateCloud(dep, None, tr)
        meshes = []
        for v in (multi, one):
            mc.setInputTSDF(v)
            mc.setMinWeight(1.0)
            meshes.append(mc.reconstruct())
        assert_same_f32(meshes[0]["vertices"], meshes[1]["vertices"], f"mesh after round {rnd}")
        tr = synth.turntable_pose(rnd, 8, sc.size, tilt=0.2)
        assert_same_f32(multi.renderView(tr), one.renderView(tr), f"renderView after round {rnd}")
    d, w, _ = one.download()
    d2 = d.copy()
    d2[20:30] = np.float32(0.25)
    multi.upload(d2, w, None)
    one.upload(d2, w, None)
    pts = np.random.RandomState(9).uniform(-0.4 * sc.size, 0.4 * sc.size, (2000, 3)).astype(np.float32)
    a, b = multi.sample(pts), one.sample(pts)
    assert np.array_equal(a[0], b[0])
    assert_same_f32(a[1][a[0]], b[1][b[0]], "getFxn after upload")
    multi.close()
    one.close()


def test_multi_handle_pipelined_host_frames_and_unorganized_ingest(gpu):
    multi, sc = make([0, 0])
    ov = fuse([multi], sc, counts=False, pipelined=True)
    multi.synchronize()
    d, w, rgb = multi.download()
    assert_same_f32(d, ov.d, "d (pipelined)")
    assert np.array_equal(w, ov.w) and np.array_equal(rgb, ov.rgb)
    # the ingest path: the z-buffer runs on the first slab's GPU, its frame fans out to the others
    multi.reset()
    one, _ = make(None)
    tr = synth.turntable_pose(0, 8, sc.size)
    dep = sc.depth(tr)
    u, vv = np.meshgrid(np.arange(W), np.arange(H))
    z = np.where(np.isfinite(dep), dep, 0).astype(np.float32)
    xyz = np.stack([(u - sc.cx) / sc.fx * z, (vv - sc.cy) / sc.fy * z, z], -1).reshape(-1, 3).astype(np.float32)
    col = sc.bgra(0).reshape(-1, 4)
    for v in (multi, one):
        v.organize(xyz, col, zero_nans=True, fetch=False)
        v.integrateStaged(tr)
        v.organize(xyz[::3], col[::3], zero_nans=True)  # (with the host copies: synchronous form)
        assert v.integrateStaged(synth.turntable_pose(1, 8, sc.size), count=True) > 0
    a, b = multi.download(), one.download()
    assert_same_f32(a[0], b[0], "d after integrateUnorganized")
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    multi.close()
    one.close()


def test_multi_handle_save_load_roundtrip(gpu, tmp_path):
    multi, sc = make([0, 0, 0], max_weight=3.0)
    v = TSDFVolumeOctree()  # the .vol format needs a cubic power-of-two grid: 32^3 here
    s = synth.scene_a(32, W, H)
    for vol in (v,):
        vol.setResolution(32, 32, 32)
        vol.setGridSize(s.size, s.size, s.size)
        vol.setImageSize(W, H)
        vol.setCameraIntrinsics(s.fx, s.fy, s.cx, s.cy)
        vol.setSensorDistanceBounds(0.0, 3 * s.size)
        vol.setIntegrateColor(True)
        vol.setDevices([0, 0, 0])
        vol.reset()
    one = TSDFVolumeOctree()
    one.setResolution(32, 32, 32)
    one.setGridSize(s.size, s.size, s.size)
    one.setImageSize(W, H)
    one.setCameraIntrinsics(s.fx, s.fy, s.cx, s.cy)
    one.setSensorDistanceBounds(0.0, 3 * s.size)
    one.setIntegrateColor(True)
    one.reset()
    for i in range(4):
        tr = synth.turntable_pose(i, 8, s.size)
        for vol in (v, one):
            vol.integrateCloud(s.depth(tr), s.bgra(i), tr)
    pa, pb = str(tmp_path / "multi.vol"), str(tmp_path / "one.vol")
    v.save(pa)
    one.save(pb)
    assert open(pa, "rb").read() == open(pb, "rb").read()
    back = TSDFVolumeOctree()
    back.setDevices([0, 0])
    back.load(pb)
    assert len(back.slabs()) == 2
    a, b = back.download(), one.download()
    assert_same_f32(a[0], b[0], "d after load into two slabs")
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for vol in (multi, v, one, back):
        vol.close()


def test_single_device_entry_points_refuse_a_multi_handle(gpu):
    multi, _ = make([0, 0])
    lib, h = capi.load(), multi._need()
    assert lib.tsdf_hip_set_stream(h, None) == capi.E_UNSUPPORTED
    assert lib.tsdf_hip_get_planes_device(h, 0, 1, None, None, None) == capi.E_UNSUPPORTED
    assert b"multi-GPU set" in lib.tsdf_hip_last_error()
    multi.close()


def _devices(n):
    """Device lists for n slabs: ordinal 0 repeated (always), plus distinct ordinals round-robin where the box has more
    than one GPU (cross-DEVICE ordering -- peer copies, events between devices -- has only ever run with
    repeated ordinals on this project's one-GPU boxes; this variant runs wherever it can)."""
    out = [[0] * n]
    ndev = capi.load().tsdf_hip_device_count()
    if ndev > 1:
        out.append([k % ndev for k in range(n)])
    return out


@pytest.mark.parametrize("n_slabs", [2, 8])
def test_render_view_hands_rays_over_in_compact_lists(gpu, n_slabs):
    """renderView on a multi handle moves hand-off records (96 B per seam crossing) and finished rays (36 B each, to the
    first slab), never the image; and equals one handle at 8 slabs as well, at an image larger than the volume test's."""
    for devices in _devices(n_slabs):
        multi, sc = make(devices)
        one, _ = make(None)
        for i in range(3):
            tr = synth.turntable_pose(i, 8, sc.size)
            for v in (multi, one):
                v.integrateCloud(sc.depth(tr), sc.bgra(i), tr)
        n_rays = W * H
        for tr in (synth.look_at_pose((0.02, 0.01, -0.9 * sc.size)), synth.turntable_pose(1, 8, sc.size, tilt=0.3),
                   synth.look_at_pose((-0.9 * sc.size, 0.0, 0.013), target=(0.0, 0.01, -0.02))):
            for camera in (False, True):
                got, want = multi.renderView(tr, 1, camera_frame=camera), one.renderView(tr, 1, camera_frame=camera)
                assert_same_f32(got, want, f"renderView {devices}")
            rounds, handed, moved, waits = multi.renderStats()
            assert 1 <= rounds <= 2 * n_slabs + 4 and waits == n_slabs * rounds
            assert 96 * handed <= moved <= 96 * handed + 36 * n_rays and (moved - 96 * handed) % 36 == 0
            # the full-image protocol this replaced moved 2 (n - 1) images of 96 B records PER ROUND
            assert moved < 2 * (n_slabs - 1) * n_rays * 96
        assert multi.renderStats()[1] > 0  # the last view runs along the seams: rays do change slabs
        multi.close()
        one.close()


def test_device_frames_back_to_back_without_counts(gpu):
    """with n_observed == NULL nothing synchronises between frames, so the source of the frame
    fan-out (slab 0's staging buffer after tsdf_hip_organize, or the caller's device buffer) must be ordered AFTER the
    other slabs' copies of the previous frame by events.  Every slab has its own stream (also on one device), so a
    missing dependency shows up here as a torn frame: 24 frames back to back, slabs of unequal work."""
    for devices in _devices(4):
        multi, sc = make(devices)
        one, _ = make(None)
        u, vv = np.meshgrid(np.arange(W), np.arange(H))
        n = 24
        for i in range(n):  # the ingest path: z-buffer on slab 0, fan-out, integrate; no count, no synchronisation
            tr = synth.turntable_pose(i, n, sc.size, tilt=0.1)
            dep = sc.depth(tr, noise_seed=100 + i)
            z = np.where(np.isfinite(dep), dep, 0).astype(np.float32)
            xyz = np.stack([(u - sc.cx) / sc.fx * z, (vv - sc.cy) / sc.fy * z, z], -1).reshape(-1, 3).astype(np.float32)
            col = sc.bgra(i).reshape(-1, 4)
            for v in (multi, one):
                v.organize(xyz, col, zero_nans=True, fetch=False)
                v.integrateStaged(tr)
        a, b = multi.download(), one.download()
        assert_same_f32(a[0], b[0], f"d after {n} staged frames, devices {devices}")
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        # caller-owned device frames (one buffer per frame, all complete before the first call)
        multi.reset()
        one.reset()
        frames = torch.empty((n, 2, H, W), dtype=torch.float32, device="cuda:0")
        poses = [synth.turntable_pose(i, n, sc.size, tilt=-0.1) for i in range(n)]
        for i, tr in enumerate(poses):
            frames[i, 0].copy_(torch.from_numpy(sc.depth(tr, noise_seed=300 + i)))
            frames[i, 1].view(torch.uint8).view(H, W, 4).copy_(torch.from_numpy(sc.bgra(i)))
        torch.cuda.synchronize()
        for i, tr in enumerate(poses):
            for v in (multi, one):
                v.integrateCloudDevice(frames[i, 0].data_ptr(), frames[i, 1].data_ptr(), tr)
        multi.synchronize()
        one.synchronize()
        a, b = multi.download(), one.download()
        assert_same_f32(a[0], b[0], f"d after {n} device frames, devices {devices}")
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        multi.close()
        one.close()


def test_counts_are_reduced_after_every_slab_has_launched(gpu):
    """n_observed on a multi handle = the sum over the slabs, each slab's counters read after ALL slabs were launched;
    tsdf_hip_last_count_detail reports the set's totals; the per-slab kernel timing hook brackets every launch."""
    import ctypes as C
    multi, sc = make([0, 0, 0])
    one, _ = make(None)
    lib = capi.load()
    capi.check(lib.tsdf_hip_multi_timing(multi._need(), 1), "timing")
    for i in range(3):
        tr = synth.turntable_pose(i, 8, sc.size)
        a = multi.integrateCloud(sc.depth(tr), sc.bgra(i), tr, count=True)
        b = one.integrateCloud(sc.depth(tr), sc.bgra(i), tr, count=True)
        da, db = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
        capi.check(lib.tsdf_hip_last_count_detail(multi._need(), da), "detail")
        capi.check(lib.tsdf_hip_last_count_detail(one._need(), db), "detail")
        assert a == b == da[0] == db[0] and da[1] == db[1] > 0
    for k in range(3):
        ms, cnt = C.c_float(0), C.c_int32(0)
        capi.check(lib.tsdf_hip_multi_kernel_ms(multi._need(), k, C.byref(ms), C.byref(cnt)), "kernel_ms")
        assert cnt.value == 3 and ms.value > 0
    assert lib.tsdf_hip_multi_timing(one._need(), 1) == capi.E_INVALID
    multi.close()
    one.close()


def test_slabs_without_peer_access_go_through_the_host_relay(gpu, monkeypatch):
    """VERDICT r03 next #6c: a GPU pair the driver refuses peer access to must not hang a multi-GPU set -- its frames, halo
    planes and ray records travel through a pinned relay buffer on the host (tsdf_multi_copy, tsdf_multi.hip).
    TSDF_HIP_NO_PEER=1 sends EVERY cross-slab copy that way, so one GPU can test it: device-pointer frames (fan-out from the
    caller's buffer), organize + integrateStaged (fan-out from slab 0's staging buffer), the mesh (halo planes), renderView
    (ray hand-off) and sampling all equal a single handle, and the relay really carried the bytes."""
    import ctypes as C
    monkeypatch.setenv("TSDF_HIP_NO_PEER", "1")
    multi, sc = make([0, 0, 0])
    monkeypatch.delenv("TSDF_HIP_NO_PEER")
    one, _ = make(None)
    keep = []
    for i in range(4):
        tr = synth.turntable_pose(i, 8, sc.size)
        dep, col = sc.depth(tr, noise_seed=7 + i), sc.bgra(i)
        t = torch.empty((2, H, W), dtype=torch.float32, device="cuda")
        t[0].copy_(torch.from_numpy(dep))
        t[1].view(torch.uint8).view(H, W, 4).copy_(torch.from_numpy(col))
        torch.cuda.synchronize()
        keep.append(t)
        n_multi = multi.integrateCloudDevice(t[0].data_ptr(), t[1].data_ptr(), tr, count=True)
        assert n_multi == one.integrateCloud(dep, col, tr, count=True)
    stats = (C.c_uint64 * 3)()
    capi.check(capi.load().tsdf_hip_multi_link_stats(multi._need(), stats), "link_stats")
    assert stats[1] == 1 and stats[2] >= 4 * 3 * 2 * W * H * 4  # every frame, to every slab, depth + colour
    d, w, rgb = multi.download()
    d1, w1, rgb1 = one.download()
    assert_same_f32(d, d1, "d")
    assert np.array_equal(w, w1) and np.array_equal(rgb, rgb1)
    mc = MarchingCubesTSDFOctree()
    meshes = []
    for v in (multi, one):
        mc.setInputTSDF(v)
        mc.setMinWeight(1.0)
        mc.setColorByRGB(True)
        meshes.append(mc.reconstruct())
    assert len(meshes[1]["vertices"]) > 1000
    assert_same_f32(meshes[0]["vertices"], meshes[1]["vertices"], "mesh")
    assert np.array_equal(meshes[0]["rgb"], meshes[1]["rgb"])
    before = int(stats[2])
    tr = synth.turntable_pose(1, 8, sc.size, tilt=0.2)
    assert_same_f32(multi.renderView(tr), one.renderView(tr), "renderView")
    capi.check(capi.load().tsdf_hip_multi_link_stats(multi._need(), stats), "link_stats")
    assert stats[2] > before  # halo planes and ray records went through the relay too
    rng = np.random.RandomState(5)
    pts = rng.uniform(-0.5 * sc.size, 0.5 * sc.size, (2000, 3)).astype(np.float32)
    a, b = multi.sample(pts), one.sample(pts)
    assert np.array_equal(a[0], b[0]) and a[0].sum() > 1000
    assert_same_f32(a[1][a[0]], b[1][b[0]], "getFxn")
    multi.close()
    one.close()


@pytest.mark.parametrize("color", [True, False])
def test_frame_pairing_on_a_multi_handle_equals_frame_by_frame(gpu, color):
    """tsdf_hip_set_frame_pairing and tsdf_hip_integrate_device2 on a multi-GPU set.  Every slab takes
    the frames through its own ring and sweeps ONCE for a pair where both poses see all of that slab; a frame waiting for its
    partner is launched by any other call on the set.  Host frames (pipelined, pairing on: pair, parked + getFxn, parked +
    download), device pairs (integrateCloudDevice2, counts included) and single frames in between -- against one handle
    going frame by frame and the oracle, voxel for voxel."""
    for devices in ([0, 0, 0], [0, 0]):
        multi, sc = make(devices, color=color)
        one, _ = make(None, color=color)
        ov = OracleVolume(multi._p)
        multi.setFramePairing(True)
        poses = [synth.turntable_pose(i, 12, sc.size, tilt=0.05 * i) for i in range(9)]
        deps = [sc.depth(tr, noise_seed=500 + i) for i, tr in enumerate(poses)]
        cols = [sc.bgra(i) if color else None for i in range(9)]

        def truth(i):
            one.integrateCloud(deps[i], cols[i], poses[i])
            return ov.integrate(deps[i], cols[i], synth.cam_from_vol_f32(poses[i]))
        pts = np.random.RandomState(4).uniform(-0.1, 0.1, (64, 3)).astype(np.float32)
        for i in (0, 1, 2):   # a pair, then a frame that waits ...
            multi.integrateCloud(deps[i], cols[i], poses[i], pipelined=True)
            truth(i)
        ok_m, val_m = multi.getFxn(pts)   # ... until getFxn launches it
        ok_o, val_o, _, _ = ov.sample(pts)
        assert np.array_equal(ok_m.astype(bool), ok_o) and np.array_equal(val_m[ok_o], val_o[ok_o])
        multi.integrateCloud(deps[3], cols[3], poses[3], pipelined=True)   # parked again; the device pair below goes AFTER it
        truth(3)
        fr = torch.empty((2, 2, H, W), dtype=torch.float32, device="cuda:0")
        for k, i in enumerate((4, 5)):
            fr[k, 0].copy_(torch.from_numpy(deps[i]))
            if color:
                fr[k, 1].view(torch.uint8).view(H, W, 4).copy_(torch.from_numpy(cols[i]))
        torch.cuda.synchronize()
        fused, counts = multi.integrateCloudDevice2((fr[0, 0].data_ptr(), fr[0, 1].data_ptr() if color else 0, poses[4]),
                                                    (fr[1, 0].data_ptr(), fr[1, 1].data_ptr() if color else 0, poses[5]), count=True)
        assert counts == [truth(4), truth(5)], counts
        assert fused == color  # the turntable sees every slab whole: with colour every slab swept ONCE for the pair; without,
        #                        two launches of the pipelined single-frame kernel are the faster way (knob fuse2 = 1)
        got = multi.integrateCloud(deps[6], cols[6], poses[6], count=True)   # a synchronous counted frame in between
        assert got == truth(6)
        multi.integrateCloud(deps[7], cols[7], poses[7], pipelined=True)
        truth(7)
        multi.setFramePairing(False)      # switching it off launches the frame that was waiting
        multi.integrateCloud(deps[8], cols[8], poses[8], pipelined=True)
        truth(8)
        a, b = multi.download(), one.download()
        assert_same_f32(a[0], ov.d, f"d vs oracle, devices {devices}")
        assert np.array_equal(a[1], ov.w) and (not color or np.array_equal(a[2], ov.rgb))
        assert_same_f32(a[0], b[0], "d vs one handle")
        assert np.array_equal(a[1], b[1])
        multi.close()
        one.close()


def test_frame_pairs_through_the_host_relay(gpu, monkeypatch):
    """Frame pairing where no slab may read another's memory (TSDF_HIP_NO_PEER=1): the two frames of a device pair and of a
    host pair reach every slab through the pinned relay (tsdf_multi_copy) before that slab's single sweep -- same voxels as
    the oracle, and the relay carried both frames to every slab."""
    import ctypes as C
    monkeypatch.setenv("TSDF_HIP_NO_PEER", "1")
    multi, sc = make([0, 0, 0], color=True)
    monkeypatch.delenv("TSDF_HIP_NO_PEER")
    ov = OracleVolume(multi._p)
    multi.setFramePairing(True)
    poses = [synth.turntable_pose(i, 12, sc.size) for i in range(4)]
    deps = [sc.depth(tr, noise_seed=900 + i) for i, tr in enumerate(poses)]
    cols = [sc.bgra(i) for i in range(4)]
    fr = torch.empty((2, 2, H, W), dtype=torch.float32, device="cuda:0")
    for k in range(2):
        fr[k, 0].copy_(torch.from_numpy(deps[k]))
        fr[k, 1].view(torch.uint8).view(H, W, 4).copy_(torch.from_numpy(cols[k]))
    torch.cuda.synchronize()
    fused, counts = multi.integrateCloudDevice2((fr[0, 0].data_ptr(), fr[0, 1].data_ptr(), poses[0]),
                                                (fr[1, 0].data_ptr(), fr[1, 1].data_ptr(), poses[1]), count=True)
    want = [ov.integrate(deps[k], cols[k], synth.cam_from_vol_f32(poses[k])) for k in range(2)]
    assert fused and counts == want
    stats = (C.c_uint64 * 3)()
    capi.check(capi.load().tsdf_hip_multi_link_stats(multi._need(), stats), "link_stats")
    assert stats[1] == 1 and stats[2] >= 2 * 3 * 2 * W * H * 4  # both frames, to every slab, depth + colour
    for k in (2, 3):   # a host pair: pinned slot -> each slab, no relay needed, same single sweep
        multi.integrateCloud(deps[k], cols[k], poses[k], pipelined=True)
        ov.integrate(deps[k], cols[k], synth.cam_from_vol_f32(poses[k]))
    d, w, rgb = multi.download()
    assert_same_f32(d, ov.d, "d vs oracle")
    assert np.array_equal(w, ov.w) and np.array_equal(rgb, ov.rgb)
    multi.close()
