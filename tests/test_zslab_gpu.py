"""GPU tier: the Z-slab pieces on real device memory.  gpurun exposes one GPU, so the two "ranks" are two
slab handles on the same device with the halo plane moved through the same C-ABI calls and torch CUDA
tensors the RCCL path uses (tsdf_hip_get/set_planes_device); plus ZSlabVolume end-to-end at world_size 1."""
import numpy as np
import pytest
import torch

from cpu_tsdf_amd import synth
from cpu_tsdf_amd.zslab import HipSlab, ZSlabVolume, morton_x_major, render_halo, slab_range
from oracle.oracle import OracleVolume
from tests.common import assert_same_f32
from tests.test_zslab_gloo import H, NF, RES, W, configure, views

pytestmark = pytest.mark.gpu


def truth():
    from tests.fake_slab import _Cfg
    cfg = _Cfg()
    configure(cfg)
    ov = OracleVolume(cfg._p)
    sc = synth.scene_a(RES, W, H)
    for i in range(NF):
        tr = synth.turntable_pose(i, 8, sc.size)
        ov.integrate(sc.depth(tr), sc.bgra(i), synth.cam_from_vol_f32(tr))
    return ov, sc


def test_two_slab_handles_with_halo_exchange_equal_one_volume(gpu):
    ov, sc = truth()
    cut = 13
    a = HipSlab(configure, 0, cut, RES, 0)
    b = HipSlab(configure, cut, RES, RES, 0)
    fd, fc = a.frame_buffers()
    for i in range(NF):
        tr = synth.turntable_pose(i, 8, sc.size)
        fd.copy_(torch.from_numpy(sc.depth(tr)))
        fc.copy_(torch.from_numpy(sc.bgra(i)))
        a.integrate_tensor(fd, fc, tr)
        b.integrate_tensor(fd, fc, tr)
    # halo: b's first plane -> a's upper halo, as device tensors
    planes = b.get_planes(cut, 1)
    b.synchronize()
    a.set_planes(cut, *planes)
    parts = [a.march(1.0, True, False), b.march(1.0, True, False)]
    cells = np.concatenate([p["cells"] for p in parts])
    order = np.argsort(morton_x_major(cells), kind="stable")
    verts = np.concatenate([p["vertices"].reshape(-1, 3, 3) for p in parts])[order].reshape(-1, 3)
    rgb = np.concatenate([p["rgb"].reshape(-1, 3, 3) for p in parts])[order].reshape(-1, 3)
    v2, c2, cells2 = ov.march(1.0, 1)
    assert len(parts[0]["cells"]) > 0 and len(parts[1]["cells"]) > 0
    assert np.array_equal(cells[order], cells2)
    assert_same_f32(verts, v2, "merged mesh")
    assert np.array_equal(rgb, c2)
    # plane round trip incl. colour packing
    d, w, c = a.get_planes(3, 2)
    a.synchronize()
    assert np.array_equal(d.cpu().numpy(), ov.d[3:5]) and np.array_equal(w.cpu().numpy(), ov.w[3:5])
    cc = c.cpu().numpy()
    assert np.array_equal(np.stack([cc & 255, (cc >> 8) & 255, (cc >> 16) & 255], -1).astype(np.uint8), ov.rgb[3:5])
    a.close()
    b.close()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_ray_handoff_between_slab_handles_equals_one_volume(gpu, world):
    """renderView across Z-slabs on real device memory: `world` slab handles on the one GPU, halos refreshed
    through get/set_planes_device, rays handed off with tsdf_hip_raycast_begin/advance, deltas merged by an
    integer sum exactly as ZSlabVolume does with the RCCL all-reduce.  Must equal the whole-volume kernel
    bit for bit."""
    from cpu_tsdf_amd.volume import TSDFVolumeOctree
    sc = synth.scene_a(RES, W, H)
    halo = render_halo(configure)
    assert 8 <= halo <= 16
    whole = TSDFVolumeOctree()
    configure(whole)
    whole.reset()
    cuts = [slab_range(RES, world, r) for r in range(world)]
    slabs = [HipSlab(configure, zb, ze, RES, 0, halo=halo) for zb, ze in cuts]
    fd, fc = slabs[0].frame_buffers()
    for i in range(NF):
        tr = synth.turntable_pose(i, 8, sc.size)
        dep, col = sc.depth(tr), sc.bgra(i)
        whole.integrateCloud(dep, col, tr)
        fd.copy_(torch.from_numpy(dep))
        fc.copy_(torch.from_numpy(col))
        for s in slabs:
            s.integrate_tensor(fd, fc, tr)
    # halo refresh: every slab takes its neighbours' boundary planes (clipped to what they own and to the grid)
    for r, s in enumerate(slabs):
        for nb in (r - 1, r + 1):
            if 0 <= nb < world:
                zb, ze = cuts[nb]
                n = min(halo, ze - zb)
                z0 = ze - n if nb < r else zb
                planes = slabs[nb].get_planes(z0, n)
                slabs[nb].synchronize()
                s.set_planes(z0, *planes)
    total_rounds = 0
    for k, tr in enumerate(views(sc.size)):
        ds = 1 + (k == 1)
        want = whole.renderView(tr, ds, camera_frame=False)
        state = slabs[0].ray_begin(tr, ds)
        for rounds in range(1, 2 * world + 5):
            delta = sum(s.ray_advance(tr, ds, state, r, world) for r, s in enumerate(slabs))
            state = torch.where(delta[:, :1] != 0, delta, state)
            if int((state[:, 0] == 1).sum()) == 0:
                break
        assert int((state[:, 0] == 2).sum()) == state.shape[0]
        assert rounds <= world + 2   # + 1: a hit whose extrapolated point lies in another slab finishes at its owner
        total_rounds += rounds
        have = state[:, 16:24].contiguous().cpu().numpy().view(np.float32).reshape(want.shape)
        assert_same_f32(have, want, f"view {k} world {world}")
        # the compact-list form (what exchange="p2p" runs on every rank): each slab advances only the records
        # routed to it; finished records leave their outputs, suspended ones move to the owner of their next plane
        state = slabs[0].ray_begin(tr, ds)
        n = state.shape[0]
        ids = torch.arange(n, device=state.device)
        lists = [state[(ids % world) == r].contiguous() for r in range(world)]
        out = torch.zeros((n, 8), dtype=torch.int32, device=state.device)
        ends = torch.tensor([ze for _, ze in cuts], device=state.device)
        moved = 0
        for _ in range(2 * world + 5):
            nxt = [[] for _ in range(world)]
            for r, s in enumerate(slabs):
                if not lists[r].shape[0]:
                    continue
                rec = s.ray_advance_list(tr, ds, lists[r], r, world)
                fin = rec[rec[:, 0] == 2]
                out[fin[:, 11].long()] = fin[:, 16:24]
                sus = rec[rec[:, 0] == 1]
                dest = torch.bucketize(sus[:, 1].long(), ends, right=True)
                assert not bool((dest == r).any())
                for q in range(world):
                    if bool((dest == q).any()):
                        nxt[q].append(sus[dest == q])
                        moved += int((dest == q).sum())
            lists = [torch.cat(p).contiguous() if p else state[:0] for p in nxt]
            if not any(l.shape[0] for l in lists):
                break
        assert not any(l.shape[0] for l in lists)
        assert_same_f32(out.cpu().numpy().view(np.float32).reshape(want.shape), want, f"list hand-off, view {k} world {world}")
        assert 0 < moved < n * world
    assert total_rounds > len(views(sc.size))  # rays really crossed slabs
    for s in slabs:
        s.close()
    whole.close()


def test_ray_handoff_finishes_far_extrapolated_hits_at_their_owner(gpu):
    """t_star = t + step * (-1 + |d0 / (d0 - d1)|) (tsdf_volume_octree.cpp:389) lands arbitrarily far ahead when the two
    trilinear samples are nearly equal; the normal's samples then need planes far outside the halo of the slab that
    found the crossing.  This small configuration (32^3, two slabs of 16 planes, halo 12) has such a ray: its hit
    point sits 12 planes past the first slab.  The record travels once more (finish flag) and the image still
    equals the whole-volume kernel's bit for bit."""
    from cpu_tsdf_amd.volume import TSDFVolumeOctree
    res, w, h = 32, 80, 60
    sc = synth.scene_a(res, w, h)

    def conf(v):
        v.setResolution(res, res, res)
        v.setGridSize(sc.size, sc.size, sc.size)
        v.setImageSize(w, h)
        v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
        v.setSensorDistanceBounds(0.0, 3 * sc.size)
        v.setIntegrateColor(True)
    halo = render_halo(conf)
    whole = TSDFVolumeOctree()
    conf(whole)
    whole.reset()
    cuts = [slab_range(res, 2, r) for r in range(2)]
    slabs = [HipSlab(conf, zb, ze, res, 0, halo=halo) for zb, ze in cuts]
    fd, fc = slabs[0].frame_buffers()
    for i in range(3):
        tr = synth.turntable_pose(i, 3, sc.size)
        dep, col = sc.depth(tr), sc.bgra(i)
        whole.integrateCloud(dep, col, tr)
        fd.copy_(torch.from_numpy(dep))
        fc.copy_(torch.from_numpy(col))
        for s in slabs:
            s.integrate_tensor(fd, fc, tr)
    for r, s in enumerate(slabs):
        nb = 1 - r
        zb, ze = cuts[nb]
        n = min(halo, ze - zb)
        z0 = ze - n if nb < r else zb
        planes = slabs[nb].get_planes(z0, n)
        slabs[nb].synchronize()
        s.set_planes(z0, *planes)
    tr = synth.turntable_pose(0, 8, sc.size)
    want = whole.renderView(tr, 1, camera_frame=False)
    state = slabs[0].ray_begin(tr, 1)
    finish_hops = 0
    for rounds in range(1, 9):
        delta = sum(s.ray_advance(tr, 1, state, r, 2) for r, s in enumerate(slabs))
        finish_hops += int(((delta[:, 0] == 1) & (delta[:, 12] == 1)).sum())
        state = torch.where(delta[:, :1] != 0, delta, state)
        if int((state[:, 0] == 1).sum()) == 0:
            break
    assert int((state[:, 0] == 2).sum()) == state.shape[0] and finish_hops >= 1
    have = state[:, 16:24].contiguous().cpu().numpy().view(np.float32).reshape(want.shape)
    assert_same_f32(have, want, "far-extrapolated hit")
    for s in slabs:
        s.close()
    whole.close()


def test_ray_handoff_refuses_a_halo_that_is_too_small(gpu):
    from cpu_tsdf_amd.capi import TsdfHipError
    sc = synth.scene_a(RES, W, H)
    cuts = [slab_range(RES, 2, r) for r in range(2)]
    slabs = [HipSlab(configure, zb, ze, RES, 0, halo=1) for zb, ze in cuts]
    fd, fc = slabs[0].frame_buffers()
    for i in range(NF):
        tr = synth.turntable_pose(i, 8, sc.size)
        fd.copy_(torch.from_numpy(sc.depth(tr)))
        fc.copy_(torch.from_numpy(sc.bgra(i)))
        for s in slabs:
            s.integrate_tensor(fd, fc, tr)
    tr = views(sc.size)[0]
    state = slabs[0].ray_begin(tr, 1)
    with pytest.raises(TsdfHipError):
        for _ in range(4):
            delta = sum(s.ray_advance(tr, 1, state, r, 2) for r, s in enumerate(slabs))
            state = torch.where(delta[:, :1] != 0, delta, state)
    for s in slabs:
        s.close()


@pytest.mark.parametrize("world", [2, 3])
def test_sample_across_slab_handles_with_stale_wide_halo(gpu, world):
    """getFxn / gradient / Hessian over Z-slab handles that keep a WIDE halo (the renderView one) of which only the
    first plane is fresh: exactly ZSlabVolume.sample's state after integrating since the last render (ADVICE r01:
    a handle used to answer from any allocated plane pair, and the stale answer won the merge).  A handle must answer
    only for points whose lower-corner plane it owns; merged exactly as ZSlabVolume.sample merges, the union must
    equal the single volume."""
    ov, sc = truth()
    halo = render_halo(configure)
    cuts = [slab_range(RES, world, r) for r in range(world)]
    slabs = [HipSlab(configure, zb, ze, RES, 0, halo=halo) for zb, ze in cuts]
    fd, fc = slabs[0].frame_buffers()
    for i in range(NF):
        tr = synth.turntable_pose(i, 8, sc.size)
        fd.copy_(torch.from_numpy(sc.depth(tr)))
        fc.copy_(torch.from_numpy(sc.bgra(i)))
        for s in slabs:
            s.integrate_tensor(fd, fc, tr)
    for r in range(world - 1):  # exchange_halo(planes=1, both=False): plane z_end from its owner, nothing else
        planes = slabs[r + 1].get_planes(cuts[r][1], 1)
        slabs[r + 1].synchronize()
        slabs[r].set_planes(cuts[r][1], *planes)
    rng = np.random.RandomState(7)
    pts = rng.uniform(-0.5 * sc.size, 0.5 * sc.size, (4000, 3)).astype(np.float32)
    vs = sc.size / RES
    for zb, ze in cuts[:-1]:  # crowd the slab seams, where the stale halo planes sit
        seam = rng.uniform(-0.45 * sc.size, 0.45 * sc.size, (1500, 3)).astype(np.float32)
        seam[:, 2] = (-0.5 * sc.size + ze * vs + rng.uniform(-halo * vs, halo * vs, 1500)).astype(np.float32)
        pts = np.concatenate([pts, seam])
    parts = [s.sample(pts) for s in slabs]
    ok, val, grad, hess = [np.array(a) for a in parts[0]]
    owners = ok.astype(np.int32)
    for o, v, g, h in parts[1:]:
        take = o & ~ok
        val[take], grad[take], hess[take] = v[take], g[take], h[take]
        ok |= o
        owners += o
    ok2, val2, grad2, hess2 = ov.sample(pts)
    assert owners.max() == 1, "a point was answered by two handles"
    assert np.array_equal(ok, ok2) and ok.sum() > 3000
    assert_same_f32(val[ok], val2[ok], "getFxn")
    assert_same_f32(grad[ok], grad2[ok], "gradient")
    assert_same_f32(hess[ok], hess2[ok], "Hessian")
    for s in slabs:
        s.close()


def test_zslab_volume_world1_end_to_end(gpu, tmp_path):
    ov, sc = truth()
    vol = ZSlabVolume(configure, RES)
    for i in range(NF):
        tr = synth.turntable_pose(i, 8, sc.size)
        vol.integrateCloud(sc.depth(tr), sc.bgra(i), tr)
    d, w, rgb = vol.download_local()
    assert_same_f32(d, ov.d, "d")
    assert np.array_equal(w, ov.w) and np.array_equal(rgb, ov.rgb)
    mesh = vol.reconstruct(w_min=1.0, color_by_confidence=True)
    v2, c2, cells2 = ov.march(1.0, 2)
    assert np.array_equal(mesh["cells"], cells2) and np.array_equal(mesh["rgb"], c2)
    assert_same_f32(mesh["vertices"], v2, "mesh")
    tv, tc, tk = vol.reconstruct_tensors(w_min=1.0, color_by_confidence=True)  # mesh kept in HBM
    assert tv.is_cuda and np.array_equal(tk.cpu().numpy().astype(np.uint64), cells2)
    assert_same_f32(tv.cpu().numpy().reshape(-1, 3), v2, "device mesh")
    assert np.array_equal(tc.cpu().numpy().reshape(-1, 3), c2)
    dv, dc, dk, first = vol.reconstruct_distributed(w_min=1.0, color_by_confidence=True)
    assert first == 0 and np.array_equal(dk.cpu().numpy().astype(np.uint64), cells2)
    assert_same_f32(dv.cpu().numpy().reshape(-1, 3), v2, "distributed mesh (world 1)")
    ply = str(tmp_path / "mesh.ply")
    assert vol.save_ply(ply, w_min=1.0, color_by_confidence=True) == len(cells2)
    body = open(ply, "rb").read().split(b"end_header\n", 1)[1]
    vrec = np.frombuffer(body[:3 * len(cells2) * 15], np.uint8).reshape(-1, 15)
    assert np.array_equal(vrec[:, :12].copy().view(np.float32), v2) and np.array_equal(vrec[:, 12:], c2)
    pts = np.random.RandomState(1).uniform(-0.06, 0.06, (300, 3)).astype(np.float32)
    ok, val, _, _ = vol.sample(pts)
    ok2, val2, _, _ = ov.sample(pts)
    assert np.array_equal(ok, ok2) and np.array_equal(val[ok], val2[ok])
    tr = synth.turntable_pose(1, 8, sc.size)
    assert np.isfinite(vol.renderView(tr)[..., 0]).sum() > 50
    # checkpoint through the block callbacks (tsdf_hip_save_blocks / tsdf_hip_load_blocks) == the one-handle file
    from cpu_tsdf_amd import capi
    from cpu_tsdf_amd.volume import TSDFVolumeOctree
    capi.set_tuning("vol_chunk", 16)
    try:
        path = str(tmp_path / "slabs.vol")
        vol.save(path)
        back = ZSlabVolume.load(path)
        assert all(np.array_equal(a, b) for a, b in zip(back.download_local(), (ov.d, ov.w, ov.rgb)))
        single = TSDFVolumeOctree()
        single.load(path)
        one = str(tmp_path / "one.vol")
        single.save(one)
        assert open(one, "rb").read() == open(path, "rb").read()
        single.close()
        back.close()
    finally:
        capi.set_tuning("vol_chunk", 256)
    vol.close()


def test_zslab_volume_frame_pairing_on_a_hip_slab(gpu):
    """ZSlabVolume.setFramePairing on a real HIP slab (world 1: HipSlab.pair_buffers + integrate_pair ->
    tsdf_hip_integrate_device2): NF frames, the odd one flushed by download_local, equal the oracle voxel for voxel."""
    ov, sc = truth()
    vol = ZSlabVolume(configure, RES)
    vol.setFramePairing(True)
    for i in range(NF):
        tr = synth.turntable_pose(i, 8, sc.size)
        vol.integrateCloud(sc.depth(tr), sc.bgra(i), tr)
    assert (vol._held is not None) == bool(NF % 2)
    d, w, rgb = vol.download_local()
    assert vol._held is None
    assert_same_f32(d, ov.d, "d")
    assert np.array_equal(w, ov.w) and np.array_equal(rgb, ov.rgb)
    vol.close()
