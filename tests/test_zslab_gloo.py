"""CPU tier: the Z-slab multi-GPU layer (cpu_tsdf_amd/zslab.py) with world_size 2 and 3 over gloo.  The slab
backend is the CPU oracle (tests/fake_slab.py), so this exercises exactly the host logic the GPU ranks run:
slab split, frame broadcast from the ingest rank, one-plane halo exchange, per-rank meshing, Morton merge,
sample routing, renderView by ray hand-off (halo refresh, advance rounds, integer all-reduce merge) -- and checks the result against a single unpartitioned volume."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpu_tsdf_amd import synth
from cpu_tsdf_amd.zslab import ZSlabVolume, morton_x_major, slab_range

RES, W, H, NF = 64, 80, 60, 4


def views(size):
    return [synth.turntable_pose(1, 8, size), synth.turntable_pose(3, 16, size, tilt=0.5),
            synth.look_at_pose((0.05, 0.02, -0.3)), synth.look_at_pose((0.0, 0.0, -0.05), target=(0.01, 0.0, 0.2)),
            synth.look_at_pose((0.02, -0.2, 0.01), target=(0.0, 0.0, 0.0))]


def configure(v):
    sc = synth.scene_a(RES, W, H)
    v.setResolution(RES, RES, RES)
    v.setGridSize(sc.size, sc.size, sc.size)
    v.setImageSize(W, H)
    v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    v.setSensorDistanceBounds(0.0, 3 * sc.size)
    v.setIntegrateColor(True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.fake_slab import OracleSlab
    vol = ZSlabVolume(configure, RES, slab_factory=OracleSlab)
    sc = synth.scene_a(RES, W, H)
    src = world - 1  # ingest on the LAST rank to prove src is honoured
    for i in range(NF):
        tr = synth.turntable_pose(i, 8, sc.size)
        if rank == src:
            vol.integrateCloud(sc.depth(tr), sc.bgra(i), tr, src=src)
        else:
            vol.integrateCloud(None, None, tr, src=src)
    mesh = vol.reconstruct(w_min=1.0, color_by_rgb=True)
    merged = vol.reconstruct_tensors(w_min=1.0, color_by_rgb=True, dst=world - 1)  # device-style merge, other root
    if rank == world - 1:
        tv, tc, tk = merged
        dist.send(torch.tensor([tv.shape[0]]), 0)
        dist.send(tv.contiguous(), 0)
        dist.send(tc.contiguous(), 0)
        dist.send(tk.contiguous(), 0)
    if rank == 0:
        n = torch.zeros(1, dtype=torch.int64)
        dist.recv(n, world - 1)
        n = int(n)
        tv, tc, tk = torch.empty((n, 9)), torch.empty((n, 9), dtype=torch.uint8), torch.empty((n,), dtype=torch.int64)
        dist.recv(tv, world - 1)
        dist.recv(tc, world - 1)
        dist.recv(tk, world - 1)
    # distributed sample sort: every rank keeps one contiguous range of the GLOBAL triangle order
    dv, dc, dk, first = vol.reconstruct_distributed(w_min=1.0, color_by_rgb=True, samples=64)
    slices = [None] * world if rank == 0 else None
    dist.gather_object((first, dv.numpy(), dc.numpy(), dk.numpy()), slices, dst=0)
    n_ply = vol.save_ply(out_path + ".ply", w_min=1.0, color_by_rgb=True)   # every rank writes its own byte ranges
    assert n_ply == sum(s[3].shape[0] for s in slices) if rank == 0 else n_ply > 0
    pts = np.random.RandomState(1).uniform(-0.06, 0.06, (300, 3)).astype(np.float32)
    samp = vol.sample(pts)
    renders, rounds = [], []
    p2p_records = []
    for k, tr in enumerate(views(sc.size)):
        renders.append(vol.renderView(tr, 1 + (k == 1), exchange="allreduce"))
        rounds.append(vol.last_render_rounds)
        again = vol.renderView(tr, 1 + (k == 1), exchange="p2p")  # records travel only to the next owner
        assert np.array_equal(again.view(np.uint32), renders[-1].view(np.uint32)), f"p2p hand-off differs, view {k}"
        p2p_records.append(vol.last_p2p_records)
        if k == 0:  # the default picks one of the two by world size; either way the same image
            assert np.array_equal(vol.renderView(tr, 1).view(np.uint32), renders[-1].view(np.uint32))
    assert max(p2p_records) < 0.8 * renders[0].shape[0] * renders[0].shape[1] * world
    zb, ze = vol.z_begin, vol.z_end
    # checkpoint: one .vol of the whole grid written on the LAST rank from blocks that straddle the slabs (16^3
    # blocks, slabs of 21-22 planes), read back on rank 0 into a new set of slabs built from the file's header
    from cpu_tsdf_amd import capi
    capi.set_tuning("vol_chunk", 16)
    vol_path = out_path + ".vol"
    vol.global_transform = synth.turntable_pose(1, 8, sc.size)
    vol.save(vol_path, dst=world - 1)
    dist.barrier()
    back = ZSlabVolume.load(vol_path, slab_factory=OracleSlab, src=0)
    assert (back.z_begin, back.z_end) == (zb, ze) and back.slab.color and not back._is_empty
    assert np.allclose(back.global_transform, vol.global_transform, rtol=1e-15, atol=1e-15)  # (the format keeps 16 digits)
    for name in ("d", "w", "rgb"):
        a, b = getattr(back.slab.ov, name)[zb:ze], getattr(vol.slab.ov, name)[zb:ze]
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"{name} after load on rank {rank}"
    with pytest.raises(Exception):   # every rank learns that the root could not read the file
        ZSlabVolume.load(vol_path + ".absent", slab_factory=OracleSlab, src=0)
    d, w = vol.slab.ov.d[zb:ze].copy(), vol.slab.ov.w[zb:ze].copy()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((zb, ze, d, w), gathered, dst=0)
    if rank == 0:
        np.savez(out_path, verts=mesh["vertices"], rgb=mesh["rgb"], cells=mesh["cells"], ok=samp[0], val=samp[1],
                 grad=samp[2], d=np.concatenate([g[2] for g in gathered]), w=np.concatenate([g[3] for g in gathered]),
                 bounds=np.array([[g[0], g[1]] for g in gathered]), rounds=np.array(rounds),
                 tverts=tv.numpy(), trgb=tc.numpy(), tcells=tk.numpy(),
                 dfirst=np.array([s[0] for s in slices]), dcount=np.array([len(s[3]) for s in slices]),
                 dverts=np.concatenate([s[1] for s in slices]), drgb=np.concatenate([s[2] for s in slices]),
                 dcells=np.concatenate([s[3] for s in slices]),
                 **{f"view{k}": r for k, r in enumerate(renders)})
    dist.barrier()
    dist.destroy_process_group()


def test_slab_range_and_morton_key():
    for nz, world in [(2048, 8), (10, 3), (7, 7), (64, 1)]:
        cuts = [slab_range(nz, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == nz
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1
    # x is the most significant bit of each octree level (src/lib/octree.cpp:119)
    cells = np.array([(1 << 42), (1 << 21), 1, (1 << 42) | (1 << 21) | 1, 2], dtype=np.uint64)
    assert morton_x_major(cells).tolist() == [4, 2, 1, 7, 8]


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_two_and_three_slabs_equal_one_volume(world, tmp_path):
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    # single-volume truth
    from oracle.oracle import OracleVolume
    from tests.fake_slab import _Cfg
    cfg = _Cfg()
    configure(cfg)
    ov = OracleVolume(cfg._p)
    sc = synth.scene_a(RES, W, H)
    for i in range(NF):
        tr = synth.turntable_pose(i, 8, sc.size)
        ov.integrate(sc.depth(tr), sc.bgra(i), synth.cam_from_vol_f32(tr))
    assert got["bounds"].tolist() == [list(slab_range(RES, world, r)) for r in range(world)]
    assert np.array_equal(got["d"], ov.d) and np.array_equal(got["w"], ov.w)
    verts, rgb, cells = ov.march(1.0, 1)
    assert len(cells) > 1000
    assert np.array_equal(got["cells"], cells), "merged triangle order"
    assert np.array_equal(got["verts"], verts) and np.array_equal(got["rgb"], rgb)
    # the tensor merge (send/recv of exactly-sized arrays + one sort by the Morton key) gives the same mesh
    assert np.array_equal(got["tcells"].astype(np.uint64), cells)
    assert np.array_equal(got["tverts"].reshape(-1, 3), verts) and np.array_equal(got["trgb"].reshape(-1, 3), rgb)
    # the distributed sort: rank slices concatenated in rank order are the same mesh; offsets add up; balanced
    assert np.array_equal(got["dcells"].astype(np.uint64), cells)
    assert np.array_equal(got["dverts"].reshape(-1, 3), verts) and np.array_equal(got["drgb"].reshape(-1, 3), rgb)
    assert got["dfirst"].tolist() == np.concatenate([[0], np.cumsum(got["dcount"])[:-1]]).tolist()
    assert got["dcount"].min() > 0.4 * len(cells) / world and got["dcount"].max() < 2.0 * len(cells) / world
    # the PLY the ranks wrote together: header, then 3n vertex records (xyz + rgb), then n faces (3, i, i+1, i+2)
    blob = open(out + ".ply", "rb").read()
    head, body = blob.split(b"end_header\n", 1)
    n = len(cells)
    assert f"element vertex {3 * n}\n".encode() in head and f"element face {n}\n".encode() in head and b"property uchar red" in head
    assert len(body) == 3 * n * 15 + n * 13
    vrec = np.frombuffer(body[:3 * n * 15], np.uint8).reshape(3 * n, 15)
    assert np.array_equal(vrec[:, :12].copy().view(np.float32), verts) and np.array_equal(vrec[:, 12:], rgb)
    frec = np.frombuffer(body[3 * n * 15:], np.uint8).reshape(n, 13)
    assert (frec[:, 0] == 3).all() and np.array_equal(frec[:, 1:].copy().view(np.int32).ravel(), np.arange(3 * n, dtype=np.int32))
    pts = np.random.RandomState(1).uniform(-0.06, 0.06, (300, 3)).astype(np.float32)
    ok, val, grad, _ = ov.sample(pts)
    assert np.array_equal(got["ok"], ok)
    assert np.array_equal(got["val"][ok], val[ok]) and np.array_equal(got["grad"][ok], grad[ok])
    # renderView by ray hand-off == the single-volume ray loop, bit for bit (NaN == NaN)
    from cpu_tsdf_amd.volume import eigen_affine_inverse, transform_cloud_with_normals
    hits = 0
    for k, tr in enumerate(views(sc.size)):
        want = transform_cloud_with_normals(ov.raycast(tr, 1 + (k == 1)), eigen_affine_inverse(tr))
        have = got[f"view{k}"]
        assert have.shape == want.shape
        assert np.array_equal(have.view(np.uint32), want.view(np.uint32)) or \
            np.array_equal(np.nan_to_num(have, nan=-7.0), np.nan_to_num(want, nan=-7.0)), f"view {k}"
        hits += int(np.isfinite(want[..., 0]).sum())
    assert hits > 2000
    assert got["rounds"].max() <= world + 2 and got["rounds"].max() >= 2   # + 1: a far-extrapolated hit finishes at its owner
    # the distributed checkpoint is byte for byte the file one writer produces from the whole grid
    from tests.common import write_vol_from_arrays
    one = str(tmp_path / "one.vol")
    write_vol_from_arrays(one, cfg._p, ov.d, ov.w, ov.rgb, global_transform=synth.turntable_pose(1, 8, sc.size))
    assert open(out + ".vol", "rb").read() == open(one, "rb").read()


def _thin_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.fake_slab import OracleSlab
    res, w, h = 32, 64, 48
    sc = synth.scene_a(res, w, h)

    def conf(v):
        v.setResolution(res, res, res)
        v.setGridSize(sc.size, sc.size, sc.size)
        v.setImageSize(w, h)
        v.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
        v.setSensorDistanceBounds(0.0, 3 * sc.size)
        v.setIntegrateColor(True)
    vol = ZSlabVolume(conf, res, slab_factory=OracleSlab)
    assert vol.halo > vol.z_end - vol.z_begin          # the halo spans more than one neighbour
    for i in range(3):
        tr = synth.turntable_pose(i, 3, sc.size)
        vol.integrateCloud(sc.depth(tr) if rank == 0 else None, sc.bgra(i) if rank == 0 else None, tr)
    views = [synth.turntable_pose(0, 8, sc.size), synth.look_at_pose((0.1, -0.05, 0.2))]
    imgs = [vol.renderView(tr, 1, exchange=ex) for tr in views for ex in ("allreduce", "p2p")]
    mesh = vol.reconstruct(w_min=1.0, color_by_rgb=True)
    # checkpoint with 16^3 blocks, each crossing three of the thin slabs; and getFxn routed to the owners
    from cpu_tsdf_amd import capi
    capi.set_tuning("vol_chunk", 16)
    vol.save(out_path + ".vol", dst=2)
    dist.barrier()
    back = ZSlabVolume.load(out_path + ".vol", slab_factory=OracleSlab, src=1)
    zb, ze = vol.z_begin, vol.z_end
    for name in ("d", "w", "rgb"):
        assert np.array_equal(getattr(back.slab.ov, name)[zb:ze].view(np.uint8), getattr(vol.slab.ov, name)[zb:ze].view(np.uint8))
    pts = np.random.RandomState(3).uniform(-0.05, 0.05, (200, 3)).astype(np.float32)
    samp = vol.sample(pts)
    if rank == 0:
        np.savez(out_path, cells=mesh["cells"], verts=mesh["vertices"], ok=samp[0], val=samp[1],
                 **{f"img{k}": im for k, im in enumerate(imgs)})
    dist.barrier()
    dist.destroy_process_group()


def test_slabs_thinner_than_the_render_halo(tmp_path):
    """Five slabs of 6-7 planes under a 12-plane render halo: every rank's halo is filled by SEVERAL owners, rays cross
    up to five slabs and hit points extrapolate into far slabs; images and mesh still equal one volume's."""
    from cpu_tsdf_amd.volume import eigen_affine_inverse, transform_cloud_with_normals
    from oracle.oracle import OracleVolume
    from tests.fake_slab import _Cfg
    out = str(tmp_path / "thin.npz")
    mp.spawn(_thin_worker, args=(5, _free_port(), out), nprocs=5, join=True)
    got = np.load(out)
    res, w, h = 32, 64, 48
    sc = synth.scene_a(res, w, h)
    cfg = _Cfg()
    cfg.setResolution(res, res, res)
    cfg.setGridSize(sc.size, sc.size, sc.size)
    cfg.setImageSize(w, h)
    cfg.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    cfg.setSensorDistanceBounds(0.0, 3 * sc.size)
    cfg.setIntegrateColor(True)
    ov = OracleVolume(cfg._p)
    for i in range(3):
        tr = synth.turntable_pose(i, 3, sc.size)
        ov.integrate(sc.depth(tr), sc.bgra(i), synth.cam_from_vol_f32(tr))
    views = [synth.turntable_pose(0, 8, sc.size), synth.look_at_pose((0.1, -0.05, 0.2))]
    k = 0
    for tr in views:
        want = transform_cloud_with_normals(ov.raycast(tr, 1), eigen_affine_inverse(tr))
        for _ in range(2):
            have = got[f"img{k}"]
            k += 1
            assert np.array_equal(np.nan_to_num(have, nan=-7.0), np.nan_to_num(want, nan=-7.0))
    verts, rgb, cells = ov.march(1.0, 1)
    assert len(cells) > 300 and np.array_equal(got["cells"], cells) and np.array_equal(got["verts"], verts)
    ok, val, _, _ = ov.sample(np.random.RandomState(3).uniform(-0.05, 0.05, (200, 3)).astype(np.float32))
    assert np.array_equal(got["ok"], ok) and ok.sum() > 20 and np.array_equal(got["val"][ok], val[ok])


def _load_worker(rank, world, port, vol_path, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_tsdf_amd import capi
    from tests.fake_slab import OracleSlab
    capi.set_tuning("vol_chunk", 16)
    back = ZSlabVolume.load(vol_path, slab_factory=OracleSlab, src=0)
    zb, ze = back.z_begin, back.z_end
    got = [None] * world if rank == 0 else None
    dist.gather_object((zb, ze, back.slab.p.layout, back.slab.ov.d[zb:ze].copy(), back.slab.ov.w[zb:ze].copy()), got, dst=0)
    if rank == 0:
        np.savez(out_path, d=np.concatenate([g[3] for g in got]), w=np.concatenate([g[4] for g in got]),
                 layouts=np.array([g[2] for g in got]))
    dist.barrier()
    dist.destroy_process_group()


def test_load_of_fractional_weights_falls_back_to_float_weights_on_every_rank(tmp_path):
    """a .vol whose weights are not observation counts (another weighting wrote it) makes a PACKED slab
    refuse the upload -- on a NON-root rank here (only the upper half of the grid holds such weights).  That rank
    used to raise inside the block protocol and leave the root blocked in its next send; now every rank finishes
    the protocol, the outcome is agreed by an all-reduce, and the whole load is repeated with float weights."""
    from cpu_tsdf_amd import capi
    from tests.common import write_vol_from_arrays
    from tests.fake_slab import _Cfg
    res = 32
    cfg = _Cfg()
    sc = synth.scene_a(res, W, H)
    cfg.setResolution(res, res, res)
    cfg.setGridSize(sc.size, sc.size, sc.size)
    cfg.setImageSize(W, H)
    cfg.setCameraIntrinsics(sc.fx, sc.fy, sc.cx, sc.cy)
    cfg.setIntegrateColor(False)
    rng = np.random.RandomState(4)
    d = rng.uniform(-1, 1, (res, res, res)).astype(np.float32)
    w = rng.randint(0, 5, (res, res, res)).astype(np.float32)
    w[res // 2 + 3:] += np.float32(0.25)  # only rank 1's planes
    path = str(tmp_path / "frac.vol")
    write_vol_from_arrays(path, cfg._p, d, w, None, chunk=16)
    out = str(tmp_path / "out.npz")
    port = _free_port()
    mp.spawn(_load_worker, args=(2, port, path, out), nprocs=2, join=True)
    got = np.load(out)
    assert (got["layouts"] == capi.LAYOUT_F32W).all()
    assert np.array_equal(got["d"], d) and np.array_equal(got["w"], w)


def _pair_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.fake_slab import OracleSlab
    sc = synth.scene_a(RES, W, H)
    src = world - 1
    vol = ZSlabVolume(configure, RES, slab_factory=OracleSlab)
    vol.setFramePairing(True)
    n_sent = 0

    def frame(i):
        tr = synth.turntable_pose(i, 8, sc.size)
        vol.integrateCloud(sc.depth(tr) if rank == src else None, sc.bgra(i) if rank == src else None, tr, src=src)
    frame(0)
    assert vol._held is not None            # parked: nothing integrated, nothing sent yet
    assert not vol.slab.ov.w[vol.z_begin:vol.z_end].any()
    frame(1)                                # the partner: both travel and are integrated, in order
    assert vol._held is None and vol.slab.ov.w[vol.z_begin:vol.z_end].max() == 2.0
    frame(2)                                # parked again ...
    pts = np.random.RandomState(2).uniform(-0.06, 0.06, (50, 3)).astype(np.float32)
    samp = vol.sample(pts)                  # ... and flushed by the next collective call of another kind
    assert vol._held is None and vol.slab.ov.w[vol.z_begin:vol.z_end].max() == 3.0
    frame(3)
    vol.setFramePairing(False)              # switching it off integrates what was waiting
    assert vol._held is None and vol.slab.ov.w[vol.z_begin:vol.z_end].max() == 4.0
    frame(4)                                # unpaired again: at once
    assert vol.slab.ov.w[vol.z_begin:vol.z_end].max() == 5.0
    zb, ze = vol.z_begin, vol.z_end
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((zb, ze, vol.slab.ov.d[zb:ze].copy(), vol.slab.ov.w[zb:ze].copy(), vol.slab.ov.rgb[zb:ze].copy()), gathered, dst=0)
    if rank == 0:
        np.savez(out_path, d=np.concatenate([g[2] for g in gathered]), w=np.concatenate([g[3] for g in gathered]),
                 rgb=np.concatenate([g[4] for g in gathered]), ok=samp[0], val=samp[1])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_frame_pairing_across_ranks_equals_frame_by_frame(world, tmp_path):
    """ZSlabVolume.setFramePairing: every other frame waits on the ingest rank and travels with its
    partner in one exchange; a parked frame is integrated on its own by the next collective call of another kind and by
    switching pairing off.  Five frames that way -- pair, parked + getFxn, parked + switch-off, single -- equal five
    integrateCloud calls on one unpartitioned volume, voxel for voxel, and getFxn saw exactly three frames."""
    out = str(tmp_path / "pair.npz")
    mp.spawn(_pair_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    from oracle.oracle import OracleVolume
    from tests.fake_slab import _Cfg
    cfg = _Cfg()
    configure(cfg)
    ov = OracleVolume(cfg._p)
    sc = synth.scene_a(RES, W, H)
    pts = np.random.RandomState(2).uniform(-0.06, 0.06, (50, 3)).astype(np.float32)
    for i in range(5):
        tr = synth.turntable_pose(i, 8, sc.size)
        ov.integrate(sc.depth(tr), sc.bgra(i), synth.cam_from_vol_f32(tr))
        if i == 2:
            ok, val, _, _ = ov.sample(pts)
    assert np.array_equal(got["d"], ov.d) and np.array_equal(got["w"], ov.w) and np.array_equal(got["rgb"], ov.rgb)
    assert np.array_equal(got["ok"], ok) and np.array_equal(got["val"][ok], val[ok])
