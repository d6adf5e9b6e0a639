"""Z-slab partition of one TSDF grid over the GPUs of a node: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The reference has no distributed code at all; this layer is new.  Voxels are
independent during integration, so the grid is cut into contiguous Z-slabs and the only traffic is

* ``integrateCloud``: one broadcast of the depth (+ colour) frame from the rank that ingested it
  (1.2 MB + 1.2 MB at 640x480) -- no voxel ever crosses a link;
* ``reconstruct`` (marching cubes): a cell reads planes z and z+1, so every rank receives ONE plane
  (the first plane of the next slab: res^2 * 8..12 B) from its +z neighbour by point-to-point send/recv,
  meshes its own cells, and the per-rank triangle lists are merged by the reference's Morton key -- on one rank
  (``reconstruct``, ``reconstruct_tensors``) or by a distributed sample sort that leaves every rank one contiguous
  range of the global triangle order (``reconstruct_distributed``);
* ``sample`` (getFxn...): a point is answered by the rank that owns the lower-corner plane.

* ``renderView``: a ray's step sequence depends on the last voxel it visited (tsdf_volume_octree.cpp:360),
  so slabs cannot march a ray independently and take the nearest hit.  Instead the ray's loop state
  travels: every rank resumes the rays whose next voxel it owns until they finish or reach another slab
  (``tsdf_hip_raycast_advance``), the per-rank deltas are merged by ONE integer SUM all-reduce per round
  (each ray is advanced by exactly one rank), and this repeats until no ray is suspended -- at most
  world + 2 rounds: a ray crosses the slabs monotonically in z, and a hit whose extrapolated point lies in
  another slab travels once more to have its normal computed there.  The refinement walk of a hit and
  its trilinear samples look back/around by up to ``render_halo`` planes, which every rank refreshes from
  both neighbours before rendering.  `exchange="p2p"` keeps only a compact record list per rank and moves each
  suspended record point-to-point to its next owner (traffic ~ rays crossing a boundary).  Results are
  bit-identical to the single-GPU kernel in both forms.

* ``save`` / ``ZSlabVolume.load``: the .vol checkpoint of the whole grid.  The root rank runs the streaming
  writer / reader of the C ABI (``tsdf_hip_save_blocks`` / ``tsdf_hip_load_blocks``); each 256^3 block it
  asks for is served by the one or two ranks whose slabs it crosses, point to point.

The slab backend is injected (``slab_factory``) so the N > 1 logic is testable on CPU with gloo; the
default backend is the HIP volume.  There is no CPU fallback in the product: the default factory raises
without a GPU.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .volume import MarchingCubesTSDFOctree, TSDFVolumeOctree


def slab_range(nz, world, rank):
    """Contiguous, balanced split of nz planes: the first nz % world ranks get one extra plane."""
    base, extra = divmod(nz, world)
    z0 = rank * base + min(rank, extra)
    return z0, z0 + base + (1 if rank < extra else 0)


def morton_x_major(cells):
    """Sort key of the reference's triangle order (octree pre-order, child index 4*(x>cx)+2*(y>cy)+(z>cz),
    src/lib/octree.cpp:119,257-264) from packed cells (x<<42 | y<<21 | z)."""
    c = np.asarray(cells, dtype=np.uint64)
    x, y, z = (c >> np.uint64(42)) & np.uint64(0x1FFFFF), (c >> np.uint64(21)) & np.uint64(0x1FFFFF), c & np.uint64(0x1FFFFF)

    def spread(v):
        v = v & np.uint64(0x1FFFFF)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v

    return (spread(x) << np.uint64(2)) | (spread(y) << np.uint64(1)) | spread(z)


def morton_x_major_torch(cells):
    """morton_x_major on a torch int64 tensor (any device).  21-bit coordinates interleave into 63 bits, so the
    signed 64-bit order equals the unsigned one."""
    m = 0x1FFFFF
    x, y, z = (cells >> 42) & m, (cells >> 21) & m, cells & m

    def spread(v):
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v

    return (spread(x) << 2) | (spread(y) << 1) | spread(z)


class HipSlab:
    """Slab backend on one MI355X: a TSDFVolumeOctree restricted to [z_begin, z_end) (+ `halo` planes on each
    side: 1 for marching cubes / sampling, tsdf_hip_render_halo() for renderView), exchanging planes as
    torch CUDA tensors."""

    def __init__(self, configure, z_begin, z_end, nz, device_index, halo=1):
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.vol = TSDFVolumeOctree()
        configure(self.vol)
        self.res = self.vol.getResolution()
        self.color = bool(self.vol._p.integrate_color)
        self.z_begin, self.z_end = z_begin, z_end
        self.vol.setZSlab(z_begin, z_end, halo=halo if (z_end < nz or z_begin > 0) else 0, device=device_index)
        self.vol.setStream(torch.cuda.current_stream(self.device).cuda_stream)
        self.vol.reset()

    def frame_buffers(self):
        """Depth and colour images of the incoming frame as views of ONE allocation ([depth | bgra]): the
        integrate kernel then needs no staging copy and the frame travels in one broadcast."""
        W, H = self.vol.getImageSize()
        self.frame_packed = torch.empty((2 if self.color else 1, H, W), dtype=torch.float32, device=self.device)
        depth = self.frame_packed[0]
        bgra = self.frame_packed[1].view(torch.uint8).view(H, W, 4) if self.color else None
        return depth, bgra

    def integrate_tensor(self, depth, bgra, trans):
        self.vol.integrateCloudDevice(depth.data_ptr(), bgra.data_ptr() if bgra is not None else 0, trans)

    def pair_buffers(self):
        """Two frames, each [depth | bgra], in ONE allocation: ZSlabVolume's frame pairing parks frame A in the first half,
        frame B in the second, and sends both in one broadcast."""
        W, H = self.vol.getImageSize()
        buf = torch.empty((2, 2 if self.color else 1, H, W), dtype=torch.float32, device=self.device)
        views = [(buf[i, 0], buf[i, 1].view(torch.uint8).view(H, W, 4) if self.color else None) for i in range(2)]
        return buf, views

    def integrate_pair(self, views, trans_a, trans_b):
        """Both frames of a pair in one call: tsdf_hip_integrate_device2 -- ONE sweep of this slab where both poses see all of
        it (k_integrate2), two launches in order otherwise; the same voxels either way."""
        (da, ca), (db, cb) = views
        self.vol.integrateCloudDevice2((da.data_ptr(), ca.data_ptr() if ca is not None else 0, trans_a),
                                       (db.data_ptr(), cb.data_ptr() if cb is not None else 0, trans_b))

    def get_planes(self, z0, nz):
        nx, ny, _ = self.res
        d = torch.empty((nz, ny, nx), dtype=torch.float32, device=self.device)
        w = torch.empty_like(d)
        rgb = torch.empty((nz, ny, nx), dtype=torch.int32, device=self.device) if self.color else None
        capi.check(capi.load().tsdf_hip_get_planes_device(self.vol._need(), z0, nz, C.c_void_p(d.data_ptr()),
                                                          C.c_void_p(w.data_ptr()),
                                                          C.c_void_p(rgb.data_ptr()) if rgb is not None else None),
                   "get_planes_device")
        return d, w, rgb

    def plane_buffers(self, nz):
        nx, ny, _ = self.res
        d = torch.empty((nz, ny, nx), dtype=torch.float32, device=self.device)
        return d, torch.empty_like(d), (torch.empty((nz, ny, nx), dtype=torch.int32, device=self.device) if self.color else None)

    def set_planes(self, z0, d, w, rgb):
        capi.check(capi.load().tsdf_hip_set_planes_device(self.vol._need(), z0, d.shape[0], C.c_void_p(d.data_ptr()),
                                                          C.c_void_p(w.data_ptr()),
                                                          C.c_void_p(rgb.data_ptr()) if rgb is not None else None),
                   "set_planes_device")

    def params(self):
        """tsdf_params of the WHOLE grid (slab fields cleared)."""
        p = capi.TsdfParams.from_buffer_copy(self.vol._p)
        p.z_begin = p.z_end = p.halo = 0
        return p

    def get_block(self, x0, y0, z0, nx, ny, nz):
        """Host copy of a box of voxels this slab holds: d, w float32 [nz, ny, nx], rgb uint8 [..., 3] or None."""
        return self.vol.download(x0, y0, z0, nx, ny, nz)

    def set_block(self, x0, y0, z0, d, w, rgb):
        self.vol.upload(d, w, rgb, x0, y0, z0)

    def march(self, w_min, by_rgb, by_confidence):
        mc = MarchingCubesTSDFOctree()
        mc.setInputTSDF(self.vol)
        mc.setMinWeight(w_min)
        mc.setColorByRGB(by_rgb)
        mc.setColorByConfidence(by_confidence)
        return mc.reconstruct(want_cells=True)

    def march_tensors(self, w_min, by_rgb, by_confidence):
        """Marching cubes of the slab, mesh left on the device: (verts (n,9) float32, rgb (n,9) uint8 or None,
        cells (n,) int64 packed x<<42|y<<21|z) as torch CUDA tensors."""
        lib = capi.load()
        n = C.c_uint64(0)
        mode = 2 if by_confidence else (1 if by_rgb else 0)
        capi.check(lib.tsdf_hip_march(self.vol._need(), C.c_float(w_min), mode, C.byref(n)), "march")
        n = int(n.value)
        verts = torch.empty((n, 9), dtype=torch.float32, device=self.device)
        rgb = torch.empty((n, 9), dtype=torch.uint8, device=self.device) if mode else None
        cells = torch.empty((n,), dtype=torch.int64, device=self.device)
        if n:
            capi.check(lib.tsdf_hip_march_fetch_device(self.vol._need(), C.c_void_p(verts.data_ptr()),
                                                       C.c_void_p(rgb.data_ptr()) if rgb is not None else None,
                                                       C.c_void_p(cells.data_ptr())), "march_fetch_device")
        return verts, rgb, cells

    def sample(self, pts):
        return self.vol.sample(pts)

    def render(self, trans, ds):
        return self.vol.renderView(trans, ds)

    @staticmethod
    def _rot_org(trans):
        trans = np.asarray(trans, dtype=np.float64)
        return (np.ascontiguousarray(trans[:3, :3].astype(np.float32).reshape(9)),
                np.ascontiguousarray(trans[:3, 3].astype(np.float32)))

    def ray_begin(self, trans, ds):
        W, H = self.vol.getImageSize()
        n = (H // ds) * (W // ds)
        state = torch.empty((n, capi.RAY_RECORD_INTS), dtype=torch.int32, device=self.device)
        rot, org = self._rot_org(trans)
        capi.check(capi.load().tsdf_hip_raycast_begin(self.vol._need(), capi.as_f32p(rot), capi.as_f32p(org), ds,
                                                      C.c_void_p(state.data_ptr())), "raycast_begin")
        return state

    def ray_advance(self, trans, ds, state, rank, world):
        delta = torch.empty_like(state)
        rot, org = self._rot_org(trans)
        capi.check(capi.load().tsdf_hip_raycast_advance(self.vol._need(), capi.as_f32p(rot), capi.as_f32p(org), ds,
                                                        rank, world, C.c_void_p(state.data_ptr()),
                                                        C.c_void_p(delta.data_ptr())), "raycast_advance")
        return delta

    def ray_advance_list(self, trans, ds, records, rank, world):
        """Advance a compact (k, 24) int32 list of ray records in place (tsdf_hip_raycast_advance_list)."""
        records = records.contiguous()
        rot, org = self._rot_org(trans)
        capi.check(capi.load().tsdf_hip_raycast_advance_list(self.vol._need(), capi.as_f32p(rot), capi.as_f32p(org), ds,
                                                             rank, world, C.c_void_p(records.data_ptr()), records.shape[0]),
                   "raycast_advance_list")
        return records

    def image_size(self):
        return self.vol.getImageSize()

    def synchronize(self):
        self.vol.synchronize()

    def close(self):
        self.vol.close()


def out_sum(out, group):
    """Every ray finished on exactly one rank (zeros elsewhere): an integer SUM all-reduce assembles the image."""
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out


def _host_staged(group):
    """gloo's POINT-TO-POINT operations hand the tensor's raw pointer to the transport (ProcessGroupGloo::send / recv build
    an unbound buffer on `data_ptr()`): there is no device support behind them and no stream ordering -- on this platform the
    host CAN dereference a device pointer, so a device tensor "works", racing with the kernels that write or read it (round
    5: three HIP ranks over gloo faulted one run in two in the ray hand-off, tests/test_zslab_hip_ranks_gpu.py).  With gloo,
    device tensors therefore travel through a host copy; RCCL ("nccl"), the backend of a real node, takes them as they are.
    (gloo's collectives -- broadcast, all_reduce, all_gather -- do stage device tensors themselves.)"""
    return dist.get_backend(group) == "gloo"


def p2p_batch(ops, group):
    """ops = [("send" | "recv", tensor, peer), ...]: one batch_isend_irecv, waited for."""
    if not ops:
        return
    stage = _host_staged(group)
    batch, landed = [], []
    for kind, t, peer in ops:
        buf = t
        if stage and t.is_cuda:
            if kind == "send":
                buf = t.detach().cpu()  # a synchronous copy, ordered behind the kernels that produced t
            else:
                buf = torch.empty(t.shape, dtype=t.dtype, device="cpu")
                landed.append((t, buf))
        batch.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, buf, peer, group))
    for req in dist.batch_isend_irecv(batch):
        req.wait()
    for t, buf in landed:
        t.copy_(buf)


def send_tensor(t, dst, group):
    dist.send(t.detach().cpu() if _host_staged(group) and t.is_cuda else t, dst, group=group)


def recv_tensor(t, src, group):
    if _host_staged(group) and t.is_cuda:
        buf = torch.empty(t.shape, dtype=t.dtype, device="cpu")
        dist.recv(buf, src, group=group)
        t.copy_(buf)
    else:
        dist.recv(t, src, group=group)


def _default_factory(configure, z_begin, z_end, nz, rank, halo=1):
    if capi.load().tsdf_hip_device_count() <= 0:
        raise RuntimeError("ZSlabVolume needs a HIP device per rank (there is no CPU fallback)")
    return HipSlab(configure, z_begin, z_end, nz, int(torch.cuda.current_device()), halo=halo)


def render_halo(configure):
    """Planes of halo per side that renderView across slabs needs for this configuration."""
    probe = TSDFVolumeOctree()  # parameters only: no handle (and no GPU memory) until reset()
    configure(probe)
    return int(capi.load().tsdf_hip_render_halo(C.byref(probe._p)))


class ZSlabVolume:
    """One logical TSDF volume, Z-slab partitioned over the ranks of `group`.

    configure(vol) applies the usual setters (setResolution, setGridSize, ...) to a volume object; it is
    called once per rank.  All methods are collective: every rank calls them in the same order."""

    def __init__(self, configure, resolution_z, group=None, slab_factory=None, halo=None):
        """halo: planes kept on each side of the slab; None = enough for renderView across slabs
        (tsdf_hip_render_halo: ~12 planes at the default 3 cm truncation and 2^-8 m voxels); 1 is enough for
        integrateCloud + reconstruct + sample only."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.nz = int(resolution_z)
        self.z_begin, self.z_end = slab_range(self.nz, self.world, self.rank)
        factory = slab_factory or _default_factory
        self.halo = 0 if self.world == 1 else (render_halo(configure) if halo is None else int(halo))
        self.slab = factory(configure, self.z_begin, self.z_end, self.nz, self.rank, halo=self.halo)
        self._frame = self.slab.frame_buffers()
        self.global_transform = np.eye(4)
        self._is_empty = True
        self.max_cell_size = (0.5, 0.5, 0.5)  # the reference's default (tsdf_volume_octree.cpp:72-74); only save() uses it
        self._store_failed, self._store_error = 0, ""
        self._pairing, self._held, self._pair = False, None, None

    # -- frame pairing (not in the reference: two integrateCloud calls, include/cpu_tsdf/impl/tsdf_volume_octree.hpp:48-103) ----
    def setFramePairing(self, flag):
        """Collective.  With pairing on, integrateCloud parks every other frame on the ingest rank and sends it TOGETHER with
        the next one -- one broadcast of [A | B] instead of two -- and every rank integrates the pair in one sweep of its slab
        where both poses see all of it (tsdf_hip_integrate_device2 -> k_integrate2; two launches otherwise).  The voxels are
        those of the two calls in order, bit for bit.  A parked frame is integrated on its own by the next call of any other
        method (every one of them is collective, so the ranks agree), or by switching pairing off."""
        if not flag:
            self._flush_pair()
        self._pairing = bool(flag)

    def _pair_buffers(self):
        if self._pair is None:
            if hasattr(self.slab, "pair_buffers"):
                self._pair = self.slab.pair_buffers()
            else:  # a backend without one allocation for two frames (tests/fake_slab.py): two plain frame buffers
                a, b = self.slab.frame_buffers(), self.slab.frame_buffers()
                self._pair = (None, [a, b])
        return self._pair

    def _integrate_pair(self, views, ta, tb):
        if hasattr(self.slab, "integrate_pair"):
            self.slab.integrate_pair(views, ta, tb)
        else:
            self.slab.integrate_tensor(views[0][0], views[0][1], ta)
            self.slab.integrate_tensor(views[1][0], views[1][1], tb)

    def _flush_pair(self):
        """A frame parked for pairing is sent and integrated on its own (collective: every rank holds the same record)."""
        if self._held is None:
            return
        ta, sa = self._held
        self._held = None
        _, views = self._pair_buffers()
        fd, fc = views[0]
        if self.world > 1:
            dist.broadcast(fd, src=sa, group=self.group)
            if fc is not None:
                dist.broadcast(fc, src=sa, group=self.group)
        self.slab.integrate_tensor(fd, fc, ta)

    # -- integrateCloud -------------------------------------------------------------------------------
    def integrateCloud(self, depth, bgra, trans, src=0):
        """`depth`/`bgra` are only read on rank `src` (numpy arrays or tensors); everyone gets the frame by
        one broadcast each, then integrates its own slab."""
        if self._pairing:
            buf, views = self._pair_buffers()
            fd, fc = views[0 if self._held is None else 1]
            if self.rank == src:
                fd.copy_(torch.as_tensor(depth).reshape(fd.shape))
                if fc is not None:
                    fc.copy_(torch.as_tensor(bgra).reshape(fc.shape))
            self._is_empty = False
            if self._held is None:  # frame A waits on the ingest rank for its partner
                self._held = (np.array(trans, dtype=np.float64), src)
                return True
            ta, sa = self._held
            self._held = None
            if self.world > 1:
                if buf is not None and sa == src:
                    dist.broadcast(buf, src=src, group=self.group)  # both frames, depth + colour, in ONE collective
                else:
                    for (d_, c_), s_ in ((views[0], sa), (views[1], src)):
                        dist.broadcast(d_, src=s_, group=self.group)
                        if c_ is not None:
                            dist.broadcast(c_, src=s_, group=self.group)
            self._integrate_pair(views, ta, np.asarray(trans, dtype=np.float64))
            return True
        fd, fc = self._frame
        if self.rank == src:
            fd.copy_(torch.as_tensor(depth).reshape(fd.shape))
            if fc is not None:
                fc.copy_(torch.as_tensor(bgra).reshape(fc.shape))
        if self.world > 1:
            packed = getattr(self.slab, "frame_packed", None)
            if packed is not None:  # depth + colour in one collective
                dist.broadcast(packed, src=src, group=self.group)
            else:
                dist.broadcast(fd, src=src, group=self.group)
                if fc is not None:
                    dist.broadcast(fc, src=src, group=self.group)
        self.slab.integrate_tensor(fd, fc, np.asarray(trans, dtype=np.float64))
        self._is_empty = False
        return True

    # -- marching cubes ---------------------------------------------------------------------------------
    def exchange_halo(self, planes=1, both=False):
        """Every rank fills the `planes` halo planes above its slab (and, with `both`, below it) with the planes their
        owners hold -- point-to-point, one batch.  The owner is usually the adjacent rank; slabs thinner than
        `planes` make the halo span several ranks, each of which sends what it owns of it."""
        self._flush_pair()
        if self.world == 1:
            return
        planes = min(int(planes), self.halo)
        lo = max(0, self.z_begin - planes) if both else self.z_begin
        hi = min(self.nz, self.z_end + planes)
        want = [(lo, self.z_begin), (self.z_end, hi)]          # my halo ranges (below, above)
        ops, recvs, keep = [], [], []
        for r in range(self.world):
            if r == self.rank:
                continue
            zb, ze = slab_range(self.nz, self.world, r)
            # what rank r wants of MY planes (the same arithmetic it does for itself)
            rlo = max(0, zb - planes) if both else zb
            rhi = min(self.nz, ze + planes)
            for a, b in ((rlo, zb), (ze, rhi)):
                s0, s1 = max(a, self.z_begin), min(b, self.z_end)
                if s0 < s1:
                    send = self.slab.get_planes(s0, s1 - s0)
                    keep.append(send)
                    ops += [("send", t, r) for t in send if t is not None]
            # what I want of rank r's planes
            for a, b in want:
                r0, r1 = max(a, zb), min(b, ze)
                if r0 < r1:
                    buf = self.slab.plane_buffers(r1 - r0)
                    recvs.append((r0, buf))
                    ops += [("recv", t, r) for t in buf if t is not None]
        self.slab.synchronize()
        p2p_batch(ops, self.group)
        for z0, buf in recvs:
            self.slab.set_planes(z0, *buf)

    def reconstruct(self, w_min=2.5, color_by_rgb=False, color_by_confidence=False, dst=0, gather=True):
        """MarchingCubesTSDFOctree::reconstruct over all slabs.  Returns the merged mesh on rank `dst`
        (vertices (3n,3) float32, polygons, rgb, cells) in the reference's triangle order; None elsewhere.
        gather=False: every rank keeps the triangles of its own slab (already in Morton order within the
        slab) -- the scalable form for meshes too large to collect on one rank."""
        self.exchange_halo()
        part = self.slab.march(w_min, color_by_rgb, color_by_confidence)
        if not gather:
            return part
        if self.world == 1:
            parts = [part]
        else:
            parts = [None] * self.world if self.rank == dst else None
            dist.gather_object(part, parts, dst=dst, group=self.group)
            if self.rank != dst:
                return None
        cells = np.concatenate([p["cells"] for p in parts])
        verts = np.concatenate([p["vertices"].reshape(-1, 3, 3) for p in parts])
        order = np.argsort(morton_x_major(cells), kind="stable")
        has_rgb = parts[0]["rgb"] is not None
        rgb = np.concatenate([p["rgb"].reshape(-1, 3, 3) for p in parts])[order].reshape(-1, 3) if has_rgb else None
        n = len(cells)
        return {"vertices": verts[order].reshape(-1, 3), "polygons": np.arange(3 * n, dtype=np.int32).reshape(n, 3),
                "rgb": rgb, "cells": cells[order]}

    def reconstruct_tensors(self, w_min=2.5, color_by_rgb=False, color_by_confidence=False, dst=0):
        """reconstruct() with the per-slab meshes merged ON THE DEVICE: every rank keeps its triangles in HBM,
        the counts are all-gathered, the triangle arrays go to rank `dst` point-to-point (RCCL send/recv of
        exactly-sized tensors, no pickling, no host copy), and rank `dst` orders them by the reference's Morton
        key with one device sort.  Returns tensors (verts (n,9) float32, rgb (n,9) uint8 or None, cells (n,)
        int64) on rank `dst`, None elsewhere."""
        self.exchange_halo()
        verts, rgb, cells = self.slab.march_tensors(w_min, color_by_rgb, color_by_confidence)
        if self.world == 1:
            return verts, rgb, cells
        dev = verts.device
        counts = torch.zeros(self.world, dtype=torch.int64, device=dev)
        counts[self.rank] = verts.shape[0]
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)
        counts = [int(c) for c in counts.tolist()]
        has_rgb = rgb is not None
        ops = []
        if self.rank == dst:
            total = sum(counts)
            all_v = torch.empty((total, 9), dtype=torch.float32, device=dev)
            all_c = torch.empty((total, 9), dtype=torch.uint8, device=dev) if has_rgb else None
            all_k = torch.empty((total,), dtype=torch.int64, device=dev)
            off = 0
            for r, n in enumerate(counts):
                sl = slice(off, off + n)
                off += n
                if n == 0:
                    continue
                if r == dst:
                    all_v[sl], all_k[sl] = verts, cells
                    if has_rgb:
                        all_c[sl] = rgb
                else:
                    ops.append(("recv", all_v[sl], r))
                    ops.append(("recv", all_k[sl], r))
                    if has_rgb:
                        ops.append(("recv", all_c[sl], r))
        elif counts[self.rank]:
            ops.append(("send", verts, dst))
            ops.append(("send", cells, dst))
            if has_rgb:
                ops.append(("send", rgb, dst))
        p2p_batch(ops, self.group)
        if self.rank != dst:
            return None
        order = torch.argsort(morton_x_major_torch(all_k), stable=True)
        return all_v[order], (all_c[order] if has_rgb else None), all_k[order]

    def reconstruct_distributed(self, w_min=2.5, color_by_rgb=False, color_by_confidence=False, samples=256):
        """reconstruct() for meshes too large for one GPU: a distributed sample sort by the reference's Morton key.
        Every rank ends up with one contiguous range of the GLOBAL triangle order -- rank 0 the first triangles,
        rank world-1 the last -- so the ranks' results, concatenated in rank order, are the reference's mesh.
        Returns (verts (n,9) float32, rgb (n,9) uint8 or None, cells (n,) int64, first) on every rank, `first` being
        the global index of this rank's first triangle.

        Each rank meshes its slab (sorted by key within the slab), `samples` evenly spaced keys per rank are
        all-gathered and world-1 splitters picked from them, whole cells go to the rank that owns their key range
        (point-to-point, exactly-sized tensors), and a stable local sort merges the received runs: triangles of
        one cell come from one slab, so their emit order survives."""
        self.exchange_halo()
        verts, rgb, cells = self.slab.march_tensors(w_min, color_by_rgb, color_by_confidence)
        if self.world == 1:
            return verts, rgb, cells, 0
        dev = verts.device
        has_rgb = rgb is not None
        keys = morton_x_major_torch(cells)
        order = torch.argsort(keys, stable=True)  # (already sorted within a slab; cheap and makes no assumption)
        keys, verts, cells = keys[order], verts[order], cells[order]
        if has_rgb:
            rgb = rgb[order]
        n = keys.shape[0]
        # splitters: the same on every rank
        mine = torch.full((samples,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
        if n:
            mine = keys[(torch.arange(samples, device=dev) * n) // samples]
        gathered = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(gathered, mine, group=self.group)
        pool = torch.sort(torch.cat(gathered)).values
        pool = pool[pool != torch.iinfo(torch.int64).max]
        if pool.numel() == 0:
            return verts, rgb, cells, 0
        split = pool[(torch.arange(1, self.world, device=dev) * pool.numel()) // self.world]
        # my triangles with split[q-1] <= key < split[q] go to rank q (all triangles of a cell share the key)
        cuts = torch.searchsorted(keys, split, right=False)
        bounds = [0] + [int(c) for c in cuts.tolist()] + [n]
        send_counts = torch.tensor([bounds[q + 1] - bounds[q] for q in range(self.world)], dtype=torch.int64, device=dev)
        table = [torch.empty_like(send_counts) for _ in range(self.world)]
        dist.all_gather(table, send_counts, group=self.group)
        table = torch.stack(table).cpu()                       # table[src][dst]
        recv_counts = [int(table[r][self.rank]) for r in range(self.world)]
        total = sum(recv_counts)
        out_v = torch.empty((total, 9), dtype=torch.float32, device=dev)
        out_c = torch.empty((total, 9), dtype=torch.uint8, device=dev) if has_rgb else None
        out_k = torch.empty((total,), dtype=torch.int64, device=dev)
        ops, off = [], 0
        for r in range(self.world):
            m = recv_counts[r]
            sl = slice(off, off + m)
            off += m
            if m == 0:
                continue
            if r == self.rank:
                src = slice(bounds[r], bounds[r + 1])
                out_v[sl], out_k[sl] = verts[src], cells[src]
                if has_rgb:
                    out_c[sl] = rgb[src]
            else:
                ops.append(("recv", out_v[sl], r))
                ops.append(("recv", out_k[sl], r))
                if has_rgb:
                    ops.append(("recv", out_c[sl], r))
        for q in range(self.world):
            src = slice(bounds[q], bounds[q + 1])
            if q == self.rank or bounds[q + 1] == bounds[q]:
                continue
            ops.append(("send", verts[src].contiguous(), q))
            ops.append(("send", cells[src].contiguous(), q))
            if has_rgb:
                ops.append(("send", rgb[src].contiguous(), q))
        p2p_batch(ops, self.group)
        order = torch.argsort(morton_x_major_torch(out_k), stable=True)
        first = int(table[:, :self.rank].sum())
        return out_v[order], (out_c[order] if has_rgb else None), out_k[order], first

    def save_ply(self, filename, w_min=2.5, color_by_rgb=False, color_by_confidence=False):
        """Mesh the whole grid and write ONE binary PLY in the layout pcl::io::savePLYFileBinary gives a
        PolygonMesh (what the reference's `integrate` / `tsdf2mesh` programs write): after a distributed sort
        every rank knows the global index of its first triangle, so each writes its own vertex and face records
        at their byte offsets -- no rank ever holds the whole mesh.  Needs a file system all ranks share (one
        node).  Vertices are moved by `global_transform` as marching_cubes_tsdf_octree.cpp:119-130 does.
        Returns the total number of triangles."""
        from .volume import transform_points_f64
        verts, rgb, _cells, first = self.reconstruct_distributed(w_min, color_by_rgb, color_by_confidence)
        n = int(verts.shape[0])
        dev = verts.device
        total_t = torch.tensor([n], dtype=torch.int64, device=dev)
        if self.world > 1:
            dist.all_reduce(total_t, op=dist.ReduceOp.SUM, group=self.group)
        total = int(total_t.item())
        has_rgb = rgb is not None
        header = ("ply\nformat binary_little_endian 1.0\ncomment PCL generated\n"
                  f"element vertex {3 * total}\nproperty float x\nproperty float y\nproperty float z\n" +
                  ("property uchar red\nproperty uchar green\nproperty uchar blue\n" if has_rgb else "") +
                  f"element face {total}\nproperty list uchar int vertex_indices\nend_header\n").encode()
        vrec, frec = 12 + (3 if has_rgb else 0), 13
        xyz = verts.cpu().numpy().reshape(-1, 3)
        if not np.array_equal(self.global_transform, np.eye(4)):
            xyz = transform_points_f64(xyz, np.asarray(self.global_transform, dtype=np.float64))
        vbuf = np.zeros((3 * n, vrec), dtype=np.uint8)
        vbuf[:, :12] = np.ascontiguousarray(xyz, dtype="<f4").view(np.uint8).reshape(-1, 12)
        if has_rgb:
            vbuf[:, 12:15] = rgb.cpu().numpy().reshape(-1, 3)
        fbuf = np.zeros((n, frec), dtype=np.uint8)
        fbuf[:, 0] = 3
        idx = (3 * first + np.arange(3 * n, dtype=np.int64)).astype("<i4").reshape(n, 3)
        fbuf[:, 1:] = idx.view(np.uint8).reshape(n, 12)
        if self.rank == 0:  # header + file size first, then everyone writes its ranges
            with open(filename, "wb") as f:
                f.write(header)
                f.truncate(len(header) + 3 * total * vrec + total * frec)
        if self.world > 1:
            dist.barrier(group=self.group)
        fd = os.open(filename, os.O_WRONLY)
        try:
            os.pwrite(fd, vbuf.tobytes(), len(header) + 3 * first * vrec)
            os.pwrite(fd, fbuf.tobytes(), len(header) + 3 * total * vrec + first * frec)
        finally:
            os.close(fd)
        if self.world > 1:
            dist.barrier(group=self.group)
        return total

    # -- getFxn / getGradient / getHessian -------------------------------------------------------------------
    def sample(self, pts, dst=0):
        """Every rank evaluates the points whose 8 neighbours it holds; rank `dst` returns the union."""
        self.exchange_halo()
        ok, val, grad, hess = self.slab.sample(np.asarray(pts, dtype=np.float32))
        if self.world == 1:
            return ok, val, grad, hess
        parts = [None] * self.world if self.rank == dst else None
        dist.gather_object((ok, val, grad, hess), parts, dst=dst, group=self.group)
        if self.rank != dst:
            return None
        ok, val, grad, hess = [np.array(a) for a in parts[0]]
        for o, v, g, h in parts[1:]:
            take = o & ~ok
            val[take], grad[take], hess[take] = v[take], g[take], h[take]
            ok |= o
        return ok, val, grad, hess

    def renderView(self, trans, downsampleBy=1, camera_frame=True, exchange=None):
        """TSDFVolumeOctree::renderView over all slabs (ray hand-off, see the module docstring).  Collective;
        every rank returns the full (H/ds, W/ds, 8) image.
        exchange="allreduce": every rank holds every ray's record, one integer SUM all-reduce of the image-sized
        delta per round (simple; traffic ~ image x rounds).  exchange="p2p": every rank holds only the records it
        is responsible for; after each round a suspended record travels point-to-point to the owner of its next
        voxel (traffic ~ rays crossing a slab boundary), and the finished rays' outputs are summed once at the end.
        Default: "allreduce" for two ranks, "p2p" beyond (the all-reduce moves the whole image up to world + 1 times)."""
        self._flush_pair()
        if self.world == 1:
            return self.slab.render(trans, downsampleBy)
        if exchange is None:
            exchange = "allreduce" if self.world <= 2 else "p2p"
        if exchange not in ("allreduce", "p2p"):
            raise ValueError(f"exchange must be 'allreduce' or 'p2p', not {exchange!r}")
        from .volume import eigen_affine_inverse, transform_cloud_with_normals
        trans = np.asarray(trans, dtype=np.float64)
        ds = int(downsampleBy)
        self.exchange_halo(self.halo, both=True)
        state = self.slab.ray_begin(trans, ds)
        self.last_render_rounds = 0
        W, H = self.slab_image_size()
        if exchange == "p2p":
            out_words = self._render_p2p(trans, ds, state)
        else:
            for _ in range(2 * self.world + 4):
                delta = self.slab.ray_advance(trans, ds, state, self.rank, self.world)
                dist.all_reduce(delta, op=dist.ReduceOp.SUM, group=self.group)
                state = torch.where(delta[:, :1] != 0, delta, state)
                self.last_render_rounds += 1
                if int((state[:, 0] == 1).sum().item()) == 0:
                    break
            else:
                raise RuntimeError("ray hand-off did not converge")
            out_words = state[:, 16:24]
        out = out_words.contiguous().cpu().numpy().view(np.float32).reshape(H // ds, W // ds, 8)
        if camera_frame:
            out = transform_cloud_with_normals(out, eigen_affine_inverse(trans))
        return out

    def _render_p2p(self, trans, ds, state):
        dev, n = state.device, state.shape[0]
        ids = torch.arange(n, device=dev)
        mine = state[(ids % self.world) == self.rank].contiguous()  # before a ray needs a voxel: id mod world
        out = torch.zeros((n, 8), dtype=torch.int32, device=dev)
        bounds = torch.tensor([slab_range(self.nz, self.world, r)[1] for r in range(self.world)], device=dev)
        self.last_p2p_records = 0
        for _ in range(2 * self.world + 4):
            if mine.shape[0]:
                mine = self.slab.ray_advance_list(trans, ds, mine, self.rank, self.world)
            self.last_render_rounds += 1
            done = mine[:, 0] == 2
            fin = mine[done]
            out[fin[:, 11].long()] = fin[:, 16:24]
            sus = mine[~done]
            dest = torch.bucketize(sus[:, 1].long(), bounds, right=True)  # owner of plane z: first rank with z_end > z
            counts = torch.zeros(self.world, dtype=torch.int64, device=dev)
            if sus.shape[0]:
                counts.scatter_add_(0, dest, torch.ones_like(dest))
            table = [torch.zeros_like(counts) for _ in range(self.world)]
            dist.all_gather(table, counts, group=self.group)
            table = torch.stack(table).tolist()  # table[src][dst]
            if sum(sum(row) for row in table) == 0:
                return out_sum(out, self.group)
            ops, parts = [], [sus[dest == self.rank]]
            sends = []
            for r in range(self.world):
                if r == self.rank:
                    continue
                if table[self.rank][r]:
                    sends.append(sus[dest == r].contiguous())
                    ops.append(("send", sends[-1], r))
                    self.last_p2p_records += int(table[self.rank][r])
                if table[r][self.rank]:
                    parts.append(torch.empty((int(table[r][self.rank]), state.shape[1]), dtype=torch.int32, device=dev))
                    ops.append(("recv", parts[-1], r))
            p2p_batch(ops, self.group)
            mine = torch.cat(parts).contiguous()
        raise RuntimeError("ray hand-off did not converge")

    # -- save / load: tsdf_volume_octree.cpp:222-275 -----------------------------------------------------
    def _owners(self, z0, nz):
        """(rank, first plane, plane count) of every slab the planes [z0, z0 + nz) cross."""
        out = []
        for r in range(self.world):
            zb, ze = slab_range(self.nz, self.world, r)
            lo, hi = max(z0, zb), min(z0 + nz, ze)
            if lo < hi:
                out.append((r, lo, hi - lo))
        return out

    def _block_bytes(self, c, n):
        return n * c * c * (11 if self.slab.color else 8)

    def _pack_block(self, d, w, rgb):
        parts = [np.ascontiguousarray(d, np.float32).view(np.uint8).reshape(-1),
                 np.ascontiguousarray(w, np.float32).view(np.uint8).reshape(-1)]
        if self.slab.color:
            parts.append(np.ascontiguousarray(rgb, np.uint8).reshape(-1))
        return np.concatenate(parts)

    def _unpack_block(self, buf, c, n):
        v = n * c * c
        d = buf[:4 * v].view(np.float32).reshape(n, c, c)
        w = buf[4 * v:8 * v].view(np.float32).reshape(n, c, c)
        rgb = buf[8 * v:11 * v].reshape(n, c, c, 3) if self.slab.color else None
        return d, w, rgb

    def _serve_fetch(self, root, x0, y0, z0, c, out=None):
        """One block request of the writer: owners send their planes of the block to `root` (which copies its own);
        on the root `out` = (d, w, rgb) numpy views of the writer's block buffers."""
        dev = self._frame[0].device
        for r, lo, n in self._owners(z0, c):
            if self.rank == r:
                part = self.slab.get_block(x0, y0, lo, c, c, n)
                if r == root:
                    got = part
                else:
                    blob = torch.from_numpy(self._pack_block(*part))
                    send_tensor(blob if _host_staged(self.group) else blob.to(dev), root, self.group)
                    continue
            elif self.rank == root:
                buf = torch.empty(self._block_bytes(c, n), dtype=torch.uint8, device="cpu" if _host_staged(self.group) else dev)
                recv_tensor(buf, r, self.group)
                got = self._unpack_block(buf.cpu().numpy(), c, n)
            else:
                continue
            out[0][lo - z0:lo - z0 + n] = got[0]
            out[1][lo - z0:lo - z0 + n] = got[1]
            if self.slab.color:
                out[2][lo - z0:lo - z0 + n] = got[2]

    def _serve_store(self, root, x0, y0, z0, c, blk=None):
        """One block of the reader: the root sends every owner its planes (and keeps its own)."""
        dev = self._frame[0].device
        for r, lo, n in self._owners(z0, c):
            if self.rank == root:
                part = (blk[0][lo - z0:lo - z0 + n], blk[1][lo - z0:lo - z0 + n],
                        blk[2][lo - z0:lo - z0 + n] if self.slab.color else None)
                if r == root:
                    self._store(x0, y0, lo, *part)
                else:
                    blob = torch.from_numpy(self._pack_block(*part))
                    send_tensor(blob if _host_staged(self.group) else blob.to(dev), r, self.group)
            elif self.rank == r:
                buf = torch.empty(self._block_bytes(c, n), dtype=torch.uint8, device="cpu" if _host_staged(self.group) else dev)
                recv_tensor(buf, root, self.group)
                self._store(x0, y0, lo, *self._unpack_block(buf.cpu().numpy(), c, n))

    def _store(self, x0, y0, z0, d, w, rgb):
        """set_block that never raises in the middle of the collective block protocol (a rank that left the loop would
        leave the root blocked in its next send): the first failure is remembered -- 2 = weights a PACKED slab cannot
        hold (the load is then repeated with float weights), 1 = anything else -- and later blocks are still received
        and dropped; load() agrees on the outcome with one all-reduce at the end."""
        if self._store_failed:
            return
        try:
            self.slab.set_block(x0, y0, z0, d, w, rgb)
        except capi.TsdfHipError as e:
            self._store_failed = 2 if e.code == capi.E_UNSUPPORTED else 1
            self._store_error = str(e)
        except Exception as e:
            self._store_failed = 1
            self._store_error = repr(e)

    def _request(self, root, req=None):
        """The root announces the next block (x0, y0, z0, edge), or edge <= 0 = done / failed; everyone gets it."""
        t = torch.tensor(req if req is not None else [0, 0, 0, 0], dtype=torch.int64, device=self._frame[0].device)
        if self.world > 1:
            dist.broadcast(t, src=root, group=self.group)
        return [int(v) for v in t.cpu()]

    def _drive_blocks(self, root, run, serve):
        """`run(callback)` on the root calls `callback(x0, y0, z0, c, d, w, rgb)` per block; the other ranks follow
        the root's requests.  Returns the root's status on every rank."""
        color = self.slab.color
        if self.rank == root:
            failure = []

            def cb(_user, x0, y0, z0, c, d, w, rgb):
                try:
                    self._request(root, [x0, y0, z0, c])
                    v = c * c * c
                    blk = (np.ctypeslib.as_array(d, (v,)).reshape(c, c, c), np.ctypeslib.as_array(w, (v,)).reshape(c, c, c),
                           np.ctypeslib.as_array(rgb, (3 * v,)).reshape(c, c, c, 3) if color else None)
                    serve(root, x0, y0, z0, c, blk)
                    return 0
                except Exception as e:  # an exception must not unwind through the C caller
                    failure.append(e)
                    return capi.E_INVALID
            rc = run(capi.BLOCK_FN(cb))
            self._request(root, [0, 0, 0, -1 if rc else 0])
            if failure:
                raise failure[0]
            capi.check(rc, "checkpoint")
            return
        while True:
            x0, y0, z0, c = self._request(root)
            if c <= 0:
                if c < 0:
                    raise RuntimeError(f"checkpoint failed on rank {root}")
                return
            serve(root, x0, y0, z0, c)

    def save(self, filename, dst=0):
        """Write the whole grid as one .vol file on rank `dst` (collective).  The file equals the one a single
        handle holding the whole grid would write."""
        self._flush_pair()
        lib = capi.load()
        p = self.slab.params()
        m = capi.TsdfVolMeta()
        m.max_cell_size[:] = list(self.max_cell_size)  # what TSDFVolumeOctree::save writes (max_cell_size_x_ ...)
        m.is_empty = int(self._is_empty)
        m.global_transform[:] = [float(v) for v in np.asarray(self.global_transform, np.float64).reshape(16)]
        self.slab.synchronize()
        self._drive_blocks(dst, lambda cb: lib.tsdf_hip_save_blocks(C.byref(p), C.byref(m), str(filename).encode(), cb, None),
                           self._serve_fetch)

    @classmethod
    def load(cls, filename, group=None, slab_factory=None, halo=None, src=0, configure_more=None, _retry_f32w=False):
        """Collective: rank `src` reads `filename` (written by either side); the volume is built from its header and
        every block goes to the ranks that own its planes.  `configure_more(vol)` is applied after the file's
        settings (device-side choices the file does not carry: setTransformOrder, setLayout)."""
        lib = capi.load()
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        hdr = [None]
        if rank == src:
            p, m = capi.TsdfParams(), capi.TsdfVolMeta()

            def on_header(_user, pp, mm):
                C.memmove(C.byref(p), pp, C.sizeof(p))
                C.memmove(C.byref(m), mm, C.sizeof(m))
                return 0
            rc = lib.tsdf_hip_load_blocks(str(filename).encode(), None, capi.HEADER_FN(on_header), capi.BLOCK_FN(), None)
            hdr[0] = (rc, bytes(p), bytes(m), lib.tsdf_hip_last_error().decode())
        if world > 1:
            dist.broadcast_object_list(hdr, src=src, group=group)
        rc, pb, mb, msg = hdr[0]
        if rc:
            raise capi.TsdfHipError(rc, "load", msg)
        p, m = capi.TsdfParams.from_buffer_copy(pb), capi.TsdfVolMeta.from_buffer_copy(mb)

        def configure(v):
            v.setResolution(*p.res)
            v.setGridSize(*p.size)
            v.setImageSize(p.image_width, p.image_height)
            v.setCameraIntrinsics(p.fx, p.fy, p.cx, p.cy)
            v.setSensorDistanceBounds(p.min_sensor_dist, p.max_sensor_dist)
            v.setDepthTruncationLimits(p.max_dist_pos, p.max_dist_neg)
            v.setWeightTruncationLimit(p.max_weight)
            v.setIntegrateColor(bool(p.integrate_color))
            if configure_more is not None:
                configure_more(v)
        self = cls(configure, p.res[2], group=group, slab_factory=slab_factory, halo=halo)
        self.global_transform = np.array(list(m.global_transform), dtype=np.float64).reshape(4, 4)
        self._is_empty = bool(m.is_empty)
        self.max_cell_size = tuple(m.max_cell_size)
        keep = []  # (the header callback object must outlive the call)

        def run(cb):
            keep.append(capi.HEADER_FN(lambda *_: 0))
            return lib.tsdf_hip_load_blocks(str(filename).encode(), None, keep[0], cb, None)
        self._drive_blocks(src, run, self._serve_store)
        self.slab.synchronize()
        # every rank learns whether any rank failed to store a block (no rank raised inside the protocol)
        status = torch.tensor([self._store_failed], dtype=torch.int32, device=self._frame[0].device)
        if world > 1:
            dist.all_reduce(status, op=dist.ReduceOp.MAX, group=group)
        status = int(status.item())
        if status:
            err = self._store_error
            self.close()
            if status == 2 and not _retry_f32w:
                # weights that are not min(k, max_weight) (a file written with another weighting) do not fit the PACKED
                # layout AUTO picked: read the file again into float weight planes, as tsdf_hip_load does
                def more(v):
                    if configure_more is not None:
                        configure_more(v)
                    v.setLayout(capi.LAYOUT_F32W)
                return cls.load(filename, group=group, slab_factory=slab_factory, halo=halo, src=src, configure_more=more,
                                _retry_f32w=True)
            raise capi.TsdfHipError(capi.E_UNSUPPORTED if status == 2 else capi.E_INVALID, "ZSlabVolume.load",
                                    err or "a block could not be stored on another rank")
        return self

    def slab_image_size(self):
        return self.slab.image_size()

    def download_local(self):
        self._flush_pair()
        return self.slab.vol.download()

    def close(self):
        self.slab.close()
