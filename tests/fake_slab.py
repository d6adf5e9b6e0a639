"""Test-only slab backend for cpu_tsdf_amd.zslab.ZSlabVolume: the CPU oracle instead of the HIP volume, CPU
tensors instead of CUDA tensors, so the N > 1 host logic (slab split, frame broadcast, halo exchange,
mesh merge, sample routing) runs under gloo without a GPU.  Lives in tests/ because the product must
never reach the oracle."""
import numpy as np
import torch

from cpu_tsdf_amd import synth
from oracle.oracle import OracleVolume


class _Cfg:
    """Records what `configure` sets, with the attribute names of capi.TsdfParams."""

    def __init__(self):
        from cpu_tsdf_amd import capi
        self._p = capi.default_params()

    def setResolution(self, x, y, z):
        self._p.res[:] = (x, y, z)

    def setGridSize(self, x, y, z):
        self._p.size[:] = (x, y, z)

    def setImageSize(self, w, h):
        self._p.image_width, self._p.image_height = w, h

    def setCameraIntrinsics(self, fx, fy, cx, cy):
        self._p.fx, self._p.fy, self._p.cx, self._p.cy = fx, fy, cx, cy

    def setSensorDistanceBounds(self, a, b):
        self._p.min_sensor_dist, self._p.max_sensor_dist = a, b

    def setDepthTruncationLimits(self, a, b):
        self._p.max_dist_pos, self._p.max_dist_neg = a, b

    def setWeightTruncationLimit(self, w):
        self._p.max_weight = w

    def setIntegrateColor(self, f):
        self._p.integrate_color = int(bool(f))

    def setTransformOrder(self, o):
        self._p.xform_order = o

    def setLayout(self, layout):
        self._p.layout = layout


class OracleSlab:
    def __init__(self, configure, z_begin, z_end, nz, rank, halo=1):
        self.halo = halo
        cfg = _Cfg()
        configure(cfg)
        self.p = cfg._p
        self.ov = OracleVolume(self.p)  # full-size arrays; only [z_begin, z_end) (+ halo) is ever written
        self.z_begin, self.z_end, self.nz = z_begin, z_end, nz
        self.color = bool(self.p.integrate_color)
        self.res = tuple(self.p.res)

    def params(self):
        from cpu_tsdf_amd import capi
        return capi.TsdfParams.from_buffer_copy(self.p)

    def get_block(self, x0, y0, z0, nx, ny, nz):
        assert self.z_begin - self.halo <= z0 and z0 + nz <= self.z_end + self.halo
        sl = (slice(z0, z0 + nz), slice(y0, y0 + ny), slice(x0, x0 + nx))
        return self.ov.d[sl].copy(), self.ov.w[sl].copy(), (self.ov.rgb[sl].copy() if self.color else None)

    def set_block(self, x0, y0, z0, d, w, rgb):
        nz, ny, nx = d.shape
        assert self.z_begin <= z0 and z0 + nz <= self.z_end
        from cpu_tsdf_amd import capi
        if self.p.layout != capi.LAYOUT_F32W and 0 <= self.p.max_weight <= 255:
            # what the HIP slab does in the PACKED layout AUTO resolves to: a weight no observation count represents
            # is refused (tsdf_hip_upload -> TSDF_HIP_E_UNSUPPORTED)
            wmax = np.float32(self.p.max_weight)
            if not (((w == np.floor(w)) & (w >= 0) & (w < np.ceil(wmax))) | (w == wmax)).all():
                raise capi.TsdfHipError(capi.E_UNSUPPORTED, "upload", "weight is not min(k, max_weight)")
        sl = (slice(z0, z0 + nz), slice(y0, y0 + ny), slice(x0, x0 + nx))
        self.ov.d[sl], self.ov.w[sl] = d, w
        if self.color:
            self.ov.rgb[sl] = rgb

    def frame_buffers(self):
        H, W = self.p.image_height, self.p.image_width
        return torch.empty((H, W), dtype=torch.float32), (torch.empty((H, W, 4), dtype=torch.uint8) if self.color else None)

    def integrate_tensor(self, depth, bgra, trans):
        # as the HIP slab does by default: the reference's integrateCloud incl. its frustum cull (hpp:93-94), this rank's planes
        self.ov.integrate_culled(depth.numpy(), bgra.numpy() if bgra is not None else None, np.asarray(trans, dtype=np.float64),
                                 synth.cam_from_vol_f32(trans), self.z_begin, self.z_end)

    def _pack_rgb(self, rgb):
        c = rgb.astype(np.int32)
        return torch.from_numpy(c[..., 0] | (c[..., 1] << 8) | (c[..., 2] << 16))

    def get_planes(self, z0, nz):
        d = torch.from_numpy(self.ov.d[z0:z0 + nz].copy())
        w = torch.from_numpy(self.ov.w[z0:z0 + nz].copy())
        return d, w, (self._pack_rgb(self.ov.rgb[z0:z0 + nz]) if self.color else None)

    def plane_buffers(self, nz):
        nx, ny, _ = self.res
        d = torch.empty((nz, ny, nx), dtype=torch.float32)
        return d, torch.empty_like(d), (torch.empty((nz, ny, nx), dtype=torch.int32) if self.color else None)

    def set_planes(self, z0, d, w, rgb):
        n = d.shape[0]
        self.ov.d[z0:z0 + n] = d.numpy()
        self.ov.w[z0:z0 + n] = w.numpy()
        if rgb is not None:
            c = rgb.numpy()
            self.ov.rgb[z0:z0 + n] = np.stack([c & 255, (c >> 8) & 255, (c >> 16) & 255], -1).astype(np.uint8)

    def march(self, w_min, by_rgb, by_confidence):
        mode = 2 if by_confidence else (1 if by_rgb else 0)
        verts, rgb, cells = self.ov.march(w_min, mode)
        z = (cells & np.uint64(0x1FFFFF)).astype(np.int64)
        keep = (z >= self.z_begin) & (z < self.z_end)  # a rank meshes the cells whose base voxel it owns
        k3 = np.repeat(keep, 3)
        return {"vertices": verts[k3], "rgb": rgb[k3] if mode else None, "cells": cells[keep]}

    def march_tensors(self, w_min, by_rgb, by_confidence):
        part = self.march(w_min, by_rgb, by_confidence)
        n = len(part["cells"])
        rgb = torch.from_numpy(np.ascontiguousarray(part["rgb"].reshape(n, 9))) if part["rgb"] is not None else None
        return (torch.from_numpy(np.ascontiguousarray(part["vertices"].reshape(n, 9))), rgb,
                torch.from_numpy(part["cells"].astype(np.int64)))

    def sample(self, pts):
        ok, val, grad, hess = self.ov.sample(pts)
        # only answer for points whose lower-corner plane is ours (the HIP slab cannot see the others)
        p = self.p
        zi = np.floor((pts[:, 2].astype(np.float64) + p.size[2] / 2.0) / p.size[2] * p.res[2]).astype(np.int64)
        ctr = ((zi + 0.5) * p.size[2] / p.res[2] - np.float32(p.size[2] / 2.0)).astype(np.float32)
        zi = zi - (pts[:, 2] < ctr)
        mine = (zi >= self.z_begin) & (zi < self.z_end)
        return ok & mine, val, grad, hess

    def image_size(self):
        return self.p.image_width, self.p.image_height

    def render(self, trans, ds):
        from cpu_tsdf_amd.volume import eigen_affine_inverse, transform_cloud_with_normals
        return transform_cloud_with_normals(self.ov.raycast(trans, ds), eigen_affine_inverse(np.asarray(trans, np.float64)))

    def ray_begin(self, trans, ds):
        return torch.from_numpy(self.ov.raycast_begin(trans, ds))

    def ray_advance(self, trans, ds, state, rank, world):
        lo, hi = max(0, self.z_begin - self.halo), min(self.nz, self.z_end + self.halo)
        delta, bad = self.ov.raycast_advance(trans, ds, state.numpy(), rank, world, self.z_begin, self.z_end, lo, hi)
        if bad:
            raise RuntimeError(f"ray hand-off: {bad} rays read planes outside the slab's halo [{lo}, {hi})")
        return torch.from_numpy(delta)

    def ray_advance_list(self, trans, ds, records, rank, world):
        lo, hi = max(0, self.z_begin - self.halo), min(self.nz, self.z_end + self.halo)
        rec = np.ascontiguousarray(records.numpy())
        bad = self.ov.raycast_advance_list(trans, ds, rec, rank, world, self.z_begin, self.z_end, lo, hi) if len(rec) else 0
        if bad:
            raise RuntimeError(f"ray hand-off: {bad} rays read planes outside the slab's halo [{lo}, {hi})")
        return torch.from_numpy(rec)

    def synchronize(self):
        pass

    def close(self):
        pass
