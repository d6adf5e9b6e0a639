"""Python-side mirror of the reference's ``TSDFVolumeOctree`` / ``MarchingCubesTSDFOctree`` interface
for the accelerated path, on top of the C ABI (``include/tsdf_hip.h``).

Method names, argument meaning and defaults follow the reference
(``include/cpu_tsdf/tsdf_volume_octree.h``, ``src/lib/tsdf_volume_octree.cpp:54-199``): configure
with setters, then ``reset()``, then ``integrateCloud`` per frame.  The organised point cloud of the
reference becomes a ``(H, W)`` float32 depth image (``pt.z``; NaN = no return) plus an optional
``(H, W, 4)`` uint8 image in PCL ``PointXYZRGBA`` byte order.  All compute happens in HIP kernels; a
missing ``libtsdf_hip.so`` or GPU raises.
"""
import ctypes as C

import numpy as np

from . import capi
from .synth import cam_from_vol_f32, eigen_affine_inverse


class NonCubicQueryWarning(UserWarning):
    """renderView / getFxn / reconstruct on a non-cubic setGridSize: answered on the flat grid's geometry (see
    TSDFVolumeOctree._cubic_for_queries)."""


class TSDFVolumeOctree:
    """Drop-in for ``cpu_tsdf::TSDFVolumeOctree`` (flat SoA grid in HBM instead of an octree)."""

    def __init__(self):
        self._p = capi.default_params()  # reference constructor defaults, tsdf_volume_octree.cpp:54-85
        self._h = None
        self._global_transform = np.eye(4)
        self._is_empty = True
        self._stream = None

    # -- configuration (tsdf_volume_octree.cpp:92-199, tsdf_volume_octree.h:101-190) ----------------
    def setResolution(self, xres, yres, zres):
        self._p.res[:] = (int(xres), int(yres), int(zres))

    def getResolution(self):
        return tuple(self._p.res)

    def setGridSize(self, xsize, ysize, zsize):
        self._p.size[:] = (float(xsize), float(ysize), float(zsize))

    def getGridSize(self):
        return tuple(self._p.size)

    def setImageSize(self, width, height):
        self._p.image_width, self._p.image_height = int(width), int(height)

    def getImageSize(self):
        return self._p.image_width, self._p.image_height

    def setDepthTruncationLimits(self, max_dist_pos, max_dist_neg):
        self._p.max_dist_pos, self._p.max_dist_neg = float(max_dist_pos), float(max_dist_neg)

    def getDepthTruncationLimits(self):
        return self._p.max_dist_pos, self._p.max_dist_neg

    def setWeightTruncationLimit(self, max_weight):
        self._p.max_weight = float(max_weight)

    def getWeightTruncationLimit(self):
        return self._p.max_weight

    def setGlobalTransform(self, trans):
        self._global_transform = np.array(trans, dtype=np.float64).reshape(4, 4)

    def getGlobalTransform(self):
        return self._global_transform.copy()

    def setCameraIntrinsics(self, fx, fy, cx, cy):
        self._p.fx, self._p.fy, self._p.cx, self._p.cy = float(fx), float(fy), float(cx), float(cy)

    def getCameraIntrinsics(self):
        return self._p.fx, self._p.fy, self._p.cx, self._p.cy

    def setMaxVoxelSize(self, x, y, z):
        """Accepted for interface compatibility; a dense grid has no coarse cells."""
        self._max_cell = (float(x), float(y), float(z))

    def setNumRandomSplts(self, n):
        """Accepted for interface compatibility; there is no pre-split pass on a dense grid."""
        self._num_random_splits = int(n)

    def setIntegrateColor(self, flag):
        self._p.integrate_color = 1 if flag else 0

    def setColorMode(self, color_mode):
        """tsdf_volume_octree.h:290: "RGB" (default), "RGBNormalized" or "LAB" (OctreeNode::instantiateByTypeString,
        octree.cpp:193-206)."""
        modes = {"RGB": capi.COLOR_RGB, "RGBNormalized": capi.COLOR_RGB_NORMALIZED, "LAB": capi.COLOR_LAB}
        if color_mode not in modes:
            raise ValueError(f"colour mode {color_mode!r} does not exist in the HIP volume (have {sorted(modes)})")
        self._p.color_mode = modes[color_mode]

    def setSensorDistanceBounds(self, min_sensor_dist, max_sensor_dist):
        self._p.min_sensor_dist, self._p.max_sensor_dist = float(min_sensor_dist), float(max_sensor_dist)

    def getSensorDistanceBounds(self):
        return self._p.min_sensor_dist, self._p.max_sensor_dist

    def setTransformOrder(self, order):
        """Not in the reference: which PCL ``transformPoint`` summation order to mirror."""
        self._p.xform_order = int(order)

    def setLayout(self, layout):
        """Not in the reference: capi.LAYOUT_AUTO / LAYOUT_F32W / LAYOUT_PACKED (include/tsdf_hip.h)."""
        self._p.layout = int(layout)

    def getLayout(self):
        """What AUTO resolved to (needs reset())."""
        return int(capi.load().tsdf_hip_layout(self._need()))

    def setZSlab(self, z_begin, z_end, halo=0, device=-1):
        """Not in the reference: own only planes [z_begin, z_end) (multi-GPU Z-slab partition)."""
        self._p.z_begin, self._p.z_end, self._p.halo, self._p.device = int(z_begin), int(z_end), int(halo), int(device)

    def setDevices(self, devices):
        """Not in the reference: spread the volume over several GPUs of this node (tsdf_hip_create_multi): Z-slabs, one
        per entry of `devices` (ordinals may repeat); every method then drives all of them from this one process.
        None / [] = one handle on setZSlab's device."""
        self._devices = [int(d) for d in devices] if devices else None

    def slabs(self):
        """[(device, z_begin, z_end, halo)] of the handle(s) behind this volume."""
        lib, h = capi.load(), self._need()
        out = []
        for k in range(lib.tsdf_hip_slab_count(h)):
            v = [C.c_int32() for _ in range(4)]
            capi.check(lib.tsdf_hip_slab_info(h, k, *[C.byref(x) for x in v]), "slab_info")
            out.append(tuple(x.value for x in v))
        return out

    def renderStats(self):
        """Multi-GPU volumes: (rounds, records handed between slabs, bytes moved between slabs, host waits) of the last
        renderView / renderColoredView."""
        out = (C.c_uint64 * 4)()
        capi.check(capi.load().tsdf_hip_multi_render_stats(self._need(), out), "multi_render_stats")
        return tuple(int(v) for v in out)

    def setStream(self, stream_ptr):
        self._stream = stream_ptr
        if self._h:
            capi.check(capi.load().tsdf_hip_set_stream(self._h, C.c_void_p(stream_ptr)), "set_stream")

    def isEmpty(self):
        return self._is_empty

    # -- lifetime ------------------------------------------------------------------------------------
    def reset(self):
        """tsdf_volume_octree.cpp:201-219."""
        lib = capi.load()
        if self._h:
            capi.check(lib.tsdf_hip_destroy(self._h), "destroy")
            self._h = None
        h = C.c_void_p()
        devs = getattr(self, "_devices", None)
        # weight_by_depth_ / weight_by_variance_ (only load() sets them, as in the reference) survive a reset there too
        weighting = tuple(int(v) for v in getattr(self, "_weighting", (0, 0)))
        if any(weighting) and self._p.layout == capi.LAYOUT_AUTO:
            self._p.layout = capi.LAYOUT_F32W
        if devs:
            arr = (C.c_int32 * len(devs))(*devs)
            capi.check(lib.tsdf_hip_create_multi(C.byref(self._p), arr, len(devs), C.byref(h)), "create_multi")
        else:
            capi.check(lib.tsdf_hip_create(C.byref(self._p), C.byref(h)), "create")
        self._h = h
        if any(weighting):
            capi.check(lib.tsdf_hip_set_weighting(h, *weighting), "set_weighting")
        if self._stream is not None and not getattr(self, "_devices", None):  # (a stream belongs to one device)
            capi.check(lib.tsdf_hip_set_stream(self._h, C.c_void_p(self._stream)), "set_stream")
        if getattr(self, "_frame_pairing", False):  # (a multi-GPU set pairs too since round 6: every slab its own ring)
            capi.check(lib.tsdf_hip_set_frame_pairing(self._h, 1), "set_frame_pairing")
        self._is_empty = True

    def close(self):
        if self._h:
            capi.load().tsdf_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _need(self):
        if not self._h:
            raise RuntimeError("call reset() first")
        return self._h

    def synchronize(self):
        capi.check(capi.load().tsdf_hip_synchronize(self._need()), "synchronize")

    # -- hot path ------------------------------------------------------------------------------------
    def setReferenceCull(self, flag):
        """Not in the reference's API, but its behaviour: integrateCloud there only visits the voxels pcl::FrustumCulling
        keeps (getFrustumCulledVoxels, tsdf_volume_octree.cpp:619-652).  Default True: every integrate call hands the six
        planes of its pose to the library (tsdf_hip_set_reference_cull), which applies them wherever they can decide a
        voxel -- this volume's voxels are the reference's in every regime.  False: no cull (every voxel updateVoxel
        itself accepts; identical for ordinary cameras, see referenceCullIsNoop())."""
        self._reference_cull = bool(flag)

    def _apply_reference_cull(self, trans):
        lib, h = capi.load(), self._need()
        if getattr(self, "_reference_cull", True):
            planes = reference_cull_planes(self._p, trans)
            capi.check(lib.tsdf_hip_set_reference_cull(h, capi.as_f32p(planes)), "set_reference_cull")
            self._cull_set = True
        elif getattr(self, "_cull_set", False):
            capi.check(lib.tsdf_hip_set_reference_cull(h, None), "set_reference_cull")
            self._cull_set = False

    def integrateCloud(self, depth, bgra=None, trans=None, count=False, pipelined=False):
        """``integrateCloud(cloud, normals, trans)`` (impl/tsdf_volume_octree.hpp:48-103).  Returns True
        (as the reference always does), or the number of observed voxels when ``count`` is set.
        ``pipelined``: return as soon as the frame sits in pinned staging (tsdf_hip_integrate_async); the upload
        overlaps the previous frame's kernel and every later call on this volume is ordered after it."""
        h = self._need()
        trans = np.eye(4) if trans is None else np.asarray(trans, dtype=np.float64)
        depth = capi.f32c(depth)
        if depth.shape != (self._p.image_height, self._p.image_width):
            raise ValueError("depth image must be (image_height, image_width)")
        self._apply_reference_cull(trans)
        T = cam_from_vol_f32(trans)
        n = C.c_uint64(0)
        col = None
        if self._p.integrate_color:
            if bgra is None:
                raise ValueError("integrate_color is set: a (H, W, 4) uint8 bgra image is required")
            col = np.ascontiguousarray(bgra, dtype=np.uint8)
            if col.shape != (self._p.image_height, self._p.image_width, 4):
                raise ValueError("bgra image must be (image_height, image_width, 4)")
        if pipelined and not count:
            capi.check(capi.load().tsdf_hip_integrate_async(h, capi.as_f32p(depth), capi.as_u8p(col) if col is not None else None,
                                                            capi.as_f32p(T)), "integrate_async")
            self._is_empty = False
            return True
        capi.check(
            capi.load().tsdf_hip_integrate(h, capi.as_f32p(depth), capi.as_u8p(col) if col is not None else None,
                                           capi.as_f32p(T), C.byref(n) if count else None), "integrate")
        self._is_empty = False
        return int(n.value) if count else True

    def integrateCloudDevice(self, depth_ptr, bgra_ptr, trans, count=False):
        """Same, with device pointers (frames already resident in HBM); asynchronous unless count."""
        h = self._need()
        self._apply_reference_cull(np.asarray(trans, dtype=np.float64))
        T = cam_from_vol_f32(np.asarray(trans, dtype=np.float64))
        n = C.c_uint64(0)
        capi.check(
            capi.load().tsdf_hip_integrate_device(h, C.c_void_p(depth_ptr), C.c_void_p(bgra_ptr) if bgra_ptr else None,
                                                  capi.as_f32p(T), C.byref(n) if count else None), "integrate_device")
        self._is_empty = False
        return int(n.value) if count else True

    def setFramePairing(self, flag):
        """Not in the reference: pipelined integrateCloud calls (pipelined=True) are integrated two per kernel sweep where
        both poses see the whole volume (tsdf_hip_set_frame_pairing); a frame waits for its partner until the next
        integrateCloud or any other call on the volume.  Same voxels, bit for bit."""
        self._frame_pairing = bool(flag)
        if self._h:
            capi.check(capi.load().tsdf_hip_set_frame_pairing(self._h, int(self._frame_pairing)), "set_frame_pairing")

    def integrateCloudDevice2(self, frame_a, frame_b, count=False):
        """Two frames in one call (tsdf_hip_integrate_device2): each frame = (depth_ptr, bgra_ptr, trans), device pointers
        with the colour image behind the depth image in one allocation.  The voxels are those of two integrateCloudDevice
        calls in this order; where both poses see the whole slab one kernel sweep does both.  Returns (fused, counts):
        counts = the two frames' observed voxels when `count`, else None."""
        h = self._need()
        args = []
        for dp, cp, trans in (frame_a, frame_b):
            trans = np.asarray(trans, dtype=np.float64)
            planes = reference_cull_planes(self._p, trans) if getattr(self, "_reference_cull", True) else None
            args.append((C.c_void_p(dp), C.c_void_p(cp) if cp else None, np.ascontiguousarray(cam_from_vol_f32(trans).reshape(12)), planes))
        n = (C.c_uint64 * 2)()
        fused = C.c_int32(0)
        (da, ca, ta, pa), (db, cb, tb, pb) = args
        capi.check(capi.load().tsdf_hip_integrate_device2(
            h, da, ca, capi.as_f32p(ta), capi.as_f32p(pa) if pa is not None else None,
            db, cb, capi.as_f32p(tb), capi.as_f32p(pb) if pb is not None else None,
            n if count else None, C.byref(fused)), "integrate_device2")
        self._cull_set = args[1][3] is not None
        self._is_empty = False
        return bool(fused.value), ([int(n[0]), int(n[1])] if count else None)

    def organize(self, xyz, bgra=None, cloud_units=1.0, zero_nans=False, world_to_cam=None, fetch=True):
        """The `integrate` program's per-cloud preparation (src/prog/integrate.cpp:559-618) on the GPU: scale,
        (0,0,0) -> NaN, optional world -> camera transform (4x4 = poses[i].inverse()), z-buffer reprojection
        into an organised frame that stays staged in the volume for integrateStaged().  xyz (n, >=3) float32,
        bgra (n, >=4) uint8 in PCL b,g,r,a order.  Returns (depth, bgra, n_valid) when `fetch`."""
        h = self._need()
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n, stride = xyz.shape
        col = np.ascontiguousarray(bgra, dtype=np.uint8) if bgra is not None else None
        H, W = self._p.image_height, self._p.image_width
        depth = np.empty((H, W), np.float32) if fetch else None
        out_c = np.empty((H, W, 4), np.uint8) if fetch else None
        nv = C.c_uint64(0)
        tf = None
        if world_to_cam is not None:
            tf = np.ascontiguousarray(np.asarray(world_to_cam, dtype=np.float64)[:3, :4])
        capi.check(capi.load().tsdf_hip_organize(
            h, capi.as_f32p(xyz), stride, capi.as_u8p(col) if col is not None else None,
            col.shape[1] if col is not None else 0, n, float(cloud_units), int(bool(zero_nans)),
            tf.ctypes.data_as(C.POINTER(C.c_double)) if tf is not None else None,
            capi.as_f32p(depth) if fetch else None, capi.as_u8p(out_c) if fetch else None,
            C.byref(nv) if fetch else None), "organize")
        return (depth, out_c, int(nv.value)) if fetch else None

    def integrateStaged(self, trans=None, count=False):
        """integrateCloud on the frame the last organize() left in the volume."""
        trans = np.eye(4) if trans is None else np.asarray(trans, dtype=np.float64)
        self._apply_reference_cull(trans)  # (as integrateCloud does: the planes belong to this frame's pose)
        T = np.ascontiguousarray(cam_from_vol_f32(trans).reshape(12))
        c = C.c_uint64(0)
        capi.check(capi.load().tsdf_hip_integrate_staged(self._need(), capi.as_f32p(T), C.byref(c) if count else None),
                   "integrate_staged")
        self._is_empty = False
        return int(c.value) if count else True

    def _cubic_for_queries(self, who):
        """The C++ shell REFUSES the queries on a non-cubic setGridSize (TSDFVolumeOctree::cubicForQueries,
        csrc/host/tsdf_volume_octree.cpp: the reference looks per-axis voxel indices, tsdf_volume_octree.cpp:553-574, up in an
        octree that is a cube of edge size_x, octree.cpp:244-266 -- a mixture of two geometries a flat grid does not
        reproduce).  This front end is also what the Z-slab hosts, the bench's slabs and the full-size tests drive, whose
        grids are deliberately flat and non-cubic, so it answers on the flat-grid geometry -- and says so: a
        NonCubicQueryWarning, once per call site (the two front ends used to disagree silently).
        ``strict_noncubic = True`` on the object turns the warning into the C++ shell's refusal (ValueError)."""
        sx, sy, sz = (float(v) for v in self._p.size)
        if sx == sy and sx == sz:
            return
        msg = (f"TSDFVolumeOctree.{who}: grid size {sx:g} x {sy:g} x {sz:g} is not a cube: the reference mixes per-axis closed-form "
               "indices with an octree that is a cube of edge size_x; this answer is for the flat grid's own geometry "
               "(the C++ drop-in refuses this call)")
        if getattr(self, "strict_noncubic", False):
            raise ValueError(msg)
        import warnings
        warnings.warn(msg, NonCubicQueryWarning, stacklevel=3)

    def renderView(self, trans=None, downsampleBy=1, camera_frame=True, pinned=False):
        """tsdf_volume_octree.cpp:278-424.  Returns (H/ds, W/ds, 8) float32: xyz, normal, t*, iterations.
        With camera_frame (the reference's behaviour) xyz/normal are moved back by trans^-1 (:422).
        pinned: the result is a view of a pinned buffer owned by this volume, which the GPU writes by DMA (no host
        copy); it is overwritten by the next pinned renderView of the same size."""
        self._cubic_for_queries("renderView")
        h = self._need()
        trans = np.eye(4) if trans is None else np.asarray(trans, dtype=np.float64)
        ds = int(downsampleBy)
        nh, nw = self._p.image_height // ds, self._p.image_width // ds
        if pinned:
            cache = self.__dict__.setdefault("_pinned_views", {})
            if (nh, nw) not in cache:
                cache[(nh, nw)] = capi.PinnedArray((nh, nw, 8), np.float32)
            out = cache[(nh, nw)].array
        else:
            out = np.empty((nh, nw, 8), dtype=np.float32)
        rot = np.ascontiguousarray(trans[:3, :3].astype(np.float32).reshape(9))
        org = np.ascontiguousarray(trans[:3, 3].astype(np.float32))
        if camera_frame:  # the final transformPointCloudWithNormals(trans^-1) (:422) runs in the kernel
            inv = np.ascontiguousarray(eigen_affine_inverse(trans)[:3, :4], dtype=np.float64)
            capi.check(capi.load().tsdf_hip_raycast_camera(h, capi.as_f32p(rot), capi.as_f32p(org), ds,
                                                           inv.ctypes.data_as(C.POINTER(C.c_double)), capi.as_f32p(out)),
                       "raycast_camera")
        else:
            capi.check(capi.load().tsdf_hip_raycast(h, capi.as_f32p(rot), capi.as_f32p(org), ds, capi.as_f32p(out)),
                       "raycast")
        return out

    def renderColoredView(self, trans=None, downsampleBy=1):
        """tsdf_volume_octree.cpp:426-450: renderView plus, per hit, the colour of the voxel that contains it
        (one batched device lookup).  Returns (cloud (H/ds, W/ds, 8) in the camera frame, rgb (H/ds, W/ds, 3) uint8);
        misses keep 0,0,0; a volume without integrateColor answers 127,127,127 like the reference's NOCOLOR octree."""
        trans = np.eye(4) if trans is None else np.asarray(trans, dtype=np.float64)
        cloud = self.renderView(trans, downsampleBy, camera_frame=True)
        rgb = np.zeros(cloud.shape[:2] + (3,), dtype=np.uint8)
        hit = ~np.isnan(cloud[..., 2])
        if hit.any():
            # :441  v_t = trans.cast<float>() * point  (Eigen: each coefficient m0*x + (m1*y + m2*z) + t [Eigen-recall])
            m = trans.astype(np.float32)
            p = cloud[..., :3][hit]
            q = np.empty_like(p)
            for r in range(3):
                q[:, r] = (m[r, 0] * p[:, 0] + (m[r, 1] * p[:, 1] + m[r, 2] * p[:, 2])) + m[r, 3]
            q = np.ascontiguousarray(q, dtype=np.float32)
            c = np.empty((len(q), 3), np.uint8)
            found = np.empty(len(q), np.uint8)
            capi.check(capi.load().tsdf_hip_lookup_rgb(self._need(), capi.as_f32p(q), len(q), capi.as_u8p(c), capi.as_u8p(found)),
                       "lookup_rgb")
            if not self._p.integrate_color:
                c[:] = 127
            c[found == 0] = 0
            rgb[hit] = c
        return cloud, rgb

    def getFxn(self, pts):
        """Batched ``getFxn`` (tsdf_volume_octree.cpp:655-672): returns (ok, val)."""
        ok, val, _, _ = self.sample(pts, want_grad=False, want_hess=False)
        return ok, val

    def getGradient(self, pts):
        """Batched ``getGradient`` (tsdf_volume_octree.cpp:681-700): returns (ok, grad (n,3))."""
        ok, _, grad, _ = self.sample(pts, want_hess=False)
        return ok, grad

    def getHessian(self, pts):
        """Batched ``getHessian`` (:703-726): returns (ok, hessian (n,3,3))."""
        ok, _, _, hess = self.sample(pts, want_grad=False)
        return ok, hess

    def getFxnAndGradient(self, pts):
        """:728-738 -- (ok, val, grad)."""
        ok, val, grad, _ = self.sample(pts, want_hess=False)
        return ok, val, grad

    def getFxnGradientAndHessian(self, pts):
        """:740-752 -- (ok, val, grad, hessian)."""
        return self.sample(pts)

    def getVoxelCenter(self, x, y, z):
        """tsdf_volume_octree.cpp:553-560: (i + 0.5) * size / (double)res - size/2 in double, stored as float
        (the closed form; the integrate kernel uses the octree's node centres instead, see centers())."""
        out = []
        for i, a in zip((x, y, z), range(3)):
            size = np.float32(self._p.size[a])
            off = np.float32(np.float64(size) / 2.0)
            out.append(np.float32((np.float64(i) + 0.5) * np.float64(size) / np.float64(self._p.res[a]) - np.float64(off)))
        return tuple(out)

    def getVoxelIndex(self, x, y, z):
        """tsdf_volume_octree.cpp:562-574: (has_voxel, (ix, iy, iz)) with floor(((double)v + size/2) / size * res)."""
        idx = []
        for v, a in zip((x, y, z), range(3)):
            size = np.float64(np.float32(self._p.size[a]))
            q = np.floor((np.float64(np.float32(v)) + size / 2.0) / size * np.float64(self._p.res[a]))
            idx.append(int(q) if np.isfinite(q) and abs(q) < 2 ** 31 else -2 ** 31)  # cvttsd2si on overflow / NaN
        ok = all(0 <= i < r for i, r in zip(idx, self._p.res))
        return ok, tuple(idx)

    def sample(self, pts, want_grad=True, want_hess=True):
        """getFxn / getGradient / getHessian (tsdf_volume_octree.cpp:655-828), batched."""
        self._cubic_for_queries("getFxn / getGradient / getHessian")
        h = self._need()
        pts = capi.f32c(pts).reshape(-1, 3)
        n = pts.shape[0]
        val = np.empty(n, dtype=np.float32)
        grad = np.empty((n, 3), dtype=np.float32) if want_grad else None
        hess = np.empty((n, 9), dtype=np.float32) if want_hess else None
        ok = np.empty(n, dtype=np.uint8)
        capi.check(
            capi.load().tsdf_hip_sample(h, capi.as_f32p(pts), n, capi.as_f32p(val),
                                        capi.as_f32p(grad) if grad is not None else None,
                                        capi.as_f32p(hess) if hess is not None else None, capi.as_u8p(ok)), "sample")
        return ok.astype(bool), val, grad, (hess.reshape(n, 3, 3) if hess is not None else None)

    # -- raw access (parity tests, save) -------------------------------------------------------------
    def download(self, x0=0, y0=0, z0=None, nx=None, ny=None, nz=None, want_rgb=None):
        h = self._need()
        rx, ry, rz = self._p.res
        zb = self._p.z_begin
        ze = self._p.z_end if (self._p.z_begin or self._p.z_end) else rz
        z0 = zb if z0 is None else z0
        nx = rx - x0 if nx is None else nx
        ny = ry - y0 if ny is None else ny
        nz = ze - z0 if nz is None else nz
        d = np.empty((nz, ny, nx), dtype=np.float32)
        w = np.empty((nz, ny, nx), dtype=np.float32)
        want_rgb = bool(self._p.integrate_color) if want_rgb is None else want_rgb
        rgb = np.empty((nz, ny, nx, 3), dtype=np.uint8) if want_rgb else None
        capi.check(
            capi.load().tsdf_hip_download(h, x0, y0, z0, nx, ny, nz, capi.as_f32p(d), capi.as_f32p(w),
                                          capi.as_u8p(rgb) if rgb is not None else None), "download")
        return d, w, rgb

    def referenceCullIsNoop(self):
        """Not in the reference: True if its frustum cull (tsdf_volume_octree.cpp:619-652) cannot change results for the
        configured camera, i.e. this volume's voxels equal the reference's (tsdf_hip_reference_cull_is_noop)."""
        return bool(capi.load().tsdf_hip_reference_cull_is_noop(C.byref(self._p)))

    def downloadColorState(self, z0=None, nz=None):
        """Not in the reference (its members are public): the float colour state of "RGBNormalized" (r_n, g_n, b_n, i)
        or "LAB" (L, A, B) voxels as an array (planes, nz, ny, nx)."""
        h = self._need()
        rx, ry, rz = self._p.res
        zb = self._p.z_begin
        ze = self._p.z_end if (self._p.z_begin or self._p.z_end) else rz
        z0 = zb if z0 is None else z0
        nz = ze - z0 if nz is None else nz
        planes = {capi.COLOR_RGB_NORMALIZED: 4, capi.COLOR_LAB: 3}.get(self._p.color_mode, 0)
        out = np.empty((planes, nz, ry, rx), dtype=np.float32)
        for c in range(planes):
            capi.check(capi.load().tsdf_hip_download_color_state(h, c, z0, nz, capi.as_f32p(out[c])), "colour state")
        return out

    def setWeighting(self, by_depth=False, by_variance=False):
        """Not in the reference (its two flags only arrive through load()): hpp:200-204 on this volume from the next
        reset() on -- F32W layout, plain kernel; tests use it."""
        self._weighting = (bool(by_depth), bool(by_variance))

    def downloadVarianceState(self):
        """(M, nsample) of every voxel: OctreeNode::M_ / nsample_ (octree.h:164-165), kept by volumes that weight by variance."""
        rx, ry, rz = self._p.res
        M, ns = np.empty((rz, ry, rx), np.float32), np.empty((rz, ry, rx), np.int32)
        capi.check(capi.load().tsdf_hip_download_variance_state(self._need(), 0, 0, 0, rx, ry, rz, capi.as_f32p(M),
                                                                ns.ctypes.data_as(C.POINTER(C.c_int32))), "download_variance_state")
        return M, ns

    def uploadVarianceState(self, M, nsample):
        rx, ry, rz = self._p.res
        M, ns = np.ascontiguousarray(M, np.float32), np.ascontiguousarray(nsample, np.int32)
        assert M.shape == ns.shape == (rz, ry, rx)
        capi.check(capi.load().tsdf_hip_upload_variance_state(self._need(), 0, 0, 0, rx, ry, rz, capi.as_f32p(M),
                                                              ns.ctypes.data_as(C.POINTER(C.c_int32))), "upload_variance_state")

    def upload(self, d=None, w=None, rgb=None, x0=0, y0=0, z0=0):
        h = self._need()
        ref = d if d is not None else (w if w is not None else rgb)
        nz, ny, nx = ref.shape[:3]
        d = capi.f32c(d) if d is not None else None
        w = capi.f32c(w) if w is not None else None
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8) if rgb is not None else None
        capi.check(
            capi.load().tsdf_hip_upload(h, x0, y0, z0, nx, ny, nz, capi.as_f32p(d) if d is not None else None,
                                        capi.as_f32p(w) if w is not None else None,
                                        capi.as_u8p(rgb) if rgb is not None else None), "upload")
        self._is_empty = False

    # -- save / load: tsdf_volume_octree.cpp:222-275 ---------------------------------------------------
    def save(self, filename):
        """The reference's .vol checkpoint (tsdf_hip_save); streamed, host memory stays at one 256^3 block."""
        m = capi.TsdfVolMeta()
        # setMaxVoxelSize, else the reference's default 0.5 m (tsdf_volume_octree.cpp:72-74) -- what the C++ class writes
        cell = getattr(self, "_max_cell", None) or (0.5, 0.5, 0.5)
        m.max_cell_size[:] = cell
        m.is_empty = int(self._is_empty)
        m.weight_by_depth, m.weight_by_variance = (int(v) for v in getattr(self, "_weighting", (0, 0)))
        m.global_transform[:] = [float(v) for v in self._global_transform.reshape(16)]
        capi.check(capi.load().tsdf_hip_save(self._need(), str(filename).encode(), C.byref(m)), "save")

    def load(self, filename):
        """Replace this volume by the one in `filename` (written by either side).  Device, layout and
        transform order stay as set on this object."""
        lib = capi.load()
        h, p, m = C.c_void_p(), capi.TsdfParams(), capi.TsdfVolMeta()
        defaults = capi.TsdfParams.from_buffer_copy(self._p)
        defaults.z_begin = defaults.z_end = defaults.halo = 0
        devs = getattr(self, "_devices", None)
        if devs:
            arr = (C.c_int32 * len(devs))(*devs)
            capi.check(lib.tsdf_hip_load_multi(str(filename).encode(), C.byref(defaults), arr, len(devs), C.byref(h), C.byref(p),
                                               C.byref(m)), "load_multi")
        else:
            capi.check(lib.tsdf_hip_load(str(filename).encode(), C.byref(defaults), C.byref(h), C.byref(p), C.byref(m)), "load")
        self.close()
        asked = self._p.layout
        self._h, self._p = h, p
        if not (asked == capi.LAYOUT_AUTO and p.layout == capi.LAYOUT_F32W and 0 <= p.max_weight <= 255):
            self._p.layout = asked
        if self._stream is not None and not getattr(self, "_devices", None):  # (a stream belongs to one device)
            capi.check(lib.tsdf_hip_set_stream(self._h, C.c_void_p(self._stream)), "set_stream")
        self._max_cell = tuple(m.max_cell_size)
        self._is_empty = bool(m.is_empty)
        # hpp:200-204: the handle integrates depth-weighted / refuses to integrate variance-weighted from here on
        self._weighting = (bool(m.weight_by_depth), bool(m.weight_by_variance))
        self._global_transform = np.array(list(m.global_transform), dtype=np.float64).reshape(4, 4)

    def centers(self, axis):
        out = np.empty(self._p.res[axis], dtype=np.float32)
        capi.check(capi.load().tsdf_hip_centers(self._need(), axis, capi.as_f32p(out)), "centers")
        return out

    def device_planes(self):
        d, w, rgb = C.c_void_p(), C.c_void_p(), C.c_void_p()
        pitch, zf, nza = C.c_int64(), C.c_int32(), C.c_int32()
        capi.check(
            capi.load().tsdf_hip_device_planes(self._need(), C.byref(d), C.byref(w), C.byref(rgb), C.byref(pitch),
                                               C.byref(zf), C.byref(nza)), "device_planes")
        return d.value, w.value, rgb.value, pitch.value, zf.value, nza.value


class MarchingCubesTSDFOctree:
    """Drop-in for ``cpu_tsdf::MarchingCubesTSDFOctree`` (include/cpu_tsdf/marching_cubes_tsdf_octree.h)."""

    def __init__(self):
        self._w_min = 2.5  # marching_cubes_tsdf_octree.h:58
        self._by_rgb = False
        self._by_conf = False
        self._vol = None

    def setInputTSDF(self, volume):
        self._vol = volume

    def setMinWeight(self, w_min):
        self._w_min = float(w_min)

    def setColorByRGB(self, flag):
        self._by_rgb = bool(flag)

    def setColorByConfidence(self, flag):
        self._by_conf = bool(flag)

    def reconstruct(self, want_cells=False):
        """marching_cubes_tsdf_octree.cpp:108-143.  Returns dict(vertices (3n,3) float32 after the global
        transform, polygons (n,3) int32 = [3i,3i+1,3i+2], rgb (3n,3) uint8 or None, cells)."""
        vol = self._vol
        vol._cubic_for_queries("MarchingCubesTSDFOctree.reconstruct")
        h = vol._need()
        lib = capi.load()
        mode = 2 if self._by_conf else (1 if self._by_rgb else 0)
        n = C.c_uint64(0)
        capi.check(lib.tsdf_hip_march(h, self._w_min, mode, C.byref(n)), "march")
        nt = int(n.value)
        verts = np.empty((nt * 3, 3), dtype=np.float32)
        rgb = np.empty((nt * 3, 3), dtype=np.uint8) if mode else None
        cells = np.empty(nt, dtype=np.uint64) if want_cells else None
        if nt:
            capi.check(
                lib.tsdf_hip_march_fetch(h, capi.as_f32p(verts), capi.as_u8p(rgb) if rgb is not None else None,
                                         cells.ctypes.data_as(C.POINTER(C.c_uint64)) if cells is not None else None),
                "march_fetch")
        g = vol.getGlobalTransform()
        if not np.array_equal(g, np.eye(4)):
            verts = transform_points_f64(verts, g)
        polys = np.arange(nt * 3, dtype=np.int32).reshape(nt, 3)
        return {"vertices": verts, "polygons": polys, "rgb": rgb, "cells": cells}


def reference_cull_planes(p, trans):
    """The six planes (l, r, t, b, far, near; 4 float32 each) pcl::FrustumCulling::applyFilter builds for the reference's
    getFrustumCulledVoxels (tsdf_volume_octree.cpp:633-646) from the forward pose `trans`: host-side Eigen / PCL
    arithmetic restated, like eigen_affine_inverse -- here by the library's own helper
    (tsdf_hip_reference_cull_planes), since Python has no Eigen to do it with."""
    tr = np.ascontiguousarray(np.asarray(trans, dtype=np.float64).reshape(4, 4))
    planes = np.empty(24, np.float32)
    capi.check(capi.load().tsdf_hip_reference_cull_planes(C.byref(p), tr.ctypes.data_as(C.POINTER(C.c_double)), capi.as_f32p(planes)),
               "reference_cull_planes")
    return planes


def transform_points_f64(xyz, m):
    """pcl::transformPointCloud with an Affine3d on float points [PCL-recall]: evaluated in double,
    x*c0 + (y*c1 + (z*c2 + c3)), rounded to float."""
    p = xyz.astype(np.float64)
    out = np.empty_like(p)
    for r in range(3):
        out[..., r] = p[..., 0] * m[r, 0] + (p[..., 1] * m[r, 1] + (p[..., 2] * m[r, 2] + m[r, 3]))
    return out.astype(np.float32)


def transform_cloud_with_normals(cloud, m):
    """pcl::transformPointCloudWithNormals (tsdf_volume_octree.cpp:422) on an (..., 8) array; with
    is_dense=false non-finite points are left untouched [PCL-recall]."""
    out = cloud.copy()
    xyz, nrm = cloud[..., 0:3], cloud[..., 3:6]
    fin = np.isfinite(xyz).all(-1)
    out[..., 0:3] = np.where(fin[..., None], transform_points_f64(xyz, m), xyz)
    p = nrm.astype(np.float64)
    rn = np.empty_like(p)
    for r in range(3):
        rn[..., r] = p[..., 0] * m[r, 0] + (p[..., 1] * m[r, 1] + p[..., 2] * m[r, 2])
    out[..., 3:6] = np.where(fin[..., None], rn.astype(np.float32), nrm)
    return out
