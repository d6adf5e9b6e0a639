"""ctypes binding of the CPU oracle (oracle/tsdf_oracle.c -> oracle/libtsdf_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libtsdf_oracle.so")
SRC = os.path.join(_HERE, "tsdf_oracle.c")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(
            os.path.getmtime(SRC), os.path.getmtime(os.path.join(_HERE, "mc_tables.h"))):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-Wall", "-o", LIB,
                               SRC, "-lm"])
    return LIB


class OracleParams(C.Structure):
    _fields_ = [
        ("res", C.c_int32 * 3), ("size", C.c_float * 3),
        ("max_dist_pos", C.c_float), ("max_dist_neg", C.c_float), ("max_weight", C.c_float),
        ("min_sensor_dist", C.c_float), ("max_sensor_dist", C.c_float),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("image_width", C.c_int32), ("image_height", C.c_int32),
        ("integrate_color", C.c_int32), ("xform_order", C.c_int32),
    ]


_lib = None
_f = C.POINTER(C.c_float)
_u8 = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.oracle_centers.argtypes = [C.c_int, C.c_float, _f]
        L.oracle_integrate.restype = C.c_uint64
        L.oracle_integrate.argtypes = [C.POINTER(OracleParams), _f, _f, _u8, _f, _u8, _f, C.c_int, C.c_int]
        L.oracle_integrate_culled.restype = C.c_uint64
        L.oracle_integrate_culled.argtypes = [C.POINTER(OracleParams), _f, _f, _u8, _f, _u8, _f, C.c_int, C.c_int, _f]
        L.oracle_reference_cull_planes.restype = None
        L.oracle_reference_cull_planes.argtypes = [C.POINTER(OracleParams), C.POINTER(C.c_double), _f]
        L.oracle_expf_many.restype = None
        L.oracle_expf_many.argtypes = [_f, C.c_size_t, _f]
        L.oracle_integrate_variance.restype = C.c_uint64
        L.oracle_integrate_variance.argtypes = [C.POINTER(OracleParams), _f, _f, _u8, _f, C.POINTER(C.c_int32), _f, _u8, _f,
                                                C.c_int, C.c_int, C.c_int]
        L.oracle_integrate_weighted.restype = C.c_uint64
        L.oracle_integrate_weighted.argtypes = [C.POINTER(OracleParams), _f, _f, _u8, _f, _u8, _f, C.c_int, C.c_int, C.c_int]
        L.oracle_integrate_lab.restype = C.c_uint64
        L.oracle_integrate_lab.argtypes = [C.POINTER(OracleParams), _f, _f, _f, _u8, _f, _u8, _f, C.c_int, C.c_int]
        L.oracle_rgb2lab_many.restype = None
        L.oracle_rgb2lab_many.argtypes = [_u8, C.c_size_t, _f]
        L.oracle_lab2rgb_many.restype = None
        L.oracle_lab2rgb_many.argtypes = [_f, C.c_size_t, _u8]
        L.oracle_integrate_rgbn.restype = C.c_uint64
        L.oracle_integrate_rgbn.argtypes = [C.POINTER(OracleParams), _f, _f, _f, _u8, _f, _u8, _f, C.c_int, C.c_int]
        L.oracle_raycast.argtypes = [C.POINTER(OracleParams), _f, _f, _f, _f, C.c_int, _f]
        L.oracle_raycast_begin.argtypes = [C.POINTER(OracleParams), _f, _f, C.c_int, C.c_void_p]
        L.oracle_raycast_advance.restype = C.c_int
        L.oracle_raycast_advance.argtypes = [C.POINTER(OracleParams), _f, _f, _f, _f, C.c_int, C.POINTER(C.c_int),
                                             C.c_void_p, C.c_void_p]
        L.oracle_organize.restype = C.c_uint64
        L.oracle_organize.argtypes = [C.POINTER(OracleParams), _f, C.c_size_t, _u8, C.c_size_t, C.c_size_t, C.c_float,
                                      C.c_int, C.POINTER(C.c_double), _f, _u8]
        L.oracle_raycast_advance_list.restype = C.c_int
        L.oracle_raycast_advance_list.argtypes = [C.POINTER(OracleParams), _f, _f, _f, _f, C.c_int, C.POINTER(C.c_int),
                                                  C.c_void_p, C.c_int]
        L.oracle_sample_batch.argtypes = [C.POINTER(OracleParams), _f, _f, C.c_size_t, _f, _f, _f, _u8]
        L.oracle_march.restype = C.c_uint64
        L.oracle_march.argtypes = [C.POINTER(OracleParams), _f, _f, _u8, C.c_float, C.c_int, _f, _u8,
                                   C.POINTER(C.c_uint64), C.c_uint64]
        L.oracle_march_box.restype = C.c_uint64
        L.oracle_march_box.argtypes = [C.POINTER(OracleParams)] + [C.POINTER(C.c_int)] * 4 + [_f, _f, _u8, C.c_float, C.c_int, _f, _u8,
                                       C.POINTER(C.c_uint64), C.c_uint64]
        L.oracle_trilinear.restype = C.c_float
        L.oracle_trilinear.argtypes = [C.POINTER(OracleParams), _f, _f, C.c_float, C.c_float, C.c_float,
                                       C.POINTER(C.c_int)]
        L.oracle_containing.restype = C.c_int
        L.oracle_containing.argtypes = [C.POINTER(OracleParams), C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f) if a is not None else None


def _bp(a):
    return a.ctypes.data_as(_u8) if a is not None else None


def params_from(p):
    """Copy the shared fields of a capi.TsdfParams (or any object with the same attributes)."""
    o = OracleParams()
    for name, _ in OracleParams._fields_:
        v = getattr(p, name)
        if name in ("res", "size"):
            getattr(o, name)[:] = list(v)
        else:
            setattr(o, name, v)
    return o


class OracleVolume:
    """Dense CPU volume driven by the C restatement; arrays are [z][y][x]."""

    def __init__(self, params, adopt=None):
        """adopt = (d, w, rgb or None): wrap existing whole-grid arrays instead of allocating a reset volume
        (used to run the oracle's raycast / sampling / meshing on grids too large to fuse on the CPU)."""
        self.p = params_from(params)
        nx, ny, nz = self.p.res
        if adopt is not None:
            self.d, self.w, self.rgb = adopt
            assert self.d.shape == (nz, ny, nx) and self.d.dtype == np.float32 and self.d.flags.c_contiguous
            assert self.w.shape == (nz, ny, nx) and self.w.dtype == np.float32 and self.w.flags.c_contiguous
            return
        self.d = np.full((nz, ny, nx), -1.0, dtype=np.float32)  # tsdf_volume_octree.cpp:217
        self.w = np.zeros((nz, ny, nx), dtype=np.float32)
        self.rgb = np.zeros((nz, ny, nx, 3), dtype=np.uint8) if self.p.integrate_color else None

    def node_size(self, axis):
        """tsdf_oracle.c node_size(): the reference's octree is a cube of edge size_x (octree.h:63-66)."""
        r = self.p.res[0]
        cubic_pow2 = r > 0 and (r & (r - 1)) == 0 and self.p.res[1] == r and self.p.res[2] == r
        return self.p.size[0] if cubic_pow2 else self.p.size[axis]

    def centers(self, axis):
        out = np.empty(self.p.res[axis], dtype=np.float32)
        lib().oracle_centers(self.p.res[axis], self.node_size(axis), _fp(out))
        return out

    def integrate(self, depth, bgra, cam_from_vol, z_begin=0, z_end=0, weight_by_depth=False):
        """weight_by_depth: hpp:200-202 (a flag only a loaded .vol can carry)."""
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        T = np.ascontiguousarray(cam_from_vol, dtype=np.float32).reshape(12)
        col = np.ascontiguousarray(bgra, dtype=np.uint8) if bgra is not None else None
        return int(lib().oracle_integrate_weighted(C.byref(self.p), _fp(self.d), _fp(self.w), _bp(self.rgb), _fp(depth),
                                                   _bp(col), _fp(T), z_begin, z_end, int(bool(weight_by_depth))))

    def reference_cull_planes(self, trans):
        """The six planes (l, r, t, b, far, near) of the reference's frustum cull for the forward pose `trans`."""
        t = np.ascontiguousarray(np.asarray(trans, np.float64).reshape(16))
        planes = np.empty(24, np.float32)
        lib().oracle_reference_cull_planes(C.byref(self.p), t.ctypes.data_as(C.POINTER(C.c_double)), _fp(planes))
        return planes

    def integrate_culled(self, depth, bgra, trans, cam_from_vol, z_begin=0, z_end=0):
        """integrateCloud as the reference runs it: getFrustumCulledVoxels first (tsdf_volume_octree.cpp:619-652)."""
        depth = np.ascontiguousarray(depth, np.float32)
        col = np.ascontiguousarray(bgra, np.uint8) if bgra is not None else None
        planes = self.reference_cull_planes(trans)
        return int(lib().oracle_integrate_culled(C.byref(self.p), _fp(self.d), _fp(self.w), _bp(self.rgb) if self.rgb is not None else None,
                                                 _fp(depth), _bp(col) if col is not None else None,
                                                 _fp(np.ascontiguousarray(cam_from_vol, np.float32)), z_begin, z_end, _fp(planes)))

    def integrate_variance(self, depth, bgra, cam_from_vol, weight_by_depth=False, z_begin=0, z_end=0):
        """integrateCloud with weight_by_variance_ (hpp:203-204); self.M / self.nsample = OctreeNode::M_ / nsample_
        (created at zero on first use, or set by the caller from a loaded .vol)."""
        if getattr(self, "M", None) is None:
            self.M = np.zeros_like(self.d)
            self.nsample = np.zeros(self.d.shape, np.int32)
        depth = np.ascontiguousarray(depth, np.float32)
        col = np.ascontiguousarray(bgra, np.uint8) if bgra is not None else None
        return int(lib().oracle_integrate_variance(C.byref(self.p), _fp(self.d), _fp(self.w), _bp(self.rgb) if self.rgb is not None else None,
                                                   _fp(self.M), self.nsample.ctypes.data_as(C.POINTER(C.c_int32)), _fp(depth),
                                                   _bp(col) if col is not None else None,
                                                   _fp(np.ascontiguousarray(cam_from_vol, np.float32)), z_begin, z_end,
                                                   int(weight_by_depth)))

    def integrate_rgbn(self, depth, bgra, cam_from_vol, z_begin=0, z_end=0):
        """integrate with RGBNormalized voxels (setColorMode("RGBNormalized")); self.cn holds r_n, g_n, b_n, i and
        self.rgb what getRGB() returns."""
        if not hasattr(self, "cn"):
            self.cn = np.zeros((4,) + self.d.shape, dtype=np.float32)
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        T = np.ascontiguousarray(cam_from_vol, dtype=np.float32).reshape(12)
        col = np.ascontiguousarray(bgra, dtype=np.uint8)
        return int(lib().oracle_integrate_rgbn(C.byref(self.p), _fp(self.d), _fp(self.w), _fp(self.cn), _bp(self.rgb),
                                               _fp(depth), _bp(col), _fp(T), z_begin, z_end))

    def integrate_lab(self, depth, bgra, cam_from_vol, z_begin=0, z_end=0):
        """integrate with LABNode voxels (setColorMode("LAB")); self.cn holds the L, A, B means and self.rgb what
        getRGB() returns (LAB2RGB of the means)."""
        if not hasattr(self, "cn"):
            self.cn = np.zeros((3,) + self.d.shape, dtype=np.float32)
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        T = np.ascontiguousarray(cam_from_vol, dtype=np.float32).reshape(12)
        col = np.ascontiguousarray(bgra, dtype=np.uint8)
        return int(lib().oracle_integrate_lab(C.byref(self.p), _fp(self.d), _fp(self.w), _fp(self.cn), _bp(self.rgb),
                                              _fp(depth), _bp(col), _fp(T), z_begin, z_end))

    def raycast(self, trans, ds=1):
        trans = np.asarray(trans, dtype=np.float64)
        rot = np.ascontiguousarray(trans[:3, :3].astype(np.float32).reshape(9))
        org = np.ascontiguousarray(trans[:3, 3].astype(np.float32))
        nh, nw = self.p.image_height // ds, self.p.image_width // ds
        out = np.empty((nh, nw, 8), dtype=np.float32)
        lib().oracle_raycast(C.byref(self.p), _fp(self.d), _fp(self.w), _fp(rot), _fp(org), ds, _fp(out))
        return out

    def organize(self, xyz, bgra=None, units=1.0, zero_nans=False, world_to_cam=None):
        """integrate.cpp:559-618 on an (n, >=3) float32 array (+ (n, 4) uint8 b,g,r,a); returns (depth, bgra, filled)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n, stride = xyz.shape
        H, W = self.p.image_height, self.p.image_width
        depth = np.empty((H, W), np.float32)
        out_c = np.empty((H, W, 4), np.uint8)
        col = np.ascontiguousarray(bgra, dtype=np.uint8) if bgra is not None else None
        tf = None
        if world_to_cam is not None:
            tf = np.ascontiguousarray(np.asarray(world_to_cam, np.float64)[:3, :4]).ctypes.data_as(C.POINTER(C.c_double))
        filled = lib().oracle_organize(C.byref(self.p), _fp(xyz), stride, _bp(col), col.shape[1] if col is not None else 0,
                                       n, units, int(zero_nans), tf, _fp(depth), _bp(out_c))
        return depth, out_c, int(filled)

    RAY_REC = 24

    def _rot_org(self, trans):
        trans = np.asarray(trans, dtype=np.float64)
        return (np.ascontiguousarray(trans[:3, :3].astype(np.float32).reshape(9)),
                np.ascontiguousarray(trans[:3, 3].astype(np.float32)))

    def raycast_begin(self, trans, ds=1):
        """Start records of the multi-slab ray hand-off (include/tsdf_hip.h, tsdf_hip_raycast_begin)."""
        rot, org = self._rot_org(trans)
        n = (self.p.image_height // ds) * (self.p.image_width // ds)
        state = np.empty((n, self.RAY_REC), dtype=np.int32)
        lib().oracle_raycast_begin(C.byref(self.p), _fp(rot), _fp(org), ds, state.ctypes.data_as(C.c_void_p))
        return state

    def raycast_advance(self, trans, ds, state, rank, world, z_begin, z_end, alloc_lo, alloc_hi):
        """One hand-off round of one slab; returns (delta, rays that read outside [alloc_lo, alloc_hi))."""
        rot, org = self._rot_org(trans)
        state = np.ascontiguousarray(state, dtype=np.int32)
        delta = np.empty_like(state)
        slab = (C.c_int * 6)(rank, world, z_begin, z_end, alloc_lo, alloc_hi)
        bad = lib().oracle_raycast_advance(C.byref(self.p), _fp(self.d), _fp(self.w), _fp(rot), _fp(org), ds, slab,
                                           state.ctypes.data_as(C.c_void_p), delta.ctypes.data_as(C.c_void_p))
        return delta, int(bad)

    def raycast_advance_list(self, trans, ds, records, rank, world, z_begin, z_end, alloc_lo, alloc_hi):
        """Compact-list form (tsdf_hip_raycast_advance_list): `records` (k, 24) int32 updated in place."""
        rot, org = self._rot_org(trans)
        assert records.dtype == np.int32 and records.flags.c_contiguous
        slab = (C.c_int * 6)(rank, world, z_begin, z_end, alloc_lo, alloc_hi)
        return int(lib().oracle_raycast_advance_list(C.byref(self.p), _fp(self.d), _fp(self.w), _fp(rot), _fp(org), ds, slab,
                                                     records.ctypes.data_as(C.c_void_p), len(records)))

    def sample(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
        n = len(pts)
        val = np.empty(n, np.float32)
        grad = np.empty((n, 3), np.float32)
        hess = np.empty((n, 9), np.float32)
        ok = np.empty(n, np.uint8)
        lib().oracle_sample_batch(C.byref(self.p), _fp(self.d), _fp(pts), n, _fp(val), _fp(grad), _fp(hess), _bp(ok))
        return ok.astype(bool), val, grad, hess.reshape(n, 3, 3)

    def march(self, w_min, color_mode=0):
        cap = 1 << 16
        while True:
            verts = np.empty((cap, 9), np.float32)
            rgb = np.empty((cap, 9), np.uint8)
            cells = np.empty(cap, np.uint64)
            n = int(lib().oracle_march(C.byref(self.p), _fp(self.d), _fp(self.w), _bp(self.rgb), w_min, color_mode,
                                       _fp(verts), _bp(rgb), cells.ctypes.data_as(C.POINTER(C.c_uint64)), cap))
            if n <= cap:
                return verts[:n].reshape(n * 3, 3), rgb[:n].reshape(n * 3, 3), cells[:n]
            cap = n


def march_box(params, org, d, w, rgb, clo, chi, w_min, color_mode=0):
    """Marching cubes of the cells with base voxel in [clo, chi) (x, y, z) of the grid `params` describes, from arrays
    d / w / rgb ([z][y][x]) that hold only the voxels starting at `org` (x, y, z): the triangles the whole-grid mesh
    has for those cells, in its order.  How full-size GPU volumes are compared box by box."""
    p = params if isinstance(params, OracleParams) else params_from(params)
    d = np.ascontiguousarray(d, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    rgb = np.ascontiguousarray(rgb, np.uint8) if rgb is not None else None
    dim = (C.c_int * 3)(d.shape[2], d.shape[1], d.shape[0])
    i3 = lambda v: (C.c_int * 3)(*[int(x) for x in v])
    cap = 1 << 16
    while True:
        verts = np.empty((cap, 9), np.float32)
        col = np.empty((cap, 9), np.uint8)
        cells = np.empty(cap, np.uint64)
        n = int(lib().oracle_march_box(C.byref(p), i3(org), dim, i3(clo), i3(chi), _fp(d), _fp(w), _bp(rgb) if rgb is not None else None,
                                       w_min, color_mode, _fp(verts), _bp(col), cells.ctypes.data_as(C.POINTER(C.c_uint64)), cap))
        if n == 2 ** 64 - 1:
            raise ValueError("march_box: the arrays do not cover the requested cells")
        if n <= cap:
            return verts[:n].reshape(n * 3, 3), col[:n].reshape(n * 3, 3), cells[:n]
        cap = n


def cells_in_box(cells, clo, chi):
    """Mask of the mesh's per-triangle cell keys (x<<42 | y<<21 | z) whose base voxel lies in [clo, chi)."""
    x, y, z = (cells >> np.uint64(42)).astype(np.int64), ((cells >> np.uint64(21)) & np.uint64(0x1fffff)).astype(np.int64), \
        (cells & np.uint64(0x1fffff)).astype(np.int64)
    return (x >= clo[0]) & (x < chi[0]) & (y >= clo[1]) & (y < chi[1]) & (z >= clo[2]) & (z < chi[2])


class SlabOracle:
    """Dense oracle restricted to planes [zb, ze): arrays hold only the slab."""

    def __init__(self, p, zb, ze):
        self.p = params_from(p)
        self.zb, self.ze = zb, ze
        nx, ny, _ = p.res
        n = ze - zb
        self.d = np.full((n, ny, nx), -1, np.float32)
        self.w = np.zeros((n, ny, nx), np.float32)
        self.rgb = np.zeros((n, ny, nx, 3), np.uint8) if p.integrate_color else None

    def integrate(self, depth, bgra, T):
        nx, ny, _ = self.p.res
        off = self.zb * ny * nx
        # the oracle indexes [z][y][x] from plane 0: hand it pointers shifted back by zb planes
        fp = C.POINTER(C.c_float)
        d = C.cast(self.d.ctypes.data - 4 * off, fp)
        w = C.cast(self.w.ctypes.data - 4 * off, fp)
        rgb = C.cast(self.rgb.ctypes.data - 3 * off, C.POINTER(C.c_uint8)) if self.rgb is not None else None
        depth = np.ascontiguousarray(depth, np.float32)
        col = np.ascontiguousarray(bgra, np.uint8) if bgra is not None else None
        T = np.ascontiguousarray(T, np.float32)
        return lib().oracle_integrate(C.byref(self.p), d, w, rgb, depth.ctypes.data_as(fp),
                                        col.ctypes.data_as(C.POINTER(C.c_uint8)) if col is not None else None,
                                        T.ctypes.data_as(fp), self.zb, self.ze)


def expf(x):
    """The host libm's expf on a float32 array (what std::exp(float) is in the reference, hpp:204)."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().oracle_expf_many(_fp(x), x.size, _fp(out))
    return out


def rgb2lab(rgb):
    """RGB2LAB (octree.cpp:436-481) of an (n,3) uint8 array of r,g,b -> (n,3) float32 L,A,B."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)
    out = np.empty(rgb.shape, dtype=np.float32)
    lib().oracle_rgb2lab_many(_bp(rgb), rgb.shape[0], _fp(out))
    return out


def lab2rgb(lab):
    """LAB2RGB (octree.cpp:483-527) of an (n,3) float32 array -> (n,3) uint8."""
    lab = np.ascontiguousarray(lab, dtype=np.float32).reshape(-1, 3)
    out = np.empty(lab.shape, dtype=np.uint8)
    lib().oracle_lab2rgb_many(_fp(lab), lab.shape[0], _bp(out))
    return out
