"""ctypes binding of oracle/_ref/libcpu_tsdf_ref.so -- the REFERENCE's own sources (sdmiller/cpu_tsdf,
unmodified) compiled against the in-repo PCL/Eigen stand-ins by oracle/Makefile, behind the flat C
wrapper oracle/ref_capi.cpp.

TEST INFRASTRUCTURE ONLY: the parity oracle for tests/ and the "reference" CPU baseline of bench.py.
The library is built where /root/reference exists (this container) and travels to the GPU box as a
prebuilt .so; nothing here reads /root/reference at run time.
"""
import ctypes as C
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_ref", "libcpu_tsdf_ref.so")
_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)
_u8 = C.POINTER(C.c_uint8)


DROPIN_LIB = os.path.join(_HERE, "_ref", "libdropin_capi.so")


def available(path=LIB):
    return os.path.exists(path)


def build_dropin(force=False):
    """The drop-in under test, seen through the SAME driver source that wraps the reference:
    oracle/ref_capi.cpp compiled against include/cpu_tsdf/*.h + libcpu_tsdf_hip.so (the product's C++
    host shell).  Output next to the reference build (git-ignored, travels with gpurun)."""
    import subprocess
    from cpu_tsdf_amd import build as b
    shell = b.build_shell()
    src = os.path.join(_HERE, "ref_capi.cpp")
    if not force and os.path.exists(DROPIN_LIB) and os.path.getmtime(DROPIN_LIB) >= max(os.path.getmtime(src),
                                                                                          os.path.getmtime(shell)):
        return DROPIN_LIB
    os.makedirs(os.path.dirname(DROPIN_LIB), exist_ok=True)
    subprocess.check_call(["g++"] + b.HOST_FLAGS + b.host_include_flags() + ["-shared", src, "-L" + b.LIBDIR,
                           "-lcpu_tsdf_hip", "-ltsdf_hip", "-Wl,-rpath,$ORIGIN/../../cpu_tsdf_amd/lib", "-o", DROPIN_LIB])
    return DROPIN_LIB


_libs = {}


def load(path=LIB):
    if path in _libs:
        return _libs[path]
    L = C.CDLL(path)
    L.ct_create.restype = C.c_void_p
    for name, args in {
        "ct_destroy": [C.c_void_p],
        "ct_set_resolution": [C.c_void_p, C.c_int, C.c_int, C.c_int],
        "ct_set_grid_size": [C.c_void_p, C.c_float, C.c_float, C.c_float],
        "ct_set_image_size": [C.c_void_p, C.c_int, C.c_int],
        "ct_set_intrinsics": [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double],
        "ct_set_sensor_bounds": [C.c_void_p, C.c_float, C.c_float],
        "ct_set_trunc": [C.c_void_p, C.c_float, C.c_float],
        "ct_set_max_weight": [C.c_void_p, C.c_float],
        "ct_set_max_voxel_size": [C.c_void_p, C.c_float, C.c_float, C.c_float],
        "ct_set_integrate_color": [C.c_void_p, C.c_int],
        "ct_set_color_mode": [C.c_void_p, C.c_char_p],
        "ct_set_num_random_splits": [C.c_void_p, C.c_int],
        "ct_set_global_transform": [C.c_void_p, _d],
        "ct_reset": [C.c_void_p],
        "ct_save": [C.c_void_p, C.c_char_p],
        "ct_load": [C.c_void_p, C.c_char_p],
    }.items():
        getattr(L, name).argtypes = args
        getattr(L, name).restype = None
    if hasattr(L, "ct_rgb2lab_many"):  # the reference build only
        L.ct_rgb2lab_many.restype = None
        L.ct_rgb2lab_many.argtypes = [_u8, C.c_size_t, _f]
        L.ct_lab2rgb_many.restype = None
        L.ct_lab2rgb_many.argtypes = [_f, C.c_size_t, _u8]
    L.ct_integrate.restype = C.c_double
    L.ct_integrate.argtypes = [C.c_void_p, _f, _u8, C.c_int, C.c_int, _d]
    L.ct_render_view.restype = C.c_double
    L.ct_render_view.argtypes = [C.c_void_p, _d, C.c_int, _f]
    L.ct_render_colored_view.argtypes = [C.c_void_p, _d, C.c_int, _f, _u8]
    L.ct_sample.argtypes = [C.c_void_p, _f, C.c_size_t, _f, _f, _f, _u8]
    L.ct_march.restype = C.c_uint64
    L.ct_march.argtypes = [C.c_void_p, C.c_float, C.c_int, _d]
    L.ct_march_fetch.restype = C.c_uint64
    L.ct_march_fetch.argtypes = [C.c_void_p, _f, _u8, C.POINTER(C.c_uint32)]
    L.ct_mesh_has_color.restype = C.c_int
    L.ct_mesh_has_color.argtypes = [C.c_void_p]
    L.ct_voxel_center.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _f]
    L.ct_voxel_index.restype = C.c_int
    L.ct_voxel_index.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int)]
    L.ct_is_reference.restype = C.c_int
    if L.ct_is_reference():
        L.ct_dump_dense.argtypes = [C.c_void_p, _f, _f, _u8, _f, _f]
        L.ct_num_leaves.restype = C.c_uint64
        L.ct_num_leaves.argtypes = [C.c_void_p]
    else:
        L.ct_download.restype = C.c_int
        L.ct_download.argtypes = [C.c_void_p, _f, _f, _u8]
        L.ct_set_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.ct_set_reference_cull.argtypes = [C.c_void_p, C.c_int]
        L.ct_set_frame_pairing.argtypes = [C.c_void_p, C.c_int]
        L.ct_get_frame_pairing.restype = C.c_int
        L.ct_get_frame_pairing.argtypes = [C.c_void_p]
        L.ct_get_num_devices.restype = C.c_int
        L.ct_get_num_devices.argtypes = [C.c_void_p]
    _libs[path] = L
    return L


def _fp(a):
    return a.ctypes.data_as(_f) if a is not None else None


def _dp(a):
    return a.ctypes.data_as(_d)


class RefVolume:
    """cpu_tsdf::TSDFVolumeOctree through the flat wrapper.  `dense=True` = setMaxVoxelSize(voxel size):
    Octree::init pre-splits to full resolution (src/lib/octree.cpp:593-599), the mode used for parity."""

    def __init__(self, res, size, width, height, fx, fy, cx, cy, zmin, zmax, trunc=(0.03, 0.03), max_weight=100.0,
                 color=False, dense=True, max_cell=0.5, lib_path=LIB, color_mode=None, devices=None, reference_cull=None,
                 frame_pairing=False):
        self.L = load(lib_path)
        self.h = C.c_void_p(self.L.ct_create())
        self.res, self.size, self.W, self.H, self.color = res, float(size), width, height, bool(color)
        L, h = self.L, self.h
        L.ct_set_resolution(h, res, res, res)
        L.ct_set_grid_size(h, size, size, size)
        L.ct_set_image_size(h, width, height)
        L.ct_set_intrinsics(h, fx, fy, cx, cy)
        L.ct_set_sensor_bounds(h, zmin, zmax)
        L.ct_set_trunc(h, *trunc)
        L.ct_set_max_weight(h, max_weight)
        cell = float(np.float32(size) / np.float32(res)) if dense else max_cell
        L.ct_set_max_voxel_size(h, cell, cell, cell)
        L.ct_set_integrate_color(h, int(color))
        if color_mode is not None:
            L.ct_set_color_mode(h, color_mode.encode())
        L.ct_set_num_random_splits(h, 1)
        if reference_cull is not None:  # drop-in build only: TSDFVolumeOctree::setReferenceCull (None = the class's default)
            L.ct_set_reference_cull(h, int(bool(reference_cull)))
        if frame_pairing:  # drop-in build only: TSDFVolumeOctree::setFramePairing
            L.ct_set_frame_pairing(h, 1)
        if devices:  # drop-in build only: TSDFVolumeOctree::setDevices
            L.ct_set_devices(h, (C.c_int * len(devices))(*devices), len(devices))
        L.ct_reset(h)

    def close(self):
        if self.h:
            self.L.ct_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def integrate(self, depth, bgra, trans):
        depth = np.ascontiguousarray(depth, np.float32)
        col = np.ascontiguousarray(bgra, np.uint8) if (bgra is not None and self.color) else None
        tr = np.ascontiguousarray(trans, np.float64).reshape(16)
        return self.L.ct_integrate(self.h, _fp(depth), col.ctypes.data_as(_u8) if col is not None else None, self.W,
                                   self.H, _dp(tr))

    def dump_dense(self):
        n = self.res
        d = np.empty((n, n, n), np.float32)
        w = np.empty((n, n, n), np.float32)
        rgb = np.empty((n, n, n, 3), np.uint8) if self.color else None
        leaf = np.empty((n, n, n), np.float32)
        ctr = np.empty((n, n, n, 3), np.float32)
        self.L.ct_dump_dense(self.h, _fp(d), _fp(w), rgb.ctypes.data_as(_u8) if rgb is not None else None, _fp(leaf),
                             _fp(ctr))
        return d, w, rgb, leaf, ctr

    def download(self):
        """Drop-in build only."""
        n = self.res
        d = np.empty((n, n, n), np.float32)
        w = np.empty((n, n, n), np.float32)
        rgb = np.empty((n, n, n, 3), np.uint8) if self.color else None
        assert self.L.ct_download(self.h, _fp(d), _fp(w), rgb.ctypes.data_as(_u8) if rgb is not None else None)
        return d, w, rgb

    def load(self, path):
        self.L.ct_load(self.h, path.encode())

    def num_leaves(self):
        return int(self.L.ct_num_leaves(self.h))

    def render_view(self, trans, ds=1):
        tr = np.ascontiguousarray(trans, np.float64).reshape(16)
        out = np.empty((self.H // ds, self.W // ds, 8), np.float32)
        dt = self.L.ct_render_view(self.h, _dp(tr), ds, _fp(out))
        return out, dt

    def render_colored_view(self, trans, ds=1):
        tr = np.ascontiguousarray(trans, np.float64).reshape(16)
        out = np.empty((self.H // ds, self.W // ds, 8), np.float32)
        rgb = np.empty((self.H // ds, self.W // ds, 3), np.uint8)
        self.L.ct_render_colored_view(self.h, _dp(tr), ds, _fp(out), rgb.ctypes.data_as(_u8))
        return out, rgb

    def sample(self, pts):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        n = len(pts)
        val, grad, hess = np.empty(n, np.float32), np.empty((n, 3), np.float32), np.empty((n, 9), np.float32)
        ok = np.empty(n, np.uint8)
        self.L.ct_sample(self.h, _fp(pts), n, _fp(val), _fp(grad), _fp(hess), ok.ctypes.data_as(_u8))
        return ok, val, grad, hess.reshape(n, 3, 3)

    def march(self, w_min, color_mode=0):
        sec = C.c_double()
        nv = int(self.L.ct_march(self.h, w_min, color_mode, C.byref(sec)))
        verts = np.empty((nv, 3), np.float32)
        rgb = np.empty((nv, 3), np.uint8) if self.L.ct_mesh_has_color(self.h) else None
        polys = np.empty((nv // 3, 3), np.uint32)
        self.L.ct_march_fetch(self.h, _fp(verts), rgb.ctypes.data_as(_u8) if rgb is not None else None,
                              polys.ctypes.data_as(C.POINTER(C.c_uint32)))
        return verts, rgb, polys, sec.value

    def save(self, path):
        self.L.ct_save(self.h, path.encode())


def time_integrate(sc, res3, size3, color, budget_s, cores):
    """bench.py cpu_baseline leg: the reference in its NATIVE adaptive-octree mode (max cell 0.5 m,
    tsdf_volume_octree.cpp:72-74) on the first frames of the same Scene-A turntable, OpenMP on all
    cores, until `budget_s` seconds of integrateCloud time are spent.  Only cubic grids."""
    from cpu_tsdf_amd import synth
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    res, size = res3[0], size3[0]
    v = RefVolume(res, size, sc.width, sc.height, sc.fx, sc.fy, sc.cx, sc.cy, 0.0, 3.0 * size, color=color,
                  dense=False, max_cell=0.5)
    spent, n, times = 0.0, 0, []
    t_wall = time.perf_counter()
    while spent < budget_s and n < 64 and time.perf_counter() - t_wall < 3 * budget_s:
        tr = synth.turntable_pose(n, 44, sc.size)
        dt = v.integrate(sc.depth(tr), sc.bgra(n) if color else None, tr)
        times.append(dt)
        spent += dt
        n += 1
    leaves = v.num_leaves()
    v.close()
    import resource
    rss_gb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    fps = n / spent
    vox = float(res3[0]) * res3[1] * res3[2]
    return {"value": vox * fps / 1e6, "unit": "Mvoxels/s", "frames_per_s": fps, "cores": cores, "kind": "reference",
            "sample": f"reference TSDFVolumeOctree (own sources + PCL/Eigen stand-ins, -O3 -fopenmp, {cores} threads), "
                      f"native adaptive octree (max cell 0.5 m), first {n} frames of the same {res}^3 workload, "
                      f"{spent:.1f} s in integrateCloud, {leaves} leaves at the end, peak RSS {rss_gb:.1f} GB; nominal-grid Mvoxels/s"}


def ref_rgb2lab(rgb):
    """The reference's own RGB2LAB (octree.cpp:436-481) on an (n,3) uint8 array."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)
    out = np.empty(rgb.shape, dtype=np.float32)
    load().ct_rgb2lab_many(rgb.ctypes.data_as(_u8), rgb.shape[0], _fp(out))
    return out


def ref_lab2rgb(lab):
    """The reference's own LAB2RGB (octree.cpp:483-527) on an (n,3) float32 array."""
    lab = np.ascontiguousarray(lab, dtype=np.float32).reshape(-1, 3)
    out = np.empty(lab.shape, dtype=np.uint8)
    load().ct_lab2rgb_many(_fp(lab), lab.shape[0], out.ctypes.data_as(_u8))
    return out
