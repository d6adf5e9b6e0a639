"""ctypes binding of the C ABI in include/tsdf_hip.h (libtsdf_hip.so).

Fails loudly when the library is missing -- there is no CPU / eager fallback by design.

Two builds of the same sources exist (cpu_tsdf_amd/build.py): libtsdf_hip.so, the product -- exactly the entry points
of include/tsdf_hip.h -- and libtsdf_hip_test.so, the product plus the test hooks of include/tsdf_hip_test.h (device-side
dividers on arbitrary operands, host-side cull predicates, calibration sweeps, run-time tuning knobs).  load() binds the
product unless use_test_library() was called first (tests/conftest.py does; bench.py does for --calib only).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TSDF_HIP_LIB_PATH: an A/B build of the same library (tools/build_variant.py); still no fallback if it is missing
PRODUCT_LIB_PATH = os.path.join(_HERE, "lib", "libtsdf_hip.so")
TEST_LIB_PATH = os.path.join(_HERE, "lib", "libtsdf_hip_test.so")
LIB_PATH = os.environ.get("TSDF_HIP_LIB_PATH") or PRODUCT_LIB_PATH

OK, E_INVALID, E_NOMEM, E_HIP, E_NODEVICE, E_UNSUPPORTED, E_IO = range(7)
XFORM_PCL_SSE, XFORM_LEFT_TO_RIGHT = 0, 1
LAYOUT_AUTO, LAYOUT_F32W, LAYOUT_PACKED = 0, 1, 2
COLOR_RGB, COLOR_RGB_NORMALIZED, COLOR_LAB = 0, 1, 2


class TsdfParams(C.Structure):
    """struct tsdf_params (include/tsdf_hip.h)."""

    _fields_ = [
        ("res", C.c_int32 * 3),
        ("size", C.c_float * 3),
        ("max_dist_pos", C.c_float),
        ("max_dist_neg", C.c_float),
        ("max_weight", C.c_float),
        ("min_sensor_dist", C.c_float),
        ("max_sensor_dist", C.c_float),
        ("fx", C.c_double),
        ("fy", C.c_double),
        ("cx", C.c_double),
        ("cy", C.c_double),
        ("image_width", C.c_int32),
        ("image_height", C.c_int32),
        ("integrate_color", C.c_int32),
        ("xform_order", C.c_int32),
        ("z_begin", C.c_int32),
        ("z_end", C.c_int32),
        ("halo", C.c_int32),
        ("device", C.c_int32),
        ("layout", C.c_int32),
        ("color_mode", C.c_int32),
    ]


class TsdfVolMeta(C.Structure):
    """struct tsdf_vol_meta (include/tsdf_hip.h): what a .vol file carries beyond tsdf_params."""

    _fields_ = [
        ("max_cell_size", C.c_float * 3),
        ("is_empty", C.c_int32),
        ("weight_by_depth", C.c_int32),
        ("weight_by_variance", C.c_int32),
        ("global_transform", C.c_double * 16),
    ]


# tsdf_block_fn / tsdf_header_fn (include/tsdf_hip.h): callbacks of tsdf_hip_save_blocks / tsdf_hip_load_blocks
BLOCK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                       C.POINTER(C.c_float), C.POINTER(C.c_uint8))
HEADER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(TsdfParams), C.POINTER(TsdfVolMeta))


class TsdfHipError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: tsdf_hip error {code} ({detail})")


_lib = None

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/tsdf_hip.h one to one
SIGNATURES = {
    "tsdf_hip_default_params": (None, [C.POINTER(TsdfParams)]),
    "tsdf_hip_create": (C.c_int, [C.POINTER(TsdfParams), C.POINTER(C.c_void_p)]),
    "tsdf_hip_create_multi": (C.c_int, [C.POINTER(TsdfParams), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_void_p)]),
    "tsdf_hip_slab_count": (C.c_int, [C.c_void_p]),
    "tsdf_hip_slab_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32)]),
    "tsdf_hip_load_multi": (C.c_int, [C.c_char_p, C.POINTER(TsdfParams), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_void_p),
                                     C.POINTER(TsdfParams), C.POINTER(TsdfVolMeta)]),
    "tsdf_hip_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "tsdf_hip_host_free": (C.c_int, [C.c_void_p]),
    "tsdf_hip_reset": (C.c_int, [C.c_void_p]),
    "tsdf_hip_destroy": (C.c_int, [C.c_void_p]),
    "tsdf_hip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tsdf_hip_synchronize": (C.c_int, [C.c_void_p]),
    "tsdf_hip_integrate": (C.c_int, [C.c_void_p, _f32p, _u8p, _f32p, _u64p]),
    "tsdf_hip_integrate_async": (C.c_int, [C.c_void_p, _f32p, _u8p, _f32p]),
    "tsdf_hip_integrate_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _f32p, _u64p]),
    "tsdf_hip_organize": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, _u8p, C.c_size_t, C.c_size_t, C.c_float, C.c_int,
                                    _f64p, _f32p, _u8p, _u64p]),
    "tsdf_hip_integrate_staged": (C.c_int, [C.c_void_p, _f32p, _u64p]),
    "tsdf_hip_set_weighting": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "tsdf_hip_frame_begin": (C.c_int, [C.c_void_p, C.POINTER(_f32p), C.POINTER(_u8p)]),
    "tsdf_hip_frame_commit": (C.c_int, [C.c_void_p, _f32p]),
    "tsdf_hip_last_count_detail": (C.c_int, [C.c_void_p, _u64p]),
    "tsdf_hip_last_read_detail": (C.c_int, [C.c_void_p, _u64p]),
    "tsdf_hip_march_timing": (C.c_int, [C.c_void_p, _f32p, _u64p]),
    "tsdf_hip_march_stats": (C.c_int, [C.c_void_p, _u64p]),
    "tsdf_hip_raycast": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int, _f32p]),
    "tsdf_hip_raycast_camera": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int, _f64p, _f32p]),
    "tsdf_hip_raycast_begin": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int, C.c_void_p]),
    "tsdf_hip_raycast_advance": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "tsdf_hip_raycast_advance_list": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "tsdf_hip_render_halo": (C.c_int, [C.POINTER(TsdfParams)]),
    "tsdf_hip_reference_cull_is_noop": (C.c_int, [C.POINTER(TsdfParams)]),
    "tsdf_hip_alloc_probe": (C.c_int, [C.c_void_p, _f32p, C.POINTER(C.c_int32)]),
    "tsdf_hip_sample": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, _f32p, _f32p, _f32p, _u8p]),
    "tsdf_hip_lookup_rgb": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, _u8p, _u8p]),
    "tsdf_hip_march": (C.c_int, [C.c_void_p, C.c_float, C.c_int, _u64p]),
    "tsdf_hip_march_fetch": (C.c_int, [C.c_void_p, _f32p, _u8p, _u64p]),
    "tsdf_hip_march_fetch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsdf_hip_download": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [_f32p, _f32p, _u8p]),
    "tsdf_hip_upload": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [_f32p, _f32p, _u8p]),
    "tsdf_hip_download_color_state": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _f32p]),
    "tsdf_hip_get_planes_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsdf_hip_set_planes_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsdf_hip_device_planes": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tsdf_hip_layout": (C.c_int, [C.c_void_p]),
    "tsdf_hip_centers": (C.c_int, [C.c_void_p, C.c_int, _f32p]),
    "tsdf_hip_save": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(TsdfVolMeta)]),
    "tsdf_hip_save_blocks": (C.c_int, [C.POINTER(TsdfParams), C.POINTER(TsdfVolMeta), C.c_char_p, BLOCK_FN, C.c_void_p]),
    "tsdf_hip_load_blocks": (C.c_int, [C.c_char_p, C.POINTER(TsdfParams), HEADER_FN, BLOCK_FN, C.c_void_p]),
    "tsdf_hip_load": (C.c_int, [C.c_char_p, C.POINTER(TsdfParams), C.POINTER(C.c_void_p), C.POINTER(TsdfParams),
                                C.POINTER(TsdfVolMeta)]),
    "tsdf_hip_error_string": (C.c_char_p, [C.c_int]),
    "tsdf_hip_last_error": (C.c_char_p, []),
    "tsdf_hip_device_count": (C.c_int, []),
    "tsdf_hip_abi_version": (C.c_int, []),
    "tsdf_hip_download_variance_state": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [_f32p, C.POINTER(C.c_int32)]),
    "tsdf_hip_upload_variance_state": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [_f32p, C.POINTER(C.c_int32)]),
    "tsdf_hip_set_reference_cull": (C.c_int, [C.c_void_p, _f32p]),
    "tsdf_hip_integrate_device2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _f32p, _f32p, C.c_void_p, C.c_void_p, _f32p, _f32p,
                                             _u64p, C.POINTER(C.c_int32)]),
    "tsdf_hip_set_frame_pairing": (C.c_int, [C.c_void_p, C.c_int]),
    "tsdf_hip_reference_cull_planes": (C.c_int, [C.POINTER(TsdfParams), _f64p, _f32p]),
    "tsdf_hip_last_launch_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "tsdf_hip_multi_render_stats": (C.c_int, [C.c_void_p, _u64p]),
    "tsdf_hip_multi_link_stats": (C.c_int, [C.c_void_p, _u64p]),
    "tsdf_hip_multi_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "tsdf_hip_multi_kernel_ms": (C.c_int, [C.c_void_p, C.c_int, _f32p, C.POINTER(C.c_int32)]),
}

# include/tsdf_hip_test.h: only libtsdf_hip_test.so (and the A/B variants of tools/build_variant.py) export these
TEST_SIGNATURES = {
    "tsdf_hip_selftest_checksum": (C.c_int, [C.c_void_p, _u64p]),
    "tsdf_hip_selftest_occupancy_mc": (C.c_int, [C.POINTER(C.c_int)]),
    "tsdf_hip_selftest_div_count": (C.c_int, [_f32p, C.POINTER(C.c_uint32), _f32p, _u8p, C.c_size_t]),
    "tsdf_hip_selftest_cvt_pk_u8": (C.c_int, [_f32p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "tsdf_hip_selftest_rgb2lab": (C.c_int, [_u8p, C.c_size_t, _f32p]),
    "tsdf_hip_selftest_lab2rgb": (C.c_int, [_f32p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "tsdf_hip_selftest_struct_oob": (C.c_int, [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.c_int]),
    "tsdf_hip_selftest_div_f32": (C.c_int, [_f32p, _f32p, _f32p, C.c_size_t]),
    "tsdf_hip_selftest_div_f64": (C.c_int, [_f64p, _f64p, _f64p, C.c_size_t]),
    "tsdf_hip_selftest_project": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _u8p]),
    "tsdf_hip_selftest_containing": (C.c_int, [C.c_void_p, _f32p, C.c_size_t, C.POINTER(C.c_int32)]),
    "tsdf_hip_selftest_sweep": (C.c_int, [C.c_void_p, _u64p, _u64p]),
    "tsdf_hip_selftest_read_sweep": (C.c_int, [C.c_void_p, C.c_int, _u64p, _u64p]),
    "tsdf_hip_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "tsdf_hip_selftest_block_flags": (C.c_int, [C.POINTER(TsdfParams), _f32p, C.c_int, C.c_int, _u8p]),
    "tsdf_hip_selftest_row_intervals": (C.c_int, [C.POINTER(TsdfParams), _f32p, _f32p, C.POINTER(C.c_uint32)]),
    "tsdf_hip_selftest_index_box": (C.c_int, [C.POINTER(TsdfParams), _f32p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tsdf_hip_selftest_expf": (C.c_int, [_f32p, C.c_size_t, _f32p]),
}


RAY_RECORD_INTS = 24  # TSDF_HIP_RAY_RECORD_INTS


def _torch_first():
    """A process that uses BOTH this library and PyTorch-ROCm must let torch load its HIP runtime first: torch wheels
    bundle their own copy of the ROCm libraries, libtsdf_hip.so links the system's (/opt/rocm), and when the system copy
    is loaded first `torch.cuda` later finds no device ("No HIP GPUs are available"; seen on MI355X, ROCm 7.2 + torch
    2.10+rocm7.0, tests/test_multi_gpu.py run on its own).  The other order works and is what bench.py and zslab.py
    always did.  So, where torch is installed and not yet imported, import it before the library
    (TSDF_HIP_NO_TORCH_PRELOAD=1 skips this; a process without torch is not affected)."""
    import sys
    if "torch" in sys.modules or os.environ.get("TSDF_HIP_NO_TORCH_PRELOAD") == "1":
        return
    try:  # only a ROCm build of torch carries a HIP runtime of its own: read its version from the metadata, without importing it
        from importlib import metadata
        if "rocm" not in metadata.version("torch"):
            return
    except Exception:  # no torch, or no metadata: nothing to order
        return
    try:
        import torch  # noqa: F401
        if os.environ.get("TSDF_HIP_VERBOSE"):
            print("cpu_tsdf_amd.capi: imported torch before libtsdf_hip.so (its bundled ROCm runtime must load first)", file=sys.stderr)
    except Exception:  # a broken torch installation must not keep the library from loading
        pass


def use_test_library():
    """Bind libtsdf_hip_test.so (the product's sources + the hooks of include/tsdf_hip_test.h) instead of the product.
    Must be called before the first load(); TSDF_HIP_LIB_PATH, if set, still wins."""
    global LIB_PATH
    if _lib is not None:
        if LIB_PATH != TEST_LIB_PATH and not os.environ.get("TSDF_HIP_LIB_PATH"):
            raise RuntimeError("capi.use_test_library() after the product library was loaded")
        return
    if not os.environ.get("TSDF_HIP_LIB_PATH"):
        LIB_PATH = TEST_LIB_PATH


def has_test_hooks():
    load()
    return _has_hooks


_has_hooks = False


def load():
    """Load the library (once) and declare every entry point.  Raises if it is not built."""
    global _lib, _has_hooks
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is not built and there is no fallback path. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc).")
    _torch_first()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _has_hooks = hasattr(lib, "tsdf_hip_set_tuning")
    if _has_hooks:
        for name, (res, args) in TEST_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def check(code, where):
    if code != OK:
        lib = load()
        detail = lib.tsdf_hip_error_string(code).decode() + ": " + lib.tsdf_hip_last_error().decode()
        raise TsdfHipError(code, where, detail)


def set_tuning(name, value):
    """Launch-shape knobs at run time: a hook of the TEST library (the product reads the TSDF_HIP_* variables once).  The
    one knob the product follows live is vol_chunk, through the environment -- so that is how it is set here, for whichever
    library (and for the C++ drop-in, which links the product) is in the process."""
    if name == "vol_chunk":
        os.environ["TSDF_HIP_VOL_CHUNK"] = str(int(value))
        return
    lib = load()
    if not _has_hooks:
        raise RuntimeError(f"set_tuning({name}): {LIB_PATH} has no test hooks; call capi.use_test_library() before the first load")
    check(lib.tsdf_hip_set_tuning(name.encode(), int(value)), f"set_tuning({name})")


def default_params():
    p = TsdfParams()
    load().tsdf_hip_default_params(C.byref(p))
    return p


def as_f32p(a):
    return a.ctypes.data_as(_f32p)


def as_u8p(a):
    return a.ctypes.data_as(_u8p)


def f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class PinnedArray:
    """A numpy array over pinned host memory (tsdf_hip_host_alloc): transfers into / out of it skip the library's
    bounce buffer.  Keep the object alive as long as `.array` (or views of it) are in use."""

    def __init__(self, shape, dtype=np.float32):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        check(load().tsdf_hip_host_alloc(self.nbytes, C.byref(p)), "host_alloc")
        self._p = p
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(p.value), dtype=dtype).reshape(shape)

    def free(self):
        if self._p:
            self.array = None
            load().tsdf_hip_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
