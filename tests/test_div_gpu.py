"""GPU tier: the shared-reciprocal dividers used inside the integrate kernel (csrc/tsdf_div.h) are
bit-identical to IEEE division (numpy on the host) over the operand ranges the kernel feeds them and
over adversarial values (zeros, denormals, huge, inf, NaN) where they must fall back to `/`."""
import ctypes as C

import numpy as np
import pytest

from cpu_tsdf_amd import capi

pytestmark = pytest.mark.gpu


def gpu_div32(a, b):
    out = np.empty_like(a)
    capi.check(capi.load().tsdf_hip_selftest_div_f32(capi.as_f32p(a), capi.as_f32p(b), capi.as_f32p(out), a.size), "div32")
    return out


def gpu_div64(a, b):
    out = np.empty_like(a)
    p = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    capi.check(capi.load().tsdf_hip_selftest_div_f64(p(a), p(b), p(out), a.size), "div64")
    return out


def same_bits(x, y):
    return (x.view(np.uint32 if x.dtype == np.float32 else np.uint64) ==
            y.view(np.uint32 if y.dtype == np.float32 else np.uint64)) | (np.isnan(x) & np.isnan(y))


def test_div32_kernel_ranges(gpu):
    rng = np.random.RandomState(1)
    n = 1 << 22
    # d update: (d*w + dn) / (w + 1), colour: (w*c + cn) / (w + 1), normalisation: raw / neg
    w = rng.randint(0, 101, n).astype(np.float32)
    num = np.concatenate([(rng.uniform(-1, 1, n // 2) * (w[: n // 2] + 1)).astype(np.float32),
                          (rng.randint(0, 256, n // 2) * (w[n // 2:] + 1) - rng.randint(0, 255, n // 2)).astype(np.float32)])
    den = (w + 1).astype(np.float32)
    with np.errstate(all="ignore"):
        want = num / den
    assert same_bits(gpu_div32(num, den), want).all()
    raw = rng.uniform(-0.05, 0.05, n).astype(np.float32)
    neg = np.full(n, 0.03, np.float32)
    assert same_bits(gpu_div32(raw, neg), raw / neg).all()


def test_div32_random_and_adversarial(gpu):
    rng = np.random.RandomState(2)
    n = 1 << 22
    a = rng.randint(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    b = rng.randint(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, 1e-38, 3e38, 2.0 ** -40, 2.0 ** 40,
                        float.fromhex('0x1.fffffep-41'), float.fromhex('0x1.000002p40'), 1.0000001, 0.99999994], np.float32)
    a[: special.size ** 2] = np.repeat(special, special.size)
    b[: special.size ** 2] = np.tile(special, special.size)
    with np.errstate(all="ignore"):
        want = a / b
    got = gpu_div32(a, b)
    ok = same_bits(got, want)
    assert ok.all(), (a[~ok][:5], b[~ok][:5], got[~ok][:5], want[~ok][:5])


def test_div64_projection_ranges(gpu):
    rng = np.random.RandomState(3)
    n = 1 << 22
    gz = rng.uniform(1e-3, 200.0, n).astype(np.float32).astype(np.float64)
    gx = (rng.uniform(-1, 1, n) * gz * 2).astype(np.float32).astype(np.float64)
    fx = 525.0 * rng.choice([1.0, 2.0, 1.0 / 3.0, 0.977], n)
    num = gx * fx
    want = num / gz
    assert same_bits(gpu_div64(num, gz), want).all()
    # wide float-sized exponents, exact-integer quotients (pixel boundaries), tiny / huge numerators
    gz = np.exp2(rng.uniform(-120, 120, n)).astype(np.float32).astype(np.float64)
    num = np.exp2(rng.uniform(-140, 140, n)) * rng.choice([-1.0, 1.0], n)
    num[: n // 4] = gz[: n // 4] * rng.randint(-2000, 2000, n // 4)
    num[n // 4: n // 4 + 8] = [0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, 1.7e308, -1.7e308]
    with np.errstate(all="ignore"):
        want = num / gz
    got = gpu_div64(num, gz)
    ok = same_bits(got, want) | (got == want)  # the sign of a zero quotient is irrelevant before `+ cx`
    # non-finite numerators never reach an accepted pixel; both sides just have to stay non-finite
    nonfin = ~np.isfinite(num)
    assert (ok | (nonfin & ~np.isfinite(got))).all(), (num[~ok][:5], gz[~ok][:5], got[~ok][:5], want[~ok][:5])


def _project_reference(g, fx, fy, cx, cy, W, H):
    """reprojectPoint in IEEE double exactly as the C++ evaluates it (tsdf_volume_octree.cpp:611-617)."""
    gx, gy, gz = (g[:, k].astype(np.float64) for k in range(3))
    with np.errstate(all="ignore"):
        ru = gx * fx / gz + cx
        rv = gy * fy / gz + cy

    def cvtt(v):
        ok = (v > -2147483649.0) & (v < 2147483648.0)
        return np.where(ok, np.trunc(np.where(ok, v, 0.0)), -2147483648.0).astype(np.int64)

    u, v = cvtt(ru), cvtt(rv)
    inside = (u >= 0) & (u < W) & (v >= 0) & (v < H)
    return np.where(inside, v * W + u, -1).astype(np.int32), ru, rv


@pytest.mark.parametrize("W,H", [(640, 480), (1280, 960), (160, 120)])
def test_certified_fp32_projection_matches_double(gpu, W, H):
    """The fp32 projection + ambiguity band + fp64 fallback must give the reference's pixel for every
    point, including points engineered to sit within a few ulps of pixel boundaries."""
    from cpu_tsdf_amd import synth
    from tests.common import make_volume
    vol, sc = make_volume(64, width=W, height=H)
    vol.reset()
    fx, fy, cx, cy = vol.getCameraIntrinsics()
    rng = np.random.RandomState(7)
    n = 1 << 22
    gz = rng.uniform(0.05, 60.0, n)
    u_t = rng.uniform(-3, W + 3, n)
    v_t = rng.uniform(-3, H + 3, n)
    # a quarter of the points sit (almost) exactly on integer pixel boundaries in u or v
    k = n // 4
    u_t[:k] = np.round(u_t[:k]) + rng.choice([0.0, 1e-7, -1e-7, 3e-5, -3e-5, 2e-4, -2e-4], k)
    v_t[k:2 * k] = np.round(v_t[k:2 * k]) + rng.choice([0.0, 1e-7, -1e-7, 3e-5, -3e-5], k)
    g = np.stack([(u_t - cx) / fx * gz, (v_t - cy) / fy * gz, gz], 1).astype(np.float32)
    g[:8, 2] = [1e-30, 1e-38, 1e-44, 3e38, 1e-20, 1.0, 1.0, 1.0]
    g[5, 0] = 0.0
    g[6, 0] = 3e38
    g[7, 1] = -3e38
    want, ru, rv = _project_reference(g, fx, fy, cx, cy, W, H)
    pf = np.empty(n, np.int32)
    pe = np.empty(n, np.int32)
    amb = np.empty(n, np.uint8)
    i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    capi.check(capi.load().tsdf_hip_selftest_project(vol._need(), capi.as_f32p(np.ascontiguousarray(g)), n, i32(pf),
                                                     i32(pe), capi.as_u8p(amb)), "selftest_project")
    assert np.array_equal(pe, want), f"exact path: {(pe != want).sum()} mismatches"
    bad = pf != want
    assert not bad.any(), (int(bad.sum()), g[bad][:4], pf[bad][:4], want[bad][:4], ru[bad][:4], rv[bad][:4])
    frac = amb.mean()
    assert 0.0 < frac < 0.55  # boundary-engineered points (half of them) are flagged, random ones almost never
    rnd = amb[2 * k + 8:]
    assert rnd.mean() < 0.01


@pytest.mark.parametrize("seed", [0, 1])
def test_count_divider_equals_ieee_division(gpu, seed):
    """The PACKED layout's weighted-mean divider: numerator / integer count k in [1, 256] through the table
    reciprocal and the scale-free ladder, guarded on the RESULT -- a normal quotient stands, anything else (zero,
    subnormal, overflow, NaN) is redone with IEEE division.  Numerators from every binade, every k, plus the corners
    where the ladder could go wrong if the claim were false (quotients around the smallest normal, the largest finite
    numerator, exact multiples and their neighbours)."""
    rng = np.random.RandomState(seed)
    n = 1 << 22
    k = rng.randint(1, 257, n).astype(np.uint32)
    a = (rng.uniform(1.0, 2.0, n) * np.exp2(rng.randint(-149, 128, n).astype(np.float64)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    m = n // 8
    a[:m] = rng.randint(0, 1 << 20, m).astype(np.float32) * k[:m]                          # exact quotients
    a[m:2 * m] = np.nextafter((rng.randint(1, 1 << 20, m).astype(np.float32) * k[m:2 * m]).astype(np.float32),
                              np.float32(np.inf) * rng.choice([-1.0, 1.0], m).astype(np.float32))  # their neighbours
    tiny = np.float32(2.0 ** -126)
    a[2 * m:3 * m] = (tiny * k[2 * m:3 * m] * rng.uniform(0.5, 2.0, m)).astype(np.float32)  # quotients around the smallest normal
    a[3 * m:3 * m + 6] = [0.0, -0.0, np.inf, -np.inf, np.nan, np.finfo(np.float32).max]
    # the typical case: d * w + dn with d in [-1, 1], w = k - 1, dn in [-1, 1]
    a[4 * m:5 * m] = (rng.uniform(-1, 1, m).astype(np.float32) * (k[4 * m:5 * m] - 1).astype(np.float32) +
                      rng.uniform(-1, 1, m).astype(np.float32)).astype(np.float32)
    out = np.empty(n, np.float32)
    fast = np.empty(n, np.uint8)
    capi.check(capi.load().tsdf_hip_selftest_div_count(capi.as_f32p(a), k.ctypes.data_as(C.POINTER(C.c_uint32)), capi.as_f32p(out),
                                                       capi.as_u8p(fast), n), "selftest_div_count")
    with np.errstate(all="ignore"):
        want = a / k.astype(np.float32)
    same = (out.view(np.uint32) == want.view(np.uint32)) | (np.isnan(out) & np.isnan(want))
    assert same.all(), (int((~same).sum()), a[~same][:5], k[~same][:5], out[~same][:5], want[~same][:5])
    assert fast[4 * m:5 * m].mean() > 0.999          # the kernel's own numerators take the ladder
    assert fast[np.abs(want) < tiny].sum() == 0       # subnormal and zero quotients never do


def test_cvt_pk_u8_rounds_to_nearest_even_and_saturates(gpu):
    """v_cvt_pk_u8_f32, the one-instruction convert-and-pack the colour update can be built with (TSDF_COLOR_PK, an A/B
    switch that is off: measured slower than convert + shift-or): it rounds to nearest even, saturates to [0, 255],
    turns NaN into 0 and leaves the other three bytes alone.  The shipped path truncates with v_cvt_u32_f32."""
    rng = np.random.RandomState(5)
    x = np.concatenate([np.float32([0.25, 0.5, 0.75, 1.5, 2.5, 3.5, 254.5, 255.4, 255.5, 256.7, 1e9, -0.25, -0.5, -3.0,
                                    np.nan, np.inf, -np.inf, 0.49999997, 1.4999999]),
                        rng.uniform(-2, 258, 1 << 16).astype(np.float32),
                        (rng.randint(0, 256, 1 << 12) + 0.5).astype(np.float32)])
    out = np.empty(len(x), dtype=np.uint32)
    capi.check(capi.load().tsdf_hip_selftest_cvt_pk_u8(capi.as_f32p(x), len(x), out.ctypes.data_as(C.POINTER(C.c_uint32))),
               "cvt_pk_u8")
    assert ((out & 0xffff00ff) == 0xAABB00DD).all()
    want = np.where(np.isnan(x), 0, np.clip(np.rint(np.nan_to_num(x.astype(np.float64), nan=0.0, posinf=1e9, neginf=-1e9)),
                                            0, 255)).astype(np.uint32)
    assert np.array_equal((out >> 8) & 255, want)
