"""Deterministic synthetic inputs: analytic depth + colour frames and camera
poses, shared by the tests, ``bench.py`` and the CPU baseline.  numpy only, no RNG unless asked.

Conventions follow the reference CLI: intrinsics ``fx = fy = 525*W/640, cx = W/2 - 0.5,
cy = H/2 - 0.5`` (src/prog/integrate.cpp:350-353); a pose is the 4x4 double ``trans`` handed to
``integrateCloud`` (camera -> volume); camera axes x right, y down, z forward; depth is the
camera-frame z of the first hit; pixels with no return are NaN (integrate.cpp:574-582).
"""
import math

import numpy as np


def cli_intrinsics(width, height):
    """src/prog/integrate.cpp:350-353."""
    f = 525.0 * width / 640.0
    return f, f, width / 2.0 - 0.5, height / 2.0 - 0.5


def eigen_affine_inverse(m):
    """``Eigen::Affine3d::inverse()`` restated operation by operation [Eigen-recall, 3.3]:
    3x3 inverse by cofactors times 1/det (``compute_inverse_size3_helper``), translation
    ``-(Rinv) * t`` with row sums ``(a + b) + c``.  ``m`` is 4x4 (or 3x4) float64; returns 4x4."""
    a = [[float(m[r][c]) for c in range(3)] for r in range(3)]
    t = [float(m[r][3]) for r in range(3)]

    def cof(i, j):
        i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
        return a[i1][j1] * a[i2][j2] - a[i1][j2] * a[i2][j1]

    c0 = [cof(0, 0), cof(1, 0), cof(2, 0)]
    det = c0[0] * a[0][0] + (c0[1] * a[1][0] + c0[2] * a[2][0])
    invdet = 1.0 / det
    inv = [[0.0] * 3 for _ in range(3)]
    for c in range(3):
        inv[0][c] = c0[c] * invdet
    inv[1][0] = cof(0, 1) * invdet
    inv[1][1] = cof(1, 1) * invdet
    inv[2][2] = cof(2, 2) * invdet
    inv[1][2] = cof(2, 1) * invdet
    inv[2][1] = cof(1, 2) * invdet
    inv[2][0] = cof(0, 2) * invdet
    out = np.eye(4, dtype=np.float64)
    for r in range(3):
        for c in range(3):
            out[r, c] = inv[r][c]
        out[r, 3] = ((-inv[r][0]) * t[0] + (-inv[r][1]) * t[1]) + (-inv[r][2]) * t[2]
    return out


def cam_from_vol_f32(trans):
    """``trans.inverse().cast<float>()`` (include/cpu_tsdf/impl/tsdf_volume_octree.hpp:54) as the 12
    floats the C ABI takes (3x4 row-major)."""
    return np.ascontiguousarray(eigen_affine_inverse(trans)[:3, :4].astype(np.float32).reshape(12))


def look_at_pose(eye, target=(0.0, 0.0, 0.0), down=(0.0, 1.0, 0.0)):
    """Camera -> volume pose: z forward (towards target), y along `down`, x = y cross z."""
    eye = np.asarray(eye, dtype=np.float64)
    z = np.asarray(target, dtype=np.float64) - eye
    z /= np.linalg.norm(z)
    y = np.asarray(down, dtype=np.float64)
    y = y - z * np.dot(y, z)
    y /= np.linalg.norm(y)
    x = np.cross(y, z)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, eye
    return m


def turntable_pose(i, n_frames, size, radius_factor=2.2, tilt=0.0):
    """Scene A camera i of F: circle of radius 2.2*S in the XZ plane, looking at the origin."""
    th = 2.0 * math.pi * i / n_frames
    r = radius_factor * size
    eye = (r * math.sin(th), -tilt * size, -r * math.cos(th))
    return look_at_pose(eye)


class Scene:
    """Sphere (radius 0.25*S at the origin) inside a box of half-extent 0.47*S of which only the far
    interior faces are visible: every ray that enters the box returns a depth."""

    def __init__(self, size, width=640, height=480, sphere=0.25, box=0.47):
        self.size = float(size)
        self.width, self.height = int(width), int(height)
        self.fx, self.fy, self.cx, self.cy = cli_intrinsics(width, height)
        self.r = sphere * self.size
        self.h = box * self.size

    def rays(self, trans):
        u = np.arange(self.width, dtype=np.float64)
        v = np.arange(self.height, dtype=np.float64)
        uu, vv = np.meshgrid(u, v)
        dc = np.stack([(uu - self.cx) / self.fx, (vv - self.cy) / self.fy, np.ones_like(uu)], -1)
        dw = dc @ trans[:3, :3].T  # unnormalised: ray parameter == camera-frame z
        return trans[:3, 3].copy(), dw

    def depth(self, trans, noise_seed=None, noise_sigma=0.001):
        """float32 (H, W) depth image seen from pose `trans`; NaN where the ray misses the box."""
        o, d = self.rays(trans)
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (-self.h - o) / d
            t2 = (self.h - o) / d
            t_near = np.minimum(t1, t2).max(-1)
            t_far = np.maximum(t1, t2).min(-1)
            hit_box = (t_far >= t_near) & (t_far > 0)
            # sphere: |o + t d|^2 = r^2, nearest positive root
            a = (d * d).sum(-1)
            b = 2.0 * (d * o).sum(-1)
            c = float(o @ o) - self.r * self.r
            disc = b * b - 4 * a * c
            ts = (-b - np.sqrt(np.where(disc >= 0, disc, np.nan))) / (2 * a)
        t = np.where(hit_box, t_far, np.nan)
        sph = np.isfinite(ts) & (ts > 0) & hit_box
        t = np.where(sph, ts, t)
        if noise_seed is not None:
            rng = np.random.RandomState(noise_seed)
            t = t + rng.normal(0.0, noise_sigma, t.shape)
        return np.ascontiguousarray(t.astype(np.float32))

    def bgra(self, i):
        """PCL PointXYZRGBA byte order (b, g, r, a): r = u & 255, g = v & 255, b = i & 255, a = 255."""
        u = np.arange(self.width, dtype=np.uint32)[None, :]
        v = np.arange(self.height, dtype=np.uint32)[:, None]
        img = np.empty((self.height, self.width, 4), dtype=np.uint8)
        img[..., 0] = i & 255
        img[..., 1] = (v & 255) + 0 * u
        img[..., 2] = (u & 255) + 0 * v
        img[..., 3] = 255
        return img


def scene_a(res, width=640, height=480, voxel=2.0 ** -8):
    """Scene A "turntable" for a cubic grid of `res` voxels of 2^-8 m (all centres exact in fp32)."""
    size = res * voxel
    return Scene(size, width, height)


def scene_b_pose(i, n_frames):
    """Scene B "README-realistic": camera inside a 10 m volume, 1.5 m from the sphere surface region,
    sweeping a small arc; sensor range 0..3 m sees ~1 % of the voxels."""
    th = 0.6 * (i / max(1, n_frames - 1) - 0.5)
    eye = (2.5 * math.sin(th), 0.1, -2.5 * math.cos(th))
    return look_at_pose(eye)


def scene_b(width=640, height=480):
    s = Scene(10.0, width, height, sphere=0.1, box=0.35)
    return s
