// libtsdf_hip.so -- integrateCloud: one thread per voxel quad, project-and-weighted-average.
//
// Replaces TSDFVolumeOctree::integrateCloud / updateVoxel
// (include/cpu_tsdf/impl/tsdf_volume_octree.hpp:48-103, :113-218) on a flat SoA grid.
// The arithmetic below follows the reference line by line in fp32 with NO fused multiply-add
// (the reference's CMake build has no -march flag, so x86-64 emits separate mul/add) and IEEE
// division; this file is compiled with -ffp-contract=off.  What has no dense counterpart (octree
// split hpp:161-187, prune :122-142, return codes :209-214, the surface pre-split pass :56-90,
// coarse frustum cull :93-94) is dropped: on a dense grid every voxel is a finest leaf and the
// per-voxel tests of updateVoxel imply the cull.
//
// Memory behaviour: a thread owns 4 x-consecutive voxels (16 B of d, 16 B of w); a wave touches
// 1 KiB contiguous per plane.  d/w(/rgb) are read only if at least one of the four voxels reaches
// addObservation, so algorithmic traffic is 16 B (24 B colour) per observed voxel plus the depth
// gather, which is served by L2 (the 640x480 frame is 1.2 MB).  HBM-bound, no MFMA.
#include <limits.h>
#include <math.h>

#include <cmath>

#include "tsdf_common.h"
#include "tsdf_div.h"

struct IntegrateArgs {
  float m[12];       // cam_from_vol, row-major 3x4
  double fx, fy, cx, cy;
  float zmin, zmax;  // min/max_sensor_dist_
  float pos, neg;    // max_dist_pos_/neg_
  float wmax;        // max_weight_
  float pos_over_neg; // max_dist_pos_ / max_dist_neg_ (IEEE fp32, host)
  int W, H;
  int nx, ny;
  int qpr;           // quads per row = ceil(nx/4)
  int rows;          // ny * planes to integrate
  int z_global0;     // global z of the first integrated plane
  int zl0;           // allocated-plane index of the first integrated plane
  int log2TX, TX, TY;
  unsigned xchunks;  // ceil(qpr / TX)
  unsigned n_tiles;
  int64_t pitch;
};

// Per-voxel state carried between the pipeline stages.
struct Obs {
  int pix;     // v*W + u, or -1 if the voxel fails hpp:146 / reprojectPoint
  float gz;    // camera-frame z of the voxel centre
  float z;     // gathered depth
  uint32_t c;  // gathered colour (PCL b,g,r,a bytes)
};

// reprojectPoint (tsdf_volume_octree.cpp:611-617) for a voxel that already passed the range test
// (so g.z > 0 and finite): u = (int)(x*fx/z + cx) evaluated in double, truncation toward zero.
// u and v divide by the same g.z, so the fp64 reciprocal is refined once (tsdf_div.h).
// v_cvt_i32_f64 saturates where x86's cvttsd2si returns INT_MIN; both land outside [0, W), and NaN
// cannot occur here (the host rejects non-finite / absurd poses before launching).
static __device__ __forceinline__ int project(const IntegrateArgs &a, float gx, float gy, float gz) {
  const Rcp64 rz = rcp64_prepare((double)gz);
  const int u = (int)(div64((double)gx * a.fx, rz) + a.cx);
  const int v = (int)(div64((double)gy * a.fy, rz) + a.cy);
  const bool in = (unsigned)u < (unsigned)a.W && (unsigned)v < (unsigned)a.H;
  return in ? v * a.W + u : -1;
}

// Is |v| inside the exponent window where the scale/fixup-free divider is exact (tsdf_div.h)?
static __device__ __forceinline__ bool in_window(float v) {
  const uint32_t b = __float_as_uint(v) & 0x7fffffffu;
  return b - 0x2b800000u <= 0x53800000u - 0x2b800000u;  // 2^-40 <= |v| <= 2^40
}

// Branch-free divider for operands already known to be in the window (or a == +0).
static __device__ __forceinline__ float div32_fast(float a, const Rcp32 &r) {
  const float q0 = a * r.y;
  const float r0 = __builtin_fmaf(r.nb, q0, a);
  const float q1 = __builtin_fmaf(r0, r.y, q0);
  const float r1 = __builtin_fmaf(r.nb, q1, a);
  return __builtin_fmaf(r1, r.y, q1);
}

static __device__ __forceinline__ uint32_t unpack_rgb_new(uint32_t bgra, int ch) {  // ch 0:r 1:g 2:b
  return (bgra >> (16 - 8 * ch)) & 255u;
}

// OctreeNode::addObservation with w_new = 1 (octree.cpp:152-163; both weightings of hpp:200-204 are
// unreachable: no setter for weight_by_depth_/weight_by_variance_), and RGBNode::addObservation
// (octree.cpp:328-337): per channel (uint8)((w*c + w_new*c_new)/(w+w_new)) with the OLD w, truncating.
// All four quotients share the divisor w + 1.  FAST = scale-free shared-reciprocal ladder, valid when
// update_is_safe(); otherwise the compiler's full IEEE division.
template <bool COLOR, bool FAST>
static __device__ __forceinline__ void add_observation(float &d, float &w, uint32_t &rgb, float dn,
                                                       uint32_t bgra, float wmax) {
  const float wn = 1.f;
  const float wsum = w + wn;
  Rcp32 rs;
  if (FAST) rs = rcp32_prepare(wsum);
  if (COLOR) {
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float num = w * (float)((rgb >> (8 * ch)) & 255u) + wn * (float)unpack_rgb_new(bgra, ch);
      const float q = FAST ? div32_fast(num, rs) : num / wsum;
      out |= ((uint32_t)(uint8_t)q) << (8 * ch);
    }
    rgb = out;
  }
  const float num = d * w + dn * wn;
  d = FAST ? div32_fast(num, rs) : num / wsum;
  w = wsum;
  if (w > wmax) w = wmax;
}

// Conditions under which every quotient of add_observation<FAST> equals IEEE division bit for bit:
// w == +0 or 2^-20 <= w <= 2^30 (then w+1 and every colour numerator w*c + c_new, c in 0..255, is +0
// or inside the window) and the distance numerator d*w + dn is +0 or inside the window.
static __device__ __forceinline__ bool update_is_safe(float d, float w, float dn) {
  const uint32_t wb = __float_as_uint(w);
  const bool w_ok = wb == 0u || (wb - 0x35800000u <= 0x4e800000u - 0x35800000u);
  const float num = d * w + dn;
  const bool n_ok = __float_as_uint(num) == 0u || in_window(num);
  return w_ok && n_ok;
}

struct Tile {
  int x4;       // first voxel x of this thread's quad, -1 if the thread has no quad in this tile
  int64_t idx;  // element index of that voxel in the SoA planes
};

static __device__ __forceinline__ Tile locate(const IntegrateArgs &a, unsigned t, unsigned tx, unsigned ty,
                                               unsigned &y, unsigned &zl) {
  Tile tl;
  tl.x4 = -1;
  tl.idx = 0;
  const unsigned rg = t / a.xchunks;
  const unsigned xc = t - rg * a.xchunks;
  const unsigned row = rg * (unsigned)a.TY + ty;
  const unsigned xq = xc * (unsigned)a.TX + tx;
  if (row >= (unsigned)a.rows || xq >= (unsigned)a.qpr) return tl;
  zl = row / (unsigned)a.ny;
  y = row - zl * (unsigned)a.ny;
  tl.x4 = (int)xq * 4;
  tl.idx = ((int64_t)(a.zl0 + (int)zl) * a.ny + y) * a.pitch + tl.x4;
  return tl;
}

// Stage 1 for one quad: transform the four centres (pcl::transformPoint, hpp:145), range-test
// (hpp:146, .cpp:616), project, and issue the depth (+colour) gathers.
template <int ORDER, bool COLOR>
static __device__ __forceinline__ void stage_project(const IntegrateArgs &a, const Tile &tl, unsigned y,
                                                     unsigned zl, const float *__restrict__ depth,
                                                     const uint32_t *__restrict__ bgra,
                                                     const float *__restrict__ ctrx,
                                                     const float *__restrict__ ctry,
                                                     const float *__restrict__ ctrz, Obs obs[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    obs[j].pix = -1;
    obs[j].gz = 0.f;
    obs[j].z = 0.f;
    obs[j].c = 0u;
  }
  if (tl.x4 < 0) return;
  const float cy = ctry[y];
  const float cz = ctrz[a.z_global0 + (int)zl];
  const float4 cx4 = *reinterpret_cast<const float4 *>(ctrx + tl.x4);
  const float cxs[4] = {cx4.x, cx4.y, cx4.z, cx4.w};
  float s[3], p1[3], p2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (ORDER == TSDF_XFORM_PCL_SSE) {
      s[r] = cy * a.m[4 * r + 1] + (cz * a.m[4 * r + 2] + a.m[4 * r + 3]);
    } else {
      p1[r] = a.m[4 * r + 1] * cy;
      p2[r] = a.m[4 * r + 2] * cz;
    }
  }
  float g[4][3];
  bool in[4], any = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (ORDER == TSDF_XFORM_PCL_SSE)
        g[j][r] = cxs[j] * a.m[4 * r] + s[r];
      else
        g[j][r] = ((a.m[4 * r] * cxs[j] + p1[r]) + p2[r]) + a.m[4 * r + 3];
    }
    // hpp:146  if (v_g.z < min_sensor_dist_ || v_g.z > max_sensor_dist_) return 0;  .cpp:616  pt.z > 0
    in[j] = !(g[j][2] < a.zmin || g[j][2] > a.zmax) && g[j][2] > 0.f && (tl.x4 + j < a.nx);
    any |= in[j];
  }
  if (!any) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pix = project(a, g[j][0], g[j][1], in[j] ? g[j][2] : 1.f);
    obs[j].pix = in[j] ? pix : -1;
    obs[j].gz = g[j][2];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (obs[j].pix >= 0) {
      obs[j].z = depth[obs[j].pix];
      if (COLOR) obs[j].c = bgra[obs[j].pix];
    }
}

// PIPE = software-pipelined tile loop: while the d/w/rgb loads of tile t are in flight the thread
// projects tile t+1 and issues its depth gathers.  SKIP = do not write back planes whose four values
// did not change (free space: d stays at the hinge value; after weight saturation nothing changes).
template <int ORDER, bool COLOR, bool PIPE, bool SKIP>
static __global__ void __launch_bounds__(256)
k_integrate(const IntegrateArgs a, float *__restrict__ D, float *__restrict__ Wt, uint32_t *__restrict__ RGB,
            const float *__restrict__ depth, const uint32_t *__restrict__ bgra,
            const float *__restrict__ ctrx, const float *__restrict__ ctry, const float *__restrict__ ctrz,
            unsigned long long *__restrict__ n_obs) {
  const unsigned tid = threadIdx.x;
  const unsigned tx = tid & (unsigned)(a.TX - 1);
  const unsigned ty = tid >> a.log2TX;
  const Rcp32 rneg = rcp32_prepare(a.neg);
  unsigned cnt = 0;

  unsigned y = 0, zl = 0;
  Obs nxt[4];
  Tile tl_n;
  unsigned t = blockIdx.x;
  if (PIPE && t < a.n_tiles) {
    tl_n = locate(a, t, tx, ty, y, zl);
    stage_project<ORDER, COLOR>(a, tl_n, y, zl, depth, bgra, ctrx, ctry, ctrz, nxt);
  }
  for (; t < a.n_tiles; t += gridDim.x) {
    Obs cur[4];
    Tile tl;
    if (PIPE) {
      tl = tl_n;
#pragma unroll
      for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    } else {
      tl = locate(a, t, tx, ty, y, zl);
      stage_project<ORDER, COLOR>(a, tl, y, zl, depth, bgra, ctrx, ctry, ctrz, cur);
    }
    // stage 2 (hpp:152-198): NaN test, projective SDF, hinge, normalisation
    float dn[4];
    bool act[4], any = false, band_safe = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float raw = cur[j].z - cur[j].gz;                       // hpp:159
      act[j] = cur[j].pix >= 0 && !isnan(cur[j].z) && !(raw < -a.neg);  // hpp:152, :193-196
      const bool clamped = raw > a.pos;                              // hpp:189-192
      dn[j] = clamped ? a.pos_over_neg : div32_fast(raw, rneg);      // hpp:198
      band_safe &= !act[j] || clamped || __float_as_uint(raw) == 0u || in_window(raw);
      any |= act[j];
    }
    if (!band_safe) {  // operands outside the scale-free window: redo with the compiler's IEEE division
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (act[j] && !(cur[j].z - cur[j].gz > a.pos)) dn[j] = (cur[j].z - cur[j].gz) / a.neg;
    }
    // stage 3: issue the read half of the read-modify-write
    float4 d4 = make_float4(0, 0, 0, 0), w4 = d4;
    uint4 c4 = make_uint4(0, 0, 0, 0);
    if (any) {
      d4 = *reinterpret_cast<const float4 *>(D + tl.idx);
      w4 = *reinterpret_cast<const float4 *>(Wt + tl.idx);
      if (COLOR) c4 = *reinterpret_cast<const uint4 *>(RGB + tl.idx);
    }
    if (PIPE) {  // stage 1 of the NEXT tile, overlapping the loads above
      const unsigned tn = t + gridDim.x;
      if (tn < a.n_tiles) {
        tl_n = locate(a, tn, tx, ty, y, zl);
        stage_project<ORDER, COLOR>(a, tl_n, y, zl, depth, bgra, ctrx, ctry, ctrz, nxt);
      }
    }
    // stage 4: running average and write-back
    if (any) {
      const float d0[4] = {d4.x, d4.y, d4.z, d4.w};
      const float w0[4] = {w4.x, w4.y, w4.z, w4.w};
      const uint32_t c0[4] = {c4.x, c4.y, c4.z, c4.w};
      float dv[4], wv[4];
      uint32_t cv[4];
      bool safe = true;
#pragma unroll
      for (int j = 0; j < 4; ++j) safe &= !act[j] || update_is_safe(d0[j], w0[j], dn[j]);
      if (safe) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dv[j] = d0[j];
          wv[j] = w0[j];
          cv[j] = c0[j];
          add_observation<COLOR, true>(dv[j], wv[j], cv[j], dn[j], cur[j].c, a.wmax);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dv[j] = d0[j];
          wv[j] = w0[j];
          cv[j] = c0[j];
          add_observation<COLOR, false>(dv[j], wv[j], cv[j], dn[j], cur[j].c, a.wmax);
        }
      }
      bool chg_d = false, chg_w = false, chg_c = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dv[j] = act[j] ? dv[j] : d0[j];
        wv[j] = act[j] ? wv[j] : w0[j];
        cv[j] = act[j] ? cv[j] : c0[j];
        chg_d |= __float_as_uint(dv[j]) != __float_as_uint(d0[j]);
        chg_w |= __float_as_uint(wv[j]) != __float_as_uint(w0[j]);
        chg_c |= cv[j] != c0[j];
        cnt += act[j] ? 1u : 0u;
      }
      if (!SKIP || chg_d) *reinterpret_cast<float4 *>(D + tl.idx) = make_float4(dv[0], dv[1], dv[2], dv[3]);
      if (!SKIP || chg_w) *reinterpret_cast<float4 *>(Wt + tl.idx) = make_float4(wv[0], wv[1], wv[2], wv[3]);
      if (COLOR && (!SKIP || chg_c))
        *reinterpret_cast<uint4 *>(RGB + tl.idx) = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    }
  }

  // one atomic per block
  __shared__ unsigned s_cnt;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  if (cnt) atomicAdd(&s_cnt, cnt);
  __syncthreads();
  if (tid == 0 && s_cnt) atomicAdd(n_obs, (unsigned long long)s_cnt);
}

static int launch_integrate(tsdf_handle h, const float *d_depth, const uint32_t *d_bgra, const float T[12],
                            uint64_t *n_observed) {
  const tsdf_params &p = h->p;
  IntegrateArgs a;
  for (int i = 0; i < 12; ++i) a.m[i] = T[i];
  a.fx = p.fx;
  a.fy = p.fy;
  a.cx = p.cx;
  a.cy = p.cy;
  a.zmin = p.min_sensor_dist;
  a.zmax = p.max_sensor_dist;
  a.pos = p.max_dist_pos;
  a.neg = p.max_dist_neg;
  a.wmax = p.max_weight;
  a.pos_over_neg = p.max_dist_pos / p.max_dist_neg;
  a.W = p.image_width;
  a.H = p.image_height;
  a.nx = h->nx;
  a.ny = h->ny;
  a.qpr = (h->nx + 3) / 4;
  const int planes = h->z_end - h->z_begin;
  a.rows = h->ny * planes;
  a.z_global0 = h->z_begin;
  a.zl0 = h->z_begin - h->z_first;
  int l2 = 0;
  while ((1 << l2) < a.qpr && l2 < 8) ++l2;
  a.log2TX = l2;
  a.TX = 1 << l2;
  a.TY = 256 / a.TX;
  a.xchunks = (unsigned)((a.qpr + a.TX - 1) / a.TX);
  const int64_t row_groups = ((int64_t)a.rows + a.TY - 1) / a.TY;
  const int64_t tiles = row_groups * a.xchunks;
  if (tiles > 0xFFFFFFFFll) {
    tsdf_set_error("slab too large for one launch");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  a.n_tiles = (unsigned)tiles;
  a.pitch = h->pitch;

  TSDF_HIP_TRY(hipMemsetAsync(h->counter, 0, sizeof(unsigned long long), h->stream));
  // A pose with a non-finite (or absurdly large) entry makes g.x/g.y/g.z non-finite or out of sensor
  // range for every voxel, and the reference then observes nothing (u/v become INT_MIN or g.z fails
  // hpp:146 / .cpp:616).  Same here, without launching: the kernel may assume finite arithmetic.
  bool pose_ok = true;
  for (int i = 0; i < 12; ++i) pose_ok &= std::isfinite(T[i]) && fabsf(T[i]) <= 1e15f;
  if (!pose_ok) {
    if (n_observed) {
      TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
      *n_observed = 0;
    }
    return TSDF_HIP_OK;
  }
  const unsigned grid = (unsigned)std::min<int64_t>(tiles, (int64_t)256 * tsdf_tuning().blocks_per_cu);
  const bool color = p.integrate_color != 0;
  if (color && !d_bgra) {
    tsdf_set_error("integrate_color is set but no colour image was given");
    return TSDF_HIP_E_INVALID;
  }
  const bool pipe = tsdf_tuning().pipeline != 0;
  const bool skip = tsdf_tuning().skip_unchanged != 0;
#define LAUNCH(ORDER, COLOR, PIPE, SKIP)                                                                   \
  hipLaunchKernelGGL((k_integrate<ORDER, COLOR, PIPE, SKIP>), dim3(grid), dim3(256), 0, h->stream, a, h->d, \
                     h->w, h->rgb, d_depth, d_bgra, h->ctr[0], h->ctr[1], h->ctr[2], h->counter)
#define LAUNCH2(ORDER, COLOR)             \
  do {                                    \
    if (pipe && skip)                     \
      LAUNCH(ORDER, COLOR, true, true);   \
    else if (pipe)                        \
      LAUNCH(ORDER, COLOR, true, false);  \
    else if (skip)                        \
      LAUNCH(ORDER, COLOR, false, true);  \
    else                                  \
      LAUNCH(ORDER, COLOR, false, false); \
  } while (0)
  if (p.xform_order == TSDF_XFORM_PCL_SSE) {
    if (color)
      LAUNCH2(TSDF_XFORM_PCL_SSE, true);
    else
      LAUNCH2(TSDF_XFORM_PCL_SSE, false);
  } else {
    if (color)
      LAUNCH2(TSDF_XFORM_LEFT_TO_RIGHT, true);
    else
      LAUNCH2(TSDF_XFORM_LEFT_TO_RIGHT, false);
  }
#undef LAUNCH2
#undef LAUNCH
  TSDF_HIP_TRY(hipGetLastError());
  if (n_observed) {
    unsigned long long c = 0;
    TSDF_HIP_TRY(hipMemcpyAsync(&c, h->counter, sizeof c, hipMemcpyDeviceToHost, h->stream));
    TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
    *n_observed = c;
  }
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_integrate_device(tsdf_handle h, const float *d_depth, const uint32_t *d_bgra,
                                         const float cam_from_vol[12], uint64_t *n_observed) {
  if (!h || !d_depth || !cam_from_vol) return TSDF_HIP_E_INVALID;
  TSDF_HIP_TRY(hipSetDevice(h->device));
  return launch_integrate(h, d_depth, d_bgra, cam_from_vol, n_observed);
}

extern "C" int tsdf_hip_integrate(tsdf_handle h, const float *depth, const uint8_t *bgra,
                                  const float cam_from_vol[12], uint64_t *n_observed) {
  if (!h || !depth || !cam_from_vol) return TSDF_HIP_E_INVALID;
  TSDF_HIP_TRY(hipSetDevice(h->device));
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  TSDF_HIP_TRY(hipMemcpyAsync(h->frame_depth, depth, npx * sizeof(float), hipMemcpyHostToDevice, h->stream));
  const bool color = h->p.integrate_color != 0;
  if (color) {
    if (!bgra) {
      tsdf_set_error("integrate_color is set but no colour image was given");
      return TSDF_HIP_E_INVALID;
    }
    TSDF_HIP_TRY(hipMemcpyAsync(h->frame_bgra, bgra, npx * 4, hipMemcpyHostToDevice, h->stream));
  }
  int rc = launch_integrate(h, h->frame_depth, color ? h->frame_bgra : nullptr, cam_from_vol, n_observed);
  if (rc) return rc;
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  return TSDF_HIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Test hooks: run the shared-reciprocal dividers of tsdf_div.h on arbitrary operands so the tests
// can compare them bit for bit with IEEE division done on the host.
static __global__ void k_selftest_div32(const float *a, const float *b, float *out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Rcp32 r = rcp32_prepare(b[i]);
  out[i] = div32(a[i], b[i], r);
}

static __global__ void k_selftest_div64(const double *a, const double *b, double *out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Rcp64 r = rcp64_prepare(b[i]);
  out[i] = div64(a[i], r);
}

template <typename T, typename K>
static int selftest_div(K kernel, const T *a, const T *b, T *out, size_t n) {
  if (!a || !b || !out || !n) return TSDF_HIP_E_INVALID;
  T *da = nullptr, *db = nullptr, *dout = nullptr;
  TSDF_HIP_TRY(hipMalloc(&da, n * sizeof(T)));
  TSDF_HIP_TRY(hipMalloc(&db, n * sizeof(T)));
  TSDF_HIP_TRY(hipMalloc(&dout, n * sizeof(T)));
  TSDF_HIP_TRY(hipMemcpy(da, a, n * sizeof(T), hipMemcpyHostToDevice));
  TSDF_HIP_TRY(hipMemcpy(db, b, n * sizeof(T), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da, db, dout, n);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipMemcpy(out, dout, n * sizeof(T), hipMemcpyDeviceToHost));
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dout);
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_selftest_div_f32(const float *a, const float *b, float *out, size_t n) {
  return selftest_div<float>(k_selftest_div32, a, b, out, n);
}

extern "C" int tsdf_hip_selftest_div_f64(const double *a, const double *b, double *out, size_t n) {
  return selftest_div<double>(k_selftest_div64, a, b, out, n);
}

// Calibration sweep for the HBM counters: read-modify-write every float4 of the d and w planes (and
// rgb when present) of the owned slab with the integrate kernel's access shape (16 B per lane per
// plane, grid-stride).  Values are written back unchanged; the byte count is known exactly:
// planes * voxels * 4 B read and the same written (MI355X_MICROARCH.md: calibrate FETCH_SIZE /
// WRITE_SIZE on a known byte count in your own access pattern).
static __global__ void __launch_bounds__(256)
k_calib_rmw(float *__restrict__ D, float *__restrict__ Wt, uint32_t *__restrict__ RGB, int64_t first4,
            int64_t n4, float addv /* 0 at run time */, uint32_t xorv /* 0 at run time */) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 *pd = reinterpret_cast<float4 *>(D) + first4 + i;
    float4 *pw = reinterpret_cast<float4 *>(Wt) + first4 + i;
    float4 d4 = *pd, w4 = *pw;
    // run-time zeros keep the stores alive without changing any value
    d4.x += addv;
    d4.y += addv;
    w4.x += addv;
    w4.y += addv;
    *pd = d4;
    *pw = w4;
    if (RGB) {
      uint4 *pc = reinterpret_cast<uint4 *>(RGB) + first4 + i;
      uint4 c4 = *pc;
      c4.x ^= xorv;
      c4.y ^= xorv;
      *pc = c4;
    }
  }
}

extern "C" int tsdf_hip_selftest_sweep(tsdf_handle h, uint64_t *bytes_read, uint64_t *bytes_written) {
  if (!h) return TSDF_HIP_E_INVALID;
  TSDF_HIP_TRY(hipSetDevice(h->device));
  const int64_t plane = h->pitch * h->ny;
  const int64_t first4 = (int64_t)(h->z_begin - h->z_first) * plane / 4;
  const int64_t n4 = (int64_t)(h->z_end - h->z_begin) * plane / 4;
  const unsigned grid = 256u * (unsigned)tsdf_tuning().blocks_per_cu;
  hipLaunchKernelGGL(k_calib_rmw, dim3(grid), dim3(256), 0, h->stream, h->d, h->w, h->rgb, first4, n4, 0.f, 0u);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  const uint64_t planes = h->rgb ? 3 : 2;
  if (bytes_read) *bytes_read = planes * (uint64_t)n4 * 16u;
  if (bytes_written) *bytes_written = planes * (uint64_t)n4 * 16u;
  return TSDF_HIP_OK;
}
