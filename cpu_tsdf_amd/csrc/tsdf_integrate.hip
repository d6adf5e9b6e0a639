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

#include "tsdf_common.h"

struct IntegrateArgs {
  float m[12];       // cam_from_vol, row-major 3x4
  double fx, fy, cx, cy;
  float zmin, zmax;  // min/max_sensor_dist_
  float pos, neg;    // max_dist_pos_/neg_
  float wmax;        // max_weight_
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

// x86 cvttsd2si semantics: NaN / out of range -> 0x80000000 ("integer indefinite").
// The reference's `u = (double expression)` (tsdf_volume_octree.cpp:614-615) compiles to that.
static __device__ __forceinline__ int cvtt_f64_i32(double v) {
  return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : INT_MIN;
}

struct VoxelObs {
  bool act;
  float dn;
  uint32_t bgra;
};

// One voxel of updateVoxel, leaf branch (hpp:143-198), everything except the read-modify-write.
template <bool COLOR>
static __device__ __forceinline__ VoxelObs
observe(const IntegrateArgs &a, float gx, float gy, float gz, const float *__restrict__ depth,
        const uint32_t *__restrict__ bgra) {
  VoxelObs o;
  o.act = false;
  o.dn = 0.f;
  o.bgra = 0u;
  // hpp:146  if (v_g.z < min_sensor_dist_ || v_g.z > max_sensor_dist_) return 0
  if (gz < a.zmin || gz > a.zmax) return o;
  // reprojectPoint, tsdf_volume_octree.cpp:611-617 -- float * double / float + double, then (int)
  const int u = cvtt_f64_i32((double)gx * a.fx / (double)gz + a.cx);
  const int v = cvtt_f64_i32((double)gy * a.fy / (double)gz + a.cy);
  if (!(gz > 0.f && u >= 0 && u < a.W && v >= 0 && v < a.H)) return o;
  const int pix = v * a.W + u;
  const float z = depth[pix];
  if (isnan(z)) return o;  // hpp:152 (only NaN is rejected; 0 and Inf are not)
  float dn = z - gz;       // hpp:159
  if (dn > a.pos)
    dn = a.pos;            // hpp:189-192
  else if (dn < -a.neg)
    return o;              // hpp:193-196
  dn = dn / a.neg;         // hpp:198
  o.act = true;
  o.dn = dn;
  if (COLOR) o.bgra = bgra[pix];
  return o;
}

// OctreeNode::addObservation with w_new = 1 (octree.cpp:152-163; both weightings of hpp:200-204
// are unreachable: no setter for weight_by_depth_/weight_by_variance_).
static __device__ __forceinline__ void add_observation(float &d, float &w, float dn, float wmax) {
  const float wn = 1.f;
  d = (d * w + dn * wn) / (w + wn);
  w = w + wn;
  if (w > wmax) w = wmax;
}

// RGBNode::addObservation (octree.cpp:328-337): per channel (uint8)((w*c + w_new*c_new)/(w+w_new))
// with the OLD w, truncating.
static __device__ __forceinline__ uint32_t blend_rgb(uint32_t rgb, uint32_t bgra, float w) {
  const float wn = 1.f;
  const float wsum = w + wn;
  const uint32_t r0 = rgb & 255u, g0 = (rgb >> 8) & 255u, b0 = (rgb >> 16) & 255u;
  const uint32_t bn = bgra & 255u, gn = (bgra >> 8) & 255u, rn = (bgra >> 16) & 255u;
  const uint32_t r = (uint32_t)(uint8_t)((w * (float)r0 + wn * (float)rn) / wsum);
  const uint32_t g = (uint32_t)(uint8_t)((w * (float)g0 + wn * (float)gn) / wsum);
  const uint32_t b = (uint32_t)(uint8_t)((w * (float)b0 + wn * (float)bn) / wsum);
  return r | (g << 8) | (b << 16);
}

template <int ORDER, bool COLOR>
static __global__ void __launch_bounds__(256)
k_integrate(const IntegrateArgs a, float *__restrict__ D, float *__restrict__ Wt, uint32_t *__restrict__ RGB,
            const float *__restrict__ depth, const uint32_t *__restrict__ bgra,
            const float *__restrict__ ctrx, const float *__restrict__ ctry, const float *__restrict__ ctrz,
            unsigned long long *__restrict__ n_obs) {
  const unsigned tid = threadIdx.x;
  const unsigned tx = tid & (unsigned)(a.TX - 1);
  const unsigned ty = tid >> a.log2TX;
  unsigned cnt = 0;

  for (unsigned t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    const unsigned rg = t / a.xchunks;
    const unsigned xc = t - rg * a.xchunks;
    const unsigned row = rg * (unsigned)a.TY + ty;
    const unsigned xq = xc * (unsigned)a.TX + tx;
    if (row >= (unsigned)a.rows || xq >= (unsigned)a.qpr) continue;
    const unsigned zl = row / (unsigned)a.ny;
    const unsigned y = row - zl * (unsigned)a.ny;
    const int x4 = (int)xq * 4;

    const float cy = ctry[y];
    const float cz = ctrz[a.z_global0 + (int)zl];
    const float4 cx4 = *reinterpret_cast<const float4 *>(ctrx + x4);
    const float cxs[4] = {cx4.x, cx4.y, cx4.z, cx4.w};

    // pcl::transformPoint (hpp:145).  Row-shared partial sums first.
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

    VoxelObs obs[4];
    bool any = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (ORDER == TSDF_XFORM_PCL_SSE)
          g[r] = cxs[j] * a.m[4 * r] + s[r];
        else
          g[r] = ((a.m[4 * r] * cxs[j] + p1[r]) + p2[r]) + a.m[4 * r + 3];
      }
      obs[j] = observe<COLOR>(a, g[0], g[1], g[2], depth, bgra);
      if (x4 + j >= a.nx) obs[j].act = false;  // padding lanes of a partial quad
      any |= obs[j].act;
    }
    if (!any) continue;

    const int64_t idx = ((int64_t)(a.zl0 + (int)zl) * a.ny + y) * a.pitch + x4;
    float4 d4 = *reinterpret_cast<const float4 *>(D + idx);
    float4 w4 = *reinterpret_cast<const float4 *>(Wt + idx);
    float dv[4] = {d4.x, d4.y, d4.z, d4.w};
    float wv[4] = {w4.x, w4.y, w4.z, w4.w};
    if (COLOR) {
      uint4 c4 = *reinterpret_cast<const uint4 *>(RGB + idx);
      uint32_t cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (obs[j].act) cv[j] = blend_rgb(cv[j], obs[j].bgra, wv[j]);
      *reinterpret_cast<uint4 *>(RGB + idx) = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (obs[j].act) {
        add_observation(dv[j], wv[j], obs[j].dn, a.wmax);
        ++cnt;
      }
    *reinterpret_cast<float4 *>(D + idx) = make_float4(dv[0], dv[1], dv[2], dv[3]);
    *reinterpret_cast<float4 *>(Wt + idx) = make_float4(wv[0], wv[1], wv[2], wv[3]);
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
  const unsigned grid = (unsigned)std::min<int64_t>(tiles, 256 * 8);
  const bool color = p.integrate_color != 0;
  if (color && !d_bgra) {
    tsdf_set_error("integrate_color is set but no colour image was given");
    return TSDF_HIP_E_INVALID;
  }
#define LAUNCH(ORDER, COLOR)                                                                               \
  hipLaunchKernelGGL((k_integrate<ORDER, COLOR>), dim3(grid), dim3(256), 0, h->stream, a, h->d, h->w, h->rgb, \
                     d_depth, d_bgra, h->ctr[0], h->ctr[1], h->ctr[2], h->counter)
  if (p.xform_order == TSDF_XFORM_PCL_SSE) {
    if (color)
      LAUNCH(TSDF_XFORM_PCL_SSE, true);
    else
      LAUNCH(TSDF_XFORM_PCL_SSE, false);
  } else {
    if (color)
      LAUNCH(TSDF_XFORM_LEFT_TO_RIGHT, true);
    else
      LAUNCH(TSDF_XFORM_LEFT_TO_RIGHT, false);
  }
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
