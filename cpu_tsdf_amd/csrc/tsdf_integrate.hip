// libtsdf_hip.so -- integrateCloud: one thread per voxel quad, project-and-weighted-average.
//
// Replaces TSDFVolumeOctree::integrateCloud / updateVoxel
// (include/cpu_tsdf/impl/tsdf_volume_octree.hpp:48-103, :113-218) on a flat SoA grid.
// The arithmetic below follows the reference line by line in fp32 with NO fused multiply-add
// (the reference's CMake build has no -march flag, so x86-64 emits separate mul/add) and IEEE
// division; this file is compiled with -ffp-contract=off.  What has no dense counterpart (octree
// split hpp:161-187, prune :122-142, return codes :209-214, the surface pre-split pass :56-90) is
// dropped: on a dense grid every voxel is a finest leaf.  The coarse frustum cull (:93-94,
// getFrustumCulledVoxels) becomes k_cull, a conservative per-block pre-pass that never changes results.
//
// Memory behaviour: a thread owns 4 x-consecutive voxels (one 16-byte vector per plane); a wave touches
// 1 KiB contiguous per plane.  The planes are read only if at least one of the four voxels reaches
// addObservation and written back only if a word changed; SURVEY's algorithmic traffic is 16 B (24 B colour)
// per observed voxel plus the frame gather, which is served by L2 (the 640x480 frame is 2.4 MB); the PACKED
// layout (tsdf_common.h) moves 10 B (16 B).  Co-limited by VALU issue and HBM; no MFMA.
#include <limits.h>
#include <string.h>
#include <math.h>

#include <algorithm>
#include <cmath>

#include "tsdf_common.h"
#include "tsdf_div.h"
#include "tsdf_buffer.h"
#ifdef TSDF_HIP_TEST_HOOKS
#include "tsdf_hip_test.h"
#endif

// (compile-time switch used by add_observation_fast below: its default has to come first)
#ifndef TSDF_COLOR_PK
#define TSDF_COLOR_PK 2  // 2: colour bytes through v_cvt_pk_u8_f32 (one convert-and-pack per channel, rounds to nearest even:
                         // tests/test_div_gpu.py pins that).  Round 2 measured it 2 % SLOWER in a kernel that waited for memory;
                         // in round 5's issue-bound kernels it is the faster form (k_integrate2 11.76 -> 11.53 ms per frame, the
                         // configs[4] slab 13.37 -> 13.10, k_integrate 13.6 either way: profiles/r05_ab_diet_call2.txt).  0 = three
                         // truncating conversions + shifts and ors; 1 assumed truncation of the packed form and is WRONG.
#endif

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct IntegrateArgs {
  float m[12];        // cam_from_vol, row-major 3x4
  float fxf, fyf, cxf, cyf;  // the same, rounded to float (fast projection path); cxf/cyf carry the +band shift
  float hb_u, hb_v;          // certificate threshold: fract(R~) > 2 * band
  float hb_max;              // max(hb_u, hb_v): the ALLIN instance certifies a quad with ONE compare of min(fract) against it
  unsigned kcap, kinc;       // PACKED count after an observation, byte 3 of a word: min(k, kcap) + kinc == min(k + 1, kmax)
  int hinge_fixed;           // PACKED: (p*w + p)/(w + 1) == p for every stored weight w, p = pos_over_neg (host-checked)
  int neg_in_window;         // max_dist_pos/neg inside the scale-free divider's window
  unsigned kmax;             // PACKED layout: saturation count ceil(max_weight)
  int wmax_is_int;           // PACKED layout: max_weight is an integer (then every stored weight is)
  float zmin, zmax;   // min/max_sensor_dist_
  float zlo;          // gz > zlo  <=>  !(gz < zmin) && gz > 0  (see make_args)
  float pos, neg;     // max_dist_pos_/neg_
  float wmax;         // max_weight_
  float pos_over_neg; // max_dist_pos_ / max_dist_neg_ (IEEE fp32, host)
  int W, H;
  int ny;             // rows this launch may touch (counted from the launch's first row)
  int plane_rows;     // rows of a whole plane: the address stride between planes
  int qpr;            // quads per row = ceil(nx/4) (counted from the launch's first quad)
  int z_global0;      // global z of the first integrated plane
  int zl0;            // allocated-plane index of the first integrated plane
  int log2TX, TX, TY; // thread tile: TX quads along x, TY rows along y (TX*TY == 256)
  int rpb;            // row groups (of TY rows) per block
  unsigned bgra_off;  // byte offset of the colour image from the depth image (same buffer descriptor)
  int expf_fused_r;      // the host libm's expf fuses r = InvLn2N * x - k (tsdf_expf_glibc; probed by tsdf_hip_set_weighting)
  int ref_cull;          // replicate the reference's frustum cull (getFrustumCulledVoxels): six plane tests per voxel centre
  float cull[24];        // its planes l, r, t, b, far, near (tsdf_hip_set_reference_cull), 4 floats each
  int band_fx, band_fy;  // "band seen" flags: cells of 64 x 4 x 1 voxels, [allocated plane][fy][fx] (tsdf_common.h)
  int implied_d;         // PACKED: in a cell whose flag is still 0 a voxel's distance follows from its count (see k_integrate)
  int x_abs0, y_abs0;    // grid x / y of the launch's first voxel / row (the launch may be a sub-box of the slab)
  int zfast;             // the hardware grid is (planes, row groups, x chunks) instead of (x chunks, row groups, planes): see tsdf_block_coords
  int64_t pitch;
};

// reprojectPoint (tsdf_volume_octree.cpp:611-617), EXACT: u = (int)(x*fx/z + cx) evaluated in double,
// truncation toward zero, for a voxel that already passed the range test (g.z > 0, finite).
// u and v divide by the same g.z, so the fp64 reciprocal is refined once (tsdf_div.h).
// v_cvt_i32_f64 saturates where x86's cvttsd2si returns INT_MIN; both land outside [0, W), and NaN
// cannot occur (the host rejects non-finite / absurd poses before launching).
static __device__ __forceinline__ int project_exact(const IntegrateArgs &a, const double *__restrict__ cam,
                                                    float gx, float gy, float gz) {
  // [phase: exact fp64 re-projection (rare: uncertified voxels)]
  const Rcp64 rz = rcp64_prepare((double)gz);
  const int u = (int)(div64((double)gx * cam[0], rz) + cam[2]);  // cam = fx, fy, cx, cy
  const int v = (int)(div64((double)gy * cam[1], rz) + cam[3]);
  const bool in = (unsigned)u < (unsigned)a.W && (unsigned)v < (unsigned)a.H;
  return in ? v * a.W + u : -1;
}

// The same in fp32, with a certificate.  R~ = fma(g*f~, rcp(z), c~ + s) is the reference's double value R shifted
// by s = band, computed with an error below band whenever R~ lies within a pixel of the image (one rounding of g*f~,
// v_rcp_f32's 1 ulp taken as 2^-22, the conversions of f and of c + s, the single rounding of the fma; the host
// computes band with a 1.5x margin), so R lies in (R~ - 2 band, R~).  Therefore:
//   * fract(R~) > 2 band  =>  floor(R~) < R < R~ < floor(R~) + 1: R and R~ share the integer cell, and a cell has one
//     truncation ((-1, 0) and [0, 1) both give 0, for R and for R~ alike);
//   * R~ far outside the image  =>  R is outside too, and so is trunc(R~) (or the point is flagged below).
// Anything else (an R~ sitting on an integer, a non-finite R~ whose fract is 0 or NaN) is flagged ambiguous and
// recomputed by project_exact.  One multiply, one fma, one fract, one compare and one conversion per coordinate.
template <bool ALLIN = false>
static __device__ __forceinline__ int project_fast(const IntegrateArgs &a, float gx, float gy, float gz,
                                                   bool &ambiguous, uint32_t *margin = nullptr) {
  const float y = __builtin_amdgcn_rcpf(gz);
  const float ru = __builtin_fmaf(gx * a.fxf, y, a.cxf);
  const float rv = __builtin_fmaf(gy * a.fyf, y, a.cyf);
  const int u = (int)ru, v = (int)rv;  // v_cvt_i32_f32: truncates, saturates, NaN -> 0
  // ALLIN: the host has proved that every voxel of the launch projects at least a pixel inside the image border and
  // has 1e-3 <= g.z (launch_integrate, `all_inside`), so R~ is finite and positive, a certified (u, v) needs no bounds
  // test, and the certificate is handed back as ONE number -- min(fract(ru), fract(rv)), to be compared against
  // max(hb_u, hb_v): slightly stricter than the two separate tests -- so that the caller can certify a whole quad
  // with one compare.  The fractions lie in [0, 1): their bit patterns order like the values, and integer minima
  // (v_min3_u32) need no NaN canonicalisation.
  if (ALLIN) {
    *margin = min(__float_as_uint(__builtin_amdgcn_fractf(ru)), __float_as_uint(__builtin_amdgcn_fractf(rv)));
    ambiguous = false;
    return (int)__umul24((unsigned)v, (unsigned)a.W) + u;
  }
  const bool cert = __builtin_amdgcn_fractf(ru) > a.hb_u && __builtin_amdgcn_fractf(rv) > a.hb_v;  // false for NaN
  // an uncertified point far outside the image is recomputed needlessly, never wrongly
  ambiguous = !cert;
  const bool in = (unsigned)u < (unsigned)a.W && (unsigned)v < (unsigned)a.H;
  return in ? (int)__umul24((unsigned)v, (unsigned)a.W) + u : -1;  // W, H < 2^24 (checked on the host)
}

// ALLIN: pcl::transformPoint (hpp:145) + the certified fp32 reprojectPoint (project_fast<true>) for the four voxels of a
// quad, on float PAIRS -- voxels (0, 1) and (2, 3): every product, sum and fma below is the scalar code's own operation on
// the same operands (v_pk_mul / v_pk_add / v_pk_fma_f32 round each half like the scalar instruction), so g, R~ and the
// certificate are bit for bit those of the scalar form; written on pairs so that the compiler keeps ONE pairing from the
// centres to the pixel index (the scalar form cost eight register shuffles per row).  cxv = the quad's x centres, ytv = the
// row's part of the transform (k_integrate's s_yt), zt = the plane's part (used by the left-to-right order only).
template <int ORDER>
static __device__ __forceinline__ void project_quad_allin(const IntegrateArgs &a, const float (&m)[12], const double *__restrict__ cam,
                                                          const f4 cxv, const f4 ytv, const float (&zt)[3], int (&pix)[4],
                                                          float (&gzs)[4]) {
  // [phase: transform (hpp:143-145)]
  const f2 cxa = {cxv.x, cxv.y}, cxb = {cxv.z, cxv.w};
  const float yt[3] = {ytv.x, ytv.y, ytv.z};
  f2 ga[3], gb[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (ORDER == TSDF_XFORM_PCL_SSE) {
      ga[q] = cxa * m[4 * q] + yt[q];
      gb[q] = cxb * m[4 * q] + yt[q];
    } else {
      ga[q] = ((cxa * m[4 * q] + yt[q]) + zt[q]) + m[4 * q + 3];
      gb[q] = ((cxb * m[4 * q] + yt[q]) + zt[q]) + m[4 * q + 3];
    }
  }
  // [phase: project (.cpp:611-617, certified fp32)]
  const f2 ya = {__builtin_amdgcn_rcpf(ga[2].x), __builtin_amdgcn_rcpf(ga[2].y)};
  const f2 yb = {__builtin_amdgcn_rcpf(gb[2].x), __builtin_amdgcn_rcpf(gb[2].y)};
  const f2 cxf2 = {a.cxf, a.cxf}, cyf2 = {a.cyf, a.cyf};
  const f2 rua = __builtin_elementwise_fma(ga[0] * a.fxf, ya, cxf2), rub = __builtin_elementwise_fma(gb[0] * a.fxf, yb, cxf2);
  const f2 rva = __builtin_elementwise_fma(ga[1] * a.fyf, ya, cyf2), rvb = __builtin_elementwise_fma(gb[1] * a.fyf, yb, cyf2);
  const float ru[4] = {rua.x, rua.y, rub.x, rub.y}, rv[4] = {rva.x, rva.y, rvb.x, rvb.y};
  gzs[0] = ga[2].x, gzs[1] = ga[2].y, gzs[2] = gb[2].x, gzs[3] = gb[2].y;
  // the certificate: the fractions lie in [0, 1), so their bit patterns order like the values; ONE compare of the least of
  // the eight against max(hb_u, hb_v) certifies the quad
  // [phase: certificate + pixel index]
  uint32_t fu[4], fv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    fu[j] = __float_as_uint(__builtin_amdgcn_fractf(ru[j]));
    fv[j] = __float_as_uint(__builtin_amdgcn_fractf(rv[j]));
    pix[j] = (int)__umul24((unsigned)(int)rv[j], (unsigned)a.W) + (int)ru[j];  // v_cvt_i32_f32 truncates; W, H < 2^24
  }
  const uint32_t hb = __float_as_uint(a.hb_max);
  const uint32_t least = min(min(min(fu[0], fv[0]), min(min(fu[1], fv[1]), fu[2])), min(min(fv[2], fu[3]), fv[3]));
  // [phase: exact fp64 re-projection (rare: uncertified voxels)]
  if (!(least > hb)) {  // rare (a fraction ~4 * band of the voxels): the exact fp64 projection, a copy per voxel of the quad,
                        // each under its own exec mask (an empty one is a skipped branch)
    const float gx[4] = {ga[0].x, ga[0].y, gb[0].x, gb[0].y}, gy[4] = {ga[1].x, ga[1].y, gb[1].x, gb[1].y};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (!(min(fu[j], fv[j]) > hb)) pix[j] = project_exact(a, cam, gx[j], gy[j], gzs[j]);
  }
}

// std::exp(float) as the reference's host evaluates it (hpp:204): glibc's expf (sysdeps/ieee754/flt-32/e_expf.c, the 2017
// table-driven algorithm [glibc-recall], NOT correctly rounded: it differs from the rounded fp64 exp on 0.035 % of the
// floats).  Restated operation by operation: z = x * 32/ln2 in double, k = nearest integer through the 1.5 * 2^52 shift,
// r = z - k, 2^(k/32) from a 32-entry table, a cubic in r.  One operation depends on how the host's libm was compiled:
// x86-64 glibc selects an FMA build on CPUs that have it, and that build fuses `r = InvLn2N * x - k`; `fused_r` says
// which one this host runs (probed with the one float the two forms disagree on, tsdf_hip_set_weighting).  Pinned
// against the host's expf on every float in +-(2^-26 .. 104) by tests/test_wvar_gpu.py.
static __device__ const uint64_t k_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

static __device__ float tsdf_expf_glibc(float x, bool fused_r) {
  const uint32_t ix = __float_as_uint(x), abstop = (ix >> 20) & 0x7ffu;
  if (abstop >= (0x42b00000u >> 20)) {  // |x| >= 88 or NaN
    if (ix == 0xff800000u) return 0.f;                 // -inf
    if (abstop >= (0x7f800000u >> 20)) return x + x;   // inf, NaN
    if (x > 0x1.62e42ep6f) return INFINITY;            // overflow
    if (x < -0x1.9fe368p6f) return 0.f;                // underflow
    if (x < -0x1.9d1d9ep6f) return 0x1p-149f;          // __math_may_uflowf: 0x1.4p-75f * 0x1.4p-75f rounds to the least subnormal
  }
  const double N = 32., InvLn2N = 0x1.71547652b82fep+0 * N, SHIFT = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
  const double xd = (double)x;
  double z = InvLn2N * xd;
  double kd = z + SHIFT;
  const uint64_t ki = (uint64_t)__double_as_longlong(kd);
  kd -= SHIFT;
  const double r = fused_r ? __builtin_fma(InvLn2N, xd, -kd) : z - kd;
  const uint64_t t = k_exp2f_tab[ki & 31u] + (ki << 47);
  const double s = __longlong_as_double((long long)t);
  z = C0 * r + C1;
  const double r2 = r * r;
  double y = C2 * r + 1.;
  y = z * r2 + y;
  y = y * s;
  return (float)y;
}

// pcl::FrustumCulling's verdict on a voxel centre (tsdf_volume_octree.cpp:619-652 -> filters/impl/frustum_culling.hpp
// [PCL-recall]): `pt.dot(plane) <= 0` for all six planes, pt = (x, y, z, 1) as Vector4f, the 4-term dot reduced as
// (p0 + p1) + (p2 + p3) [Eigen-recall]; no FMA (this file is built with -ffp-contract=off).  A NaN coordinate fails.
static __device__ __forceinline__ bool reference_cull_keeps(const IntegrateArgs &a, float x, float y, float z) {
  bool keep = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float *pl = a.cull + 4 * k;
    keep = keep && ((x * pl[0] + y * pl[1]) + (z * pl[2] + 1.0f * pl[3]) <= 0.f);
  }
  return keep;
}

// Scale-free divider (LLVM's fp32 division ladder without v_div_scale / v_div_fixup) for a divisor
// prepared by rcp32_prepare.  Exact (== IEEE) when the divisor is in [2^-40, 2^40] and the numerator is +0
// or has magnitude in [2^-100, 2^40]: nothing in the ladder can then overflow, underflow or go denormal.
static __device__ __forceinline__ float div32_fast(float a, const Rcp32 &r) {
  const float q0 = a * r.y;
  const float r0 = __builtin_fmaf(r.nb, q0, a);
  const float q1 = __builtin_fmaf(r0, r.y, q0);
  const float r1 = __builtin_fmaf(r.nb, q1, a);
  return __builtin_fmaf(r1, r.y, q1);
}

static __device__ __forceinline__ bool numerator_ok(float v) {  // +0, or 2^-100 <= |v| <= 2^40 (no NaN/Inf)
  const float m = fabsf(v);
  return (m >= 0x1p-100f && m <= 0x1p40f) || __float_as_uint(v) == 0u;
}

// OctreeNode::addObservation with w_new = 1 (octree.cpp:152-163; both weightings of hpp:200-204 are
// unreachable: no setter for weight_by_depth_/weight_by_variance_), and RGBNode::addObservation
// (octree.cpp:328-337): per channel (uint8)((w*c + w_new*c_new)/(w+w_new)) with the OLD w, truncating.
// General version: the compiler's full IEEE divisions.
template <bool COLOR>
static __device__ __forceinline__ void add_observation_ieee(float &d, float &w, uint32_t &rgb, float dn,
                                                            uint32_t bgra, float wmax) {
  // [phase: IEEE fallback (rare)]
  const float wn = 1.f;
  const float wsum = w + wn;
  if (COLOR) {
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float c_old = (float)((rgb >> (8 * ch)) & 255u);
      const float c_new = (float)((bgra >> (16 - 8 * ch)) & 255u);  // PCL b,g,r,a -> r,g,b
      const float num = w * c_old + wn * c_new;
      out |= ((uint32_t)(uint8_t)(num / wsum)) << (8 * ch);
    }
    rgb = out;
  }
  d = (d * w + dn * wn) / wsum;
  w = wsum;
  if (w > wmax) w = wmax;
}

// Fast version, valid when update_is_safe(): w is an INTEGER in [0, 1024] and the distance numerator
// passes numerator_ok().
//  * distance: shared-reciprocal scale-free ladder == IEEE division (tests/test_div_gpu.py).
//  * colour: N = w*c_old + c_new is an exact integer < 2^18 and D = w + 1 an integer <= 1025, so the
//    reference's (uint8)fl(N/D) equals floor(N/D): fl() moves the quotient by < 2^-24 * 256, far less than
//    the 1/D that separates a non-integer N/D from the next integer.  floor(N/D) is computed as
//    trunc(fma(N, y, y/2)) = trunc((N + 1/2) * y): the true value lies >= 1/(2D) >= 4.8e-4 from the integers
//    on either side, the two roundings and y's error move it by < 256 * 2^-23 = 3.1e-5.
//    With TSDF_COLOR_PK the conversion and the packing are one v_cvt_pk_u8_f32 per channel (it writes one byte of a
//    word and keeps the others: `base` carries byte 3, the PACKED layout's count).  That instruction rounds to nearest
//    even (tsdf_hip_selftest_cvt_pk_u8 / tests/test_div_gpu.py), so the value converted is (N + 1/2) * y - 1/2, which
//    lies within 1/2 - 1/(2D) of floor(N/D).
template <bool COLOR, bool DIST = true>
static __device__ __forceinline__ void add_observation_fast(float &d, float &w, uint32_t &rgb, float dn,
                                                            uint32_t bgra, float wmax, const Rcp32 &rs, uint32_t base = 0u,
                                                            const float *hy_tab = nullptr) {
  // [phase: colour update (octree.cpp:328-337)]
  const float wsum = w + 1.f;  // rs = rcp32_prepare(wsum) (nb and y are all that is used)
  if (COLOR) {
#if TSDF_COLOR_PK
    // hy_tab: the offset comes with the count's table entry (KEntry) instead of being derived from y per voxel
    const float hy = hy_tab ? *hy_tab : TSDF_COLOR_PK == 2 ? __builtin_fmaf(0.5f, rs.y, -0.5f) : 0.5f * rs.y;
    const float t0 = __builtin_fmaf(__builtin_fmaf(w, (float)(rgb & 255u), (float)((bgra >> 16) & 255u)), rs.y, hy);
    const float t1 = __builtin_fmaf(__builtin_fmaf(w, (float)((rgb >> 8) & 255u), (float)((bgra >> 8) & 255u)), rs.y, hy);
    const float t2 = __builtin_fmaf(__builtin_fmaf(w, (float)((rgb >> 16) & 255u), (float)(bgra & 255u)), rs.y, hy);
    rgb = __builtin_amdgcn_cvt_pk_u8_f32(t0, 0u, __builtin_amdgcn_cvt_pk_u8_f32(t1, 1u, __builtin_amdgcn_cvt_pk_u8_f32(t2, 2u, base)));
#else
    const float hy = hy_tab ? *hy_tab : 0.5f * rs.y;
    const uint32_t q0 = (uint32_t)__builtin_fmaf(__builtin_fmaf(w, (float)(rgb & 255u), (float)((bgra >> 16) & 255u)), rs.y, hy);
    const uint32_t q1 = (uint32_t)__builtin_fmaf(__builtin_fmaf(w, (float)((rgb >> 8) & 255u), (float)((bgra >> 8) & 255u)), rs.y, hy);
    const uint32_t q2 = (uint32_t)__builtin_fmaf(__builtin_fmaf(w, (float)((rgb >> 16) & 255u), (float)(bgra & 255u)), rs.y, hy);
    rgb = q0 | (q1 << 8) | (q2 << 16) | base;
#endif
  }
  // [phase: d update (octree.cpp:152-163)]
  if (DIST) d = div32_fast(d * w + dn, rs);
  w = wsum;
  if (w > wmax) w = wmax;
}

// The colour half of add_observation_fast for a PAIR of voxels on float pairs -- (0, 1), then (2, 3), one channel at a time:
// the same two fmas per channel and voxel (N = fma(w, c_old, c_new), t = fma(N, y, hy): v_pk_fma_f32 rounds each half like
// v_fma_f32, as in project_quad_allin), in half the instructions -- 12 packed fmas instead of 24 (round 6).  y / hy: Rcp32(k + 1).y
// and the rounding offset of each voxel's divisor (s_rcp); base: the word v_cvt_pk_u8_f32 packs into (byte 3 = the new count).
// MEASURED and left OFF (profiles/r06_ab_color_f2_call14.txt, five alternations at 2048^3 + colour): the pairs do not fit the 64
// registers of the eight-wave instance (spills on the row path: 18.7 ms); at seven waves (72 VGPRs, -12 vector instructions per
// wave-row, no scalar spills) 13.09 against 13.20 ms without it and 13.03 for the shipped eight-wave instance, 12.88 against 12.37 on a
// configs[4] slab -- a twelfth fewer vector instructions buy nothing: with colour this kernel is gated by its 58.6 GB per launch.
#ifndef TSDF_COLOR_F2
#define TSDF_COLOR_F2 0
#endif
static __device__ __forceinline__ void colour_pair_pk(const f2 w, const uint32_t c0a, const uint32_t c0b, const uint32_t csa, const uint32_t csb,
                                                      const f2 y, const f2 hy, const uint32_t base_a, const uint32_t base_b, uint32_t &cva, uint32_t &cvb) {
  f2 t[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const f2 o = {(float)((c0a >> (8 * ch)) & 255u), (float)((c0b >> (8 * ch)) & 255u)};
    const f2 n = {(float)((csa >> (16 - 8 * ch)) & 255u), (float)((csb >> (16 - 8 * ch)) & 255u)};  // PCL b,g,r,a -> r,g,b
    t[ch] = __builtin_elementwise_fma(__builtin_elementwise_fma(w, o, n), y, hy);
  }
  cva = __builtin_amdgcn_cvt_pk_u8_f32(t[0].x, 0u, __builtin_amdgcn_cvt_pk_u8_f32(t[1].x, 1u, __builtin_amdgcn_cvt_pk_u8_f32(t[2].x, 2u, base_a)));
  cvb = __builtin_amdgcn_cvt_pk_u8_f32(t[0].y, 0u, __builtin_amdgcn_cvt_pk_u8_f32(t[1].y, 1u, __builtin_amdgcn_cvt_pk_u8_f32(t[2].y, 2u, base_b)));
}

static __device__ __forceinline__ bool update_is_safe(float d, float w, float dn) {
  const bool w_ok = fabsf(w - 512.f) <= 512.f && __builtin_amdgcn_fractf(w) == 0.f;  // integer in [0, 1024]
  return w_ok && numerator_ok(d * w + dn);
}

// (buffer-descriptor access helpers: tsdf_buffer.h)

// Logical block coordinates (x chunk, row group, plane) and logical grid extents of an integrate launch.  The hardware hands
// out workgroups in the order x, y, z of the grid and deals consecutive ones to the eight XCDs in turn.  Default: x chunks
// fastest -- the ~1800 blocks in flight stream three or four whole planes, whose voxels project to the WHOLE image.  zfast:
// planes fastest -- the blocks in flight are one (x chunk, few row groups) column through all planes, which projects to a
// narrow bundle of rays: the frame pixels they gather stay within an XCD's 4 MB L2 even when the frame does not
// (1280x960 + colour = 9.8 MB, BASELINE configs[4]).
struct BlockCoords {
  unsigned bx, by, bz, gdx, gdy;
};
static __device__ __forceinline__ BlockCoords tsdf_block_coords(int zfast) {
  BlockCoords c;
  c.by = blockIdx.y, c.gdy = gridDim.y;
  c.bx = zfast ? blockIdx.z : blockIdx.x;
  c.bz = zfast ? blockIdx.x : blockIdx.z;
  c.gdx = zfast ? gridDim.z : gridDim.x;
  return c;
}

// Grid: x = chunks of TX quads along the row, y = groups of rpb*TY rows, z = planes.  No persistent
// blocks: the hardware dispatcher balances the tail, and no index needs an integer division.
// A thread owns one quad column (4 x-consecutive voxels) and walks rpb rows of it.  Everything that depends
// only on x -- the centre products c.x * m[r][0] -- is computed once; per row only the y/z part of the rigid
// transform is redone.  FASTPROJ selects the certified fp32 projection with exact fallback; otherwise every
// voxel goes through project_exact.  The x centre table is padded with NaN beyond nx, which fails the range
// test, so a partial last quad needs no predicate.  Planes whose four values did not change are not written
// back (free space: d stays at the hinge value; after weight saturation nothing changes).
// PACKED: the weight is the observation count k in byte 3 of the colour word (COLOR) or in the uint8 plane
// K8 (no colour); w = min(k, max_weight), k' = min(k + 1, kmax); a thread then moves 8 (5) bytes per voxel
// each way instead of 12 (8), and 1/(k+1) comes from a 256-entry LDS table of refined reciprocals.
// COUNT = accumulate the observed-voxel counter.
// Implied distances (see k_integrate's s_bin): the "band seen" flags of a block's cells as they stood BEFORE the launch, in
// the order of the block's own s_band (row group major, TX / 16 cells per row group); cells without voxels read 0.
static __device__ __forceinline__ void tsdf_flags_before(const IntegrateArgs &a, const BlockCoords &bc, const uint8_t *band,
                                                         uint8_t *s_bin, unsigned tid) {
  const int row0 = (int)bc.by * a.rpb * a.TY;
  const int rows = min(a.rpb * a.TY, a.ny - row0);
  const int lf = max(0, a.log2TX - 4), fxb = 1 << lf;  // cells per row group of the block: TX / 16, a power of two
  const int yg0 = (a.y_abs0 + row0) >> 2, yg1 = (a.y_abs0 + row0 + max(rows, 1) - 1) >> 2;
  const int xc0 = (a.x_abs0 + (int)bc.bx * a.TX * 4) >> 6;
  const int n_fl = (yg1 - yg0 + 1) << lf;
  const int n_all = ((a.rpb * a.TY + 3) >> 2) << lf;  // every cell tsdf_quiet_passes may look at (<= 1024)
  for (int i = (int)tid; i < n_all; i += 256) {
    const int yg = yg0 + (i >> lf), xc = xc0 + (i & (fxb - 1));
    s_bin[i] = i < n_fl && yg < a.band_fy && xc < a.band_fx ? band[((int64_t)(a.zl0 + (int)bc.bz) * a.band_fy + yg) * a.band_fx + xc] : (uint8_t)0;
  }
}
// ... and from them, per wave: bit r = none of the cells the wave's 64 quads lie in during pass r (rows ty + r * TY) had its
// flag set.  A wave covers max(1, 64 / TX) rows by min(TX, 64) quads per pass; lane r works out pass r.  (Passes beyond 63
// -- only with the rows_per_block knob turned up on a one-row tile -- read their distances.)
static __device__ __forceinline__ uint64_t tsdf_quiet_passes(const IntegrateArgs &a, const uint8_t *s_bin, unsigned tid) {
  const int lane = (int)(tid & 63u), w0 = (int)(tid & ~63u);  // the wave's first thread
  const int fxb = max(1, a.TX >> 4);
  const int rows_w = max(1, 64 >> a.log2TX), ncell = max(1, min(a.TX, 64) >> 4);
  const int ty0 = w0 >> a.log2TX, c0 = (w0 & (a.TX - 1)) >> 4;
  bool q = lane < a.rpb;
  if (q) {
    for (int rr = 0; rr < rows_w; ++rr)
      for (int c = 0; c < ncell; ++c) q &= s_bin[((ty0 + rr + lane * a.TY) >> 2) * fxb + c0 + c] == 0;
  }
  return __builtin_amdgcn_ballot_w64(q);
}

#ifndef TSDF_GUARD_ON_RESULT
#define TSDF_GUARD_ON_RESULT 1  // PACKED update: guard the divider on its result (v_cmp_class) instead of on its numerator
#endif
#ifndef TSDF_EARLY_VOXEL_LOADS
#define TSDF_EARLY_VOXEL_LOADS 2  // 1: a quad's voxel words are requested together with the frame gather (one memory round trip per
                                  // row instead of two, a fifth more traffic: measured in round 4, left off); 2 (round 5, ON): only
                                  // by the quads PREDICTED to be observed (their own outcome one row earlier) -- 13.8 -> 13.45 ms at
                                  // 2048^3 + colour, 9.96 -> 9.27 without, the configs[4] slab 12.87 -> 12.60, for 0.8 % more traffic
                                  // (58.19 -> 58.63 GB; plain early loads: 71.1 GB for 13.6 ms); profiles/r05_ab_diet_call3.txt
#endif

// Round 5 instruction diet of the PACKED instances (profiles/r05_isa_phase_mix.txt has the per-phase counts):
#ifndef TSDF_KTAB
#define TSDF_KTAB 0  // 1: everything addObservation derives from the count k alone -- the decoded weight, the refined reciprocal of
                     // k + 1, the colour rounding offset, the count after the observation -- comes from ONE 16-byte LDS entry per
                     // voxel (ds_read_b128) instead of an 8-byte entry + five VALU operations per voxel.  Measured in k_integrate
                     // (profiles/r05_ab_diet_call1.txt): 14.15 against 13.87 ms without it -- sixteen registers of table values
                     // in flight cost more than the operations they save (and with the v_cvt_pk_u8 colour path on top they spill:
                     // 16.3 ms).  OFF there; k_integrate2, which runs at 5 waves and has the registers, keeps its table.
#endif
#ifndef TSDF_LEAN_K
#define TSDF_LEAN_K 1  // (without TSDF_KTAB) the count's decode without its v_min, the next count without its mask: see the decode step
#endif
#ifndef TSDF_SWAR_K
#define TSDF_SWAR_K 1  // PACKED without colour: the quad's four count bytes are updated in one word (see k4n in k_integrate)
#endif
#ifndef TSDF_LEAN_BAND
#define TSDF_LEAN_BAND 1  // the in-band test of a quad behind one compare of the least raw distance (see there)
#endif
#ifndef TSDF_AMB_UNROLL
#define TSDF_AMB_UNROLL 1  // the exact fp64 re-projection of uncertified voxels: four straight copies, one per voxel of the quad,
                           // each under its own exec mask (an empty one is a skipped branch), instead of ONE copy in a loop that
                           // selects its operands by voxel index through ~90 exec-mask instructions per trip
#endif
#ifndef TSDF_PROJ_F2
#define TSDF_PROJ_F2 1  // ALLIN: transform + projection written on explicit float pairs (v_pk_mul / v_pk_add / v_pk_fma_f32 with
                        // the pairs (0,1), (2,3) throughout: no register shuffles), the x centres re-read from LDS per row
#endif
struct __attribute__((aligned(16))) KEntry {
  float w;      // tsdf_decode_w(k) = min(k, max_weight)
  float y;      // Rcp32(k + 1).y: the refined reciprocal of the divisor k + 1
  float hy;     // the colour average's rounding offset: y / 2 before a truncating conversion, y / 2 - 1 / 2 before v_cvt_pk_u8_f32
  uint32_t k1;  // min(k + 1, kmax) << 24: the count after this observation, in byte 3 of a word
};
static __device__ __forceinline__ KEntry tsdf_ktab_entry(const IntegrateArgs &a, unsigned k) {
  KEntry e;
  e.w = __builtin_fminf((float)k, a.wmax);
  e.y = rcp32_prepare((float)(k + 1u)).y;
  e.hy = TSDF_COLOR_PK == 2 ? __builtin_fmaf(0.5f, e.y, -0.5f) : 0.5f * e.y;
  e.k1 = min(k << 24, a.kcap) + a.kinc;  // saturate BEFORE adding (kmax == 255 would wrap byte 3); both 0 when kmax == 0
  return e;
}
#ifndef TSDF_SKIP_FIXED_HINGE
#define TSDF_SKIP_FIXED_HINGE 1  // PACKED: waves whose observed voxels all stay at the hinge value skip the d ladder
#endif
// Timing-only experiment switches (WRONG results; tools/ab_bound.sh builds them to find what binds the kernel):
//   TSDF_EXP_NO_STORE   no voxel plane is written back          TSDF_EXP_NO_GATHER  the frame is not read (constant far depth)
//   TSDF_EXP_NO_VLOAD   no voxel plane is read (hinge / zero)   TSDF_WPE_MAX        cap on resident waves per SIMD
#ifndef TSDF_EXP_NO_STORE
#define TSDF_EXP_NO_STORE 0
#endif
#ifndef TSDF_EXP_NO_GATHER
#define TSDF_EXP_NO_GATHER 0
#endif
#ifndef TSDF_EXP_NO_VLOAD
#define TSDF_EXP_NO_VLOAD 0
#endif
#if TSDF_EXP_NO_VLOAD && TSDF_EARLY_VOXEL_LOADS
#error "TSDF_EXP_NO_VLOAD declares the voxel words itself: build it with -DTSDF_EARLY_VOXEL_LOADS=0 (tools/ab_bound.sh does)"
#endif
#if !TSDF_GUARD_ON_RESULT && TSDF_EARLY_VOXEL_LOADS
#error "TSDF_GUARD_ON_RESULT=0 tests the numerator d0 * w0 + dn BEFORE unread distance words are rebuilt from the counts, and with TSDF_EARLY_VOXEL_LOADS those words have no initial value: build that A/B with -DTSDF_EARLY_VOXEL_LOADS=0"
#endif
#ifndef TSDF_GATHER_AUX
#define TSDF_GATHER_AUX 0  // cache policy of the ALLIN instance's frame gather (A/B: the default keeps the frame in L2)
#endif
#ifndef TSDF_WPE_MAX
#define TSDF_WPE_MAX 8
#endif
#ifndef TSDF_NO_BAND
#define TSDF_NO_BAND 0  // A/B: compile the "band seen" flag store out of k_integrate
#endif
#ifndef TSDF_BAND_PER_WAVE
#define TSDF_BAND_PER_WAVE 1  // the block's "band seen" flags are collected per WAVE (four private LDS sets) and each wave writes its own
                              // out when it is done: no barrier at the block's end.  The row loop timing itself (profiles/r06_phase_c0.txt)
                              // showed a wave spending 12-17 % of its life in that barrier, waiting for the block's slowest wave with a
                              // wave slot in hand -- an eighth of the occupancy the register budget buys.  Every writer stores the same
                              // 1, so sets that overlap (narrow grids: a cell's four rows belong to two waves) need no merging.
#endif
// TSDF_BAND_DECL declares the flag set(s) and clears them (`tid` in scope); TSDF_BAND(i) is flag cell i of the set this thread's wave
// writes (the set's address is worked out at each -- rare -- use: nothing lives in a register across the row loop for it)
#if TSDF_BAND_PER_WAVE
#define TSDF_BAND_DECL                                                                                     \
  __shared__ __attribute__((aligned(16))) uint8_t s_band_sets[4][1024];                                    \
  reinterpret_cast<u4 *>(&s_band_sets[0][0])[tid] = (u4){0u, 0u, 0u, 0u} /* 256 threads x 16 B: all four sets; the prologue's barrier follows */
#define TSDF_BAND(i) s_band_sets[threadIdx.x >> 6][i]
#define TSDF_BAND_SYNC() ((void)0)
#define TSDF_BAND_FIRST ((int)(tid & 63u))
#define TSDF_BAND_STEP 64
#else
#define TSDF_BAND_DECL                                                          \
  __shared__ __attribute__((aligned(16))) uint8_t s_band[1024 + 64];            \
  reinterpret_cast<uint32_t *>(s_band)[tid] = 0u, s_band[1024 + (tid & 63u)] = 0
#define TSDF_BAND(i) s_band[i]
#define TSDF_BAND_SYNC() __syncthreads()
#define TSDF_BAND_FIRST ((int)tid)
#define TSDF_BAND_STEP 256
#endif
#ifndef TSDF_RECOMPUTE_PX
#define TSDF_RECOMPUTE_PX 1  // 72 instead of 80 VGPRs: 7 waves per SIMD (A/B on the GPU: 17.3-17.5 against 17.8-17.9 ms)
#endif
#ifndef TSDF_WPE_PACKED
#define TSDF_WPE_PACKED 7  // waves per SIMD the non-counting ALLIN instances ask for: 72 VGPRs, reachable since the x products are redone per
                           // row (TSDF_RECOMPUTE_PX)
#endif
#ifndef TSDF_WPE_PACKED_COLOR
#define TSDF_WPE_PACKED_COLOR 8  // ... and the timed colour instance (ALLIN, PACKED, colour, certified projection): 64 VGPRs.  Round 4
                                 // measured 8 waves at 21.4 ms (spills on the row path); after round 5's diet the row body fits
                                 // (no spill reload on the row path, tools/isa_guard.py) and the eighth wave buys 1.5-2 %: 13.31 / 13.48
                                 // against 13.63 / 13.65 ms, the configs[4] slab 13.11 against 13.25 (profiles/r05_ab_eight_waves_call19.txt).
                                 // The colourless instance stays at TSDF_WPE_PACKED: it fits 8 waves as it is (49 VGPRs), and asking
                                 // for them shrinks its scalar budget (9.51 against 9.07 ms)
#endif
// waves per SIMD an instance asks for
#define TSDF_WPE_OF(ORDER, COLOR, FASTPROJ, COUNT, PACKED, ALLIN) \
  ((ALLIN) && !(COUNT) && (PACKED) && (COLOR) && (FASTPROJ) ? TSDF_WPE_PACKED_COLOR : (ALLIN) && !(COUNT) && ((PACKED) || !(COLOR)) ? TSDF_WPE_PACKED : TSDF_WPE_GENERAL)
#ifndef TSDF_WPE_GENERAL
#define TSDF_WPE_GENERAL 7  // ... and every other instance (general, row intervals, counting): 72 VGPRs + two dozen SGPRs spilled into
                            // VGPR lanes + 30-60 B of scratch, 17.7 against 18.7 ms at 2048^3 + colour (round 4).  Round 3 kept these at 6
                            // because ONE run of a 7-wave build "gave wrong voxels".  Round 4 rebuilt that configuration (and 8 waves, and
                            // SGPR spills to scratch instead of lanes), audited its ISA (spill register only ever touched by lane
                            // accesses; no VALU-writes-SGPR -> VMEM hazard) and ran the bisect script and the integrate / fused test
                            // modules on it on three GPU boxes: no voxel differs (profiles/r04_wave7_bisect.txt).  Not reproducible;
                            // tests/test_integrate_gpu.py::test_every_reachable_k_integrate_instance_equals_the_oracle now gates every
                            // instance of the shipped build
#endif
// TSDF_PHASE_TIMER (diagnostic build only, never shipped: tools/build_variant.py phase -DTSDF_PHASE_TIMER=1): a thread trace in
// software.  This image has no decoder library for rocprofv3 --att (profiles/r06_att_attempt.txt), so the row loop times
// ITSELF: every wave reads the shader clock (s_memtime) at the phase boundaries of a row and adds the differences to a dozen
// lane-private accumulators (VGPRs: opaque to the compiler, so they do not raise the scalar register pressure), one atomic add
// per accumulator and wave at the end into 1024 striped slots of g_phase; the host sums them after each launch and appends a
// line to $TSDF_HIP_PHASE_FILE.  A mark names the values its phase produced as asm inputs, so the compiler's s_waitcnt for
// them falls in front of it: the interval that ends there INCLUDES the wait.  Reading the clock costs a scalar-memory round
// trip per mark (the kernel runs ~1.6x slower): the SHARES of the phases are what this build is for, not its time.
#ifndef TSDF_PHASE_TIMER
#define TSDF_PHASE_TIMER 0
#endif
#if TSDF_PHASE_TIMER
#define TSDF_NPHASE 16
__device__ unsigned long long g_phase[1024 * TSDF_NPHASE];
static __device__ __forceinline__ uint32_t tsdf_clock_lo() {
  uint64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return (uint32_t)t;
}
#define TSDF_NACC 9  // lane-private accumulators that live across the row loop: phases 0-7 + the count of observed rows (8)
#define PT_DECL                                                   \
  uint32_t pt_acc[TSDF_NACC];                                     \
  _Pragma("unroll") for (int i_ = 0; i_ < TSDF_NACC; ++i_) asm volatile("v_mov_b32 %0, 0" : "=v"(pt_acc[i_])); \
  const unsigned pt_blk = (blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y) & 1023u; \
  uint32_t pt_prev = tsdf_clock_lo();                             \
  const uint32_t pt_t0 = pt_prev
#define PT_MARK(k)                           \
  do {                                       \
    const uint32_t pt_now = tsdf_clock_lo(); \
    pt_acc[k] += pt_now - pt_prev;           \
    pt_prev = pt_now;                        \
  } while (0)
/* outside the row loop: straight to the global slot, nothing kept in a register */                    \

#define PT_MARK_OUT(k)                                                                                 \
  do {                                                                                                 \
    const uint32_t pt_now = tsdf_clock_lo();                                                           \
    if ((threadIdx.x & 63u) == 0u) atomicAdd(&g_phase[pt_blk * TSDF_NPHASE + (k)], (unsigned long long)(pt_now - pt_prev)); \
    pt_prev = pt_now;                                                                                  \
  } while (0)
#define PT_COUNT(k) pt_acc[k] += 1u
#define PT_DEP4(a_, b_, c_, d_) asm volatile("" ::"v"(a_), "v"(b_), "v"(c_), "v"(d_) : "memory")
#define PT_DEP1(a_) asm volatile("" ::"v"(a_) : "memory")
#define PT_FLUSH()                                                                                       \
  do {                                                                                                   \
    const uint32_t pt_now = tsdf_clock_lo();                                                             \
    if ((threadIdx.x & 63u) == 0u) {                                                                     \
      atomicAdd(&g_phase[pt_blk * TSDF_NPHASE + 15], (unsigned long long)(pt_now - pt_t0));              \
      _Pragma("unroll") for (int i_ = 0; i_ < TSDF_NACC; ++i_)                                           \
          atomicAdd(&g_phase[pt_blk * TSDF_NPHASE + i_], (unsigned long long)pt_acc[i_]);                \
    }                                                                                                    \
  } while (0)
#else
#define PT_DECL
#define PT_MARK(k)
#define PT_MARK_OUT(k)
#define PT_COUNT(k)
#define PT_DEP4(a_, b_, c_, d_)
#define PT_DEP1(a_)
#define PT_FLUSH()
#endif
// ALLIN (only with FASTPROJ): the host has proved (launch_integrate, `all_inside`: the slab's eight corner voxels, a
// convex frustum) that EVERY voxel of the launch passes the sensor-range test of hpp:146 and projects inside the image
// with a pixel to spare, and nx is a multiple of 4: the per-voxel range compares, the image-bounds compares and the
// "nothing in range" exits go away -- the turntable / object-in-front-of-the-camera case the headline is quoted on.
// LIVE: the launch comes with the frame's ROW INTERVALS (k_rows below: per voxel row of the launch, the x range
// [lo, lo + len) outside of which no voxel of the row can be integrated -- conservative for updateVoxel's own tests, EXACT
// for the reference's frustum cull when it is replicated) and with k_cull's block flags: 0 = no row of the block meets
// its x range (the block leaves at once), 1 = every row's interval covers the block's whole x range (the intervals decide
// nothing in it), 2 = some do not: then a wave skips every row whose interval misses its 64 quads before any arithmetic,
// and the voxels of a quad that lie outside the interval are masked like voxels out of sensor range.
// (Never with ALLIN.  A slab wholly in view that the reference's cull cuts takes this instance for all its blocks: the
// alternative -- an ALLIN pass over the blocks flagged 1 plus an interval pass over those flagged 2 -- was built and
// measured slower, profiles/r04_refcull_dual_launch_measured_and_removed.txt.)
template <int ORDER, bool COLOR, bool FASTPROJ, bool COUNT, bool PACKED, bool ALLIN = false, bool LIVE = false>
static __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TSDF_WPE_OF(ORDER, COLOR, FASTPROJ, COUNT, PACKED, ALLIN) < TSDF_WPE_MAX ? TSDF_WPE_OF(ORDER, COLOR, FASTPROJ, COUNT, PACKED, ALLIN) : TSDF_WPE_MAX, TSDF_WPE_MAX)))
k_integrate(const IntegrateArgs a, float *__restrict__ D, float *__restrict__ Wt, uint32_t *__restrict__ RGB,
            uint8_t *__restrict__ K8, const float *__restrict__ depth, const double *__restrict__ cam,
            const float *__restrict__ ctrx, const float *__restrict__ ctry, const float *__restrict__ ctrz,
            unsigned long long *__restrict__ n_obs, const uint8_t *__restrict__ live, uint8_t *__restrict__ band,
            const uint32_t *__restrict__ row_iv) {
  // brick-level frustum cull (k_cull below): a block none of whose voxels can be observed leaves at once
  static_assert(!(ALLIN && LIVE), "the ALLIN instance knows no row intervals");
  const BlockCoords bc = tsdf_block_coords(a.zfast);
  PT_DECL;
  bool strad = false;  // LIVE: this block's rows need their intervals (block-uniform)
  if (LIVE) {
    const unsigned flag = live[bc.bx + bc.gdx * (bc.by + bc.gdy * bc.bz)];
    if (flag == 0u) return;
    strad = flag == 2u;
  }
  const unsigned tid = threadIdx.x;
#if TSDF_KTAB
  __shared__ KEntry s_tab[PACKED ? 256 : 1];  // per count k: decoded weight, Rcp32(k + 1).y, colour rounding offset, next count
#else
  __shared__ f2 s_rcp[256];  // s_rcp[k] = {Rcp32(k + 1).y, the colour average's rounding offset for that divisor (KEntry::hy)}
#endif
  // per row of this block (rpb * TY <= 256): the row's part of pcl::transformPoint -- yt[q] = cy * m[q][1] + (cz * m[q][2] + m[q][3])
  // in PCL's SSE order, m[q][1] * cy in the other -- worked out once per block by the thread of that number instead of by every
  // thread in every row (the very same operations: bit-identical), and the z part with it: no per-thread zt registers
  __shared__ f4 s_yt[256];
  constexpr bool F2 = TSDF_PROJ_F2 && ALLIN && FASTPROJ;  // transform + projection on float pairs (see the row loop)
  constexpr bool LEAN_K1 = TSDF_LEAN_K && !TSDF_KTAB && TSDF_COLOR_PK == 2 && PACKED && COLOR;  // (see the decode step)
  __shared__ f4 s_cx[F2 ? 256 : 1];  // F2: every thread's own four x centres, re-read each row (16 B of LDS instead of four registers)
  __shared__ uint32_t s_iv[LIVE ? 256 : 1];  // LIVE: the row intervals of this block's rows, lo | len << 16 (launch-relative x)
  // "band seen" flags of this block's flag cells (64 x 4 x 1 voxels: <= 64 row groups x TX / 16 cells), collected in
  // LDS by the waves that take the in-band path anyway and written out once when the block is done: the free-space
  // hot path pays nothing for them (a global byte store per in-band row cost 3-5 % of the kernel, measured)
  TSDF_BAND_DECL;
#if TSDF_KTAB
  if (PACKED) s_tab[tid] = tsdf_ktab_entry(a, tid);
#else
  if (PACKED) {
    const KEntry e = tsdf_ktab_entry(a, tid);
    s_rcp[tid] = (f2){e.y, e.hy};
  }
#endif
  {
    const int yy = (int)bc.by * a.rpb * a.TY + (int)tid;
    {
      const float cy_ = ctry[yy < a.ny ? yy : a.ny - 1], cz_ = ctrz[a.z_global0 + (int)bc.bz];
      f4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 3; ++q)
        t[q] = ORDER == TSDF_XFORM_PCL_SSE ? cy_ * a.m[4 * q + 1] + (cz_ * a.m[4 * q + 2] + a.m[4 * q + 3]) : a.m[4 * q + 1] * cy_;
      s_yt[tid] = t;
    }
    if (F2) {  // each thread parks the x centres of ITS quad (ALLIN: nx is a multiple of 4, every quad is whole)
      const int xq_ = (int)bc.bx * a.TX + (int)(tid & (unsigned)(a.TX - 1));
      if (xq_ < a.qpr) s_cx[tid] = *reinterpret_cast<const f4 *>(ctrx + xq_ * 4);
    }
    if (LIVE && strad) s_iv[tid] = yy < a.ny ? row_iv[(int64_t)bc.bz * a.ny + yy] : 0u;  // [launch plane][launch row]
  }
  // IMPLIED DISTANCES (PACKED, a.implied_d).  The flags as they stood BEFORE this launch, same cells and same order as
  // s_band.  While the flags describe the planes (tsdf_hip_volume::band_exact) and every launch since the reset used the
  // same hinge value p = pos / neg with (p*w + p)/(w + 1) == p (a.hinge_fixed; the host keeps that record,
  // tsdf_hip_volume::rest_state), a cell whose flag is 0 has only ever been observed in FREE SPACE -- every update of a
  // voxel in it was addObservation(p, ...) -- so the distance of each of its voxels follows from its count: k == 0 is the
  // reset value -1 (never observed; kmax >= 1), k > 0 is p.  Such a quad does not READ its distance words: they are
  // rebuilt from the counts, the update runs on them unchanged, and whatever it changes (a first observation: -1 -> p; a
  // first observation inside the band, which also sets the flag) is stored as ever.  A cell belongs to ONE block of a
  // launch and a voxel to one thread, so the pre-launch flag is the right one for every row of the cell whatever the other
  // waves of the block do meanwhile.  In the headline's regime this is most of the volume (free space between the camera
  // and the surface): 71 % of the observed voxels' distance words stay unread at 2048^3, 19 of the launch's 77 GB -- for
  // 3-5 % of its time (16.2-16.6 -> 15.5-15.8 ms; 12.0-12.3 -> 11.6 ms without colour; profiles/r04_ab_implied_distances.txt):
  // what binds the kernel is the row's dependent chain and the VALU, not the bytes.  The decision is per WAVE and per pass
  // (a scalar branch): a per-lane one (exec-masked load, finer: 85 % unread) needs the lane's cell index in a register the
  // kernel does not have -- it was spilled and its reload waited for every store in flight: 18.1 ms, measured.
  __shared__ uint8_t s_bin[PACKED ? 1024 : 1];
  if (PACKED && a.implied_d) tsdf_flags_before(a, bc, band, s_bin, tid);
  __syncthreads();
  // bit r: in its pass r over the block's rows this WAVE touches no flagged cell (wave-uniform: one scalar branch per row)
  const uint64_t quiet = PACKED && a.implied_d ? tsdf_quiet_passes(a, s_bin, tid) : 0ull;
  PT_MARK_OUT(9);  // block prologue: tables, row transforms, flags, barrier
  const int tx = (int)(tid & (unsigned)(a.TX - 1));
  const int ty = (int)(tid >> a.log2TX);
  const int xq = (int)bc.bx * a.TX + tx;
  const int zl = (int)bc.bz;
  const Rcp32 rneg = rcp32_prepare(a.neg);
  unsigned cnt = 0, chg = 0, imp = 0, rdb = 0;
  // wave-uniform bases
  const int row0 = (int)bc.by * a.rpb * a.TY;
  const int rows = min(a.rpb * a.TY, a.ny - row0);
  const int64_t e0 = ((int64_t)(a.zl0 + zl) * a.plane_rows + row0) * a.pitch;
  const unsigned span = (unsigned)rows * (unsigned)a.pitch;  // elements of this block's row group
  const rsrc_t rsD = make_rsrc(D + e0, span * 4u);
  const rsrc_t rsW = make_rsrc(PACKED ? D : Wt + e0, PACKED ? 0u : span * 4u);
  const rsrc_t rsC = make_rsrc(COLOR ? RGB + e0 : (uint32_t *)D, COLOR ? span * 4u : 0u);
  const rsrc_t rsK = make_rsrc(PACKED && !COLOR ? K8 + e0 : (uint8_t *)D, PACKED && !COLOR ? span : 0u);
  const unsigned npix = (unsigned)a.W * (unsigned)a.H;
  const rsrc_t rsF = make_rsrc(depth, COLOR ? a.bgra_off + npix * 4u : npix * 4u);  // [depth ... bgra]
  // ALLIN: every pixel index is valid, so the gather can address the frame by INDEX (stride 4, `idxen`): no shift per
  // pixel, and no range check is wanted (NUM_RECORDS at its maximum, whichever unit the hardware counts it in)
  const i4_rsrc rsFi = make_rsrc_2d(depth, 4u, 0xffffffffu);
  if (xq < a.qpr) {
    const int x4 = xq * 4;
    const float cz = ctrz[a.z_global0 + zl];
    const float4 cx4 = *reinterpret_cast<const float4 *>(ctrx + x4);
    const float cxs[4] = {cx4.x, cx4.y, cx4.z, cx4.w};
    // ---- pcl::transformPoint (hpp:145): the x products and the z part, once per thread -------------------
    float px_[4][3], zt[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int j = 0; j < 4; ++j) px_[j][r] = cxs[j] * a.m[4 * r];  // == m * cx bit for bit
      zt[r] = ORDER == TSDF_XFORM_PCL_SSE ? cz * a.m[4 * r + 2] + a.m[4 * r + 3] : a.m[4 * r + 2] * cz;
    }
    const unsigned voff = (unsigned)(ty * (int)a.pitch + x4) * 4u;  // byte offset inside the row group
    const unsigned row_step = (unsigned)a.TY * (unsigned)a.pitch * 4u;
#if TSDF_EARLY_VOXEL_LOADS
    bool pred_obs = true;  // TSDF_EARLY_VOXEL_LOADS == 2: this quad was observed in the previous row (the first row asks early)
#endif
    // [phase: row setup (loop control, row's transform part from LDS)]
    for (int r = 0; r < a.rpb; ++r) {
      const int y = row0 + ty + r * a.TY;
      if (y >= a.ny) break;
      unsigned iv_lo = 0u, iv_len = 0u;
      if (LIVE && strad) {  // the row's interval: a quad that misses it has nothing to do (a wave all of whose quads miss skips the row)
        const uint32_t iv = s_iv[ty + r * a.TY];
        iv_lo = iv & 0xffffu, iv_len = iv >> 16;
        if (!((unsigned)(x4 + 3) - iv_lo < iv_len + 3u)) continue;
      }
      const unsigned soff = (unsigned)r * row_step;
      const f4 ytv = s_yt[ty + r * a.TY];
      const float yt[3] = {ytv.x, ytv.y, ytv.z};
#if TSDF_RECOMPUTE_PX
      // the x products are redone per row (12 multiplies, six packed) instead of living in 12 registers across the
      // loop: the kernel waits for memory, not for the VALU, and registers are what limits the waves in flight
      float px[4][3];
      if constexpr (!F2) {
        float cxr[4] = {cxs[0], cxs[1], cxs[2], cxs[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(cxr[j]));  // (keeps LLVM from hoisting the products back out)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) px[j][q] = cxr[j] * a.m[4 * q];
      }
#else
      const float (&px)[4][3] = px_;
#endif
      auto transform = [&](int j, int q) -> float {
        if (ORDER == TSDF_XFORM_PCL_SSE) return px[j][q] + yt[q];
        return ((px[j][q] + yt[q]) + zt[q]) + a.m[4 * q + 3];
      };
      // [phase: transform + project, general instance (not in ALLIN)]
      // ---- range test (hpp:146, .cpp:616) + reprojectPoint (.cpp:611-617), voxel by voxel ----------------
      int pix[4];
      float gzs[4];
      bool any = false, lowz = false;
      if constexpr (F2) {
        asm volatile("" ::: "memory");  // (a compiler barrier, no instruction: the read below is redone each row, so neither
                                        // the centres nor their products live in registers across the loop)
        project_quad_allin<ORDER>(a, a.m, cam, s_cx[tid], ytv, zt, pix, gzs);
        PT_DEP4(pix[0], pix[1], pix[2], pix[3]);
        PT_MARK(0);  // loop control + LDS reads + transform + projection + certificate (+ exact fallback)
      } else {
      unsigned amb_mask = 0;
      uint32_t margin[4] = {0u, 0u, 0u, 0u};  // ALLIN: bits of min(fract(ru), fract(rv)) per voxel
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gx = transform(j, 0), gy = transform(j, 1), gz = transform(j, 2);
        // hpp:146 + .cpp:616: !(gz < zmin || gz > zmax) && gz > 0, as two compares: zlo is the largest float every
        // accepted gz exceeds (the float below zmin when zmin > 0, else 0; a NaN gz fails, as there)
        const bool in = ALLIN || (gz > a.zlo && !(gz > a.zmax) && (!LIVE || !strad || (unsigned)(x4 + j) - iv_lo < iv_len));
        gzs[j] = gz;
        if (!ALLIN) lowz |= in && gz < 0x1p-14f;  // (ALLIN: the host checked g.z > 1e-3 for the whole slab)
        int p;
        if (FASTPROJ) {
          bool amb;
          p = project_fast<ALLIN>(a, gx, gy, gz, amb, &margin[j]);  // garbage in, garbage out: masked by `in` below
          if (amb && in) amb_mask |= 1u << j;
        } else {
          p = project_exact(a, cam, gx, gy, in ? gz : 1.f);
        }
        pix[j] = in ? p : -1;
        any |= in;
      }
      if (!ALLIN && !any) continue;
      if (ALLIN && FASTPROJ) {  // one compare certifies the quad; which voxels failed is only worked out off the hot path
        const uint32_t hb = __float_as_uint(a.hb_max);
        if (!(min(min(margin[0], margin[1]), min(margin[2], margin[3])) > hb)) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (!(margin[j] > hb)) amb_mask |= 1u << j;
        }
      }
      // Voxels whose fp32 projection could not be certified: redo them exactly (rare: a fraction ~4*band of the voxels).
#if TSDF_AMB_UNROLL
      if (FASTPROJ && amb_mask) {  // a copy of the fp64 code per voxel of the quad, each under its own exec mask
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (amb_mask >> j & 1u) pix[j] = project_exact(a, cam, transform(j, 0), transform(j, 1), transform(j, 2));
      }
#else
      // ... one at a time through a single copy of the fp64 code
      while (FASTPROJ && amb_mask) {
        const int j = __builtin_ctz(amb_mask);
        amb_mask &= amb_mask - 1;
        float g[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
          g[q] = j == 0 ? transform(0, q) : j == 1 ? transform(1, q) : j == 2 ? transform(2, q) : transform(3, q);
        const int p = project_exact(a, cam, g[0], g[1], g[2]);
        if (j == 0) pix[0] = p;
        if (j == 1) pix[1] = p;
        if (j == 2) pix[2] = p;
        if (j == 3) pix[3] = p;
      }
#endif
      }
#if TSDF_EARLY_VOXEL_LOADS
      constexpr bool EARLY = PACKED;  // (F32W reads three planes per voxel and is the one key that feels the extra bytes: 22.3 -> 24.6 ms)
      // the voxel words of the quad are requested TOGETHER with the frame gather instead of after its result: one memory
      // latency per row instead of two in a row, at the price of reading the planes for quads none of whose voxels turns out
      // to be observed (behind the surface, no return).  Measured on top of the implied distances (s_bin), same box,
      // alternating (profiles/r04_ab_implied_distances.txt): 2048^3 + colour 15.48-15.97 -> 15.32-15.41 ms for 58.2 -> 71.1 GB
      // per launch, without colour 11.60 -> 11.29, the configs[4] slab 14.61 -> 14.58: one to three per cent for a fifth more
      // traffic, and a frame that sees little of the slab (a wall in front of the camera) would wait a voxel-plane round trip
      // for every row it has nothing to do in.  OFF.
      // (The compiler issues them IN FRONT of the gather, whose wait therefore covers them: vector memory returns in issue
      // order.  Behind it -- the gather's results first, the voxel words still in flight -- LLVM sinks them below the
      // `if (!any) continue` that follows, sched_barrier or not; the branch round the distance load is what keeps them here.)
      const bool d_read = !PACKED || !(r < 64 && (quiet >> r & 1ull));
      // (no initial value: where the distance words are not read they are either rebuilt from the counts -- a wave in which a
      // distance can move -- or never looked at; four v_mov per row for a value nobody reads were 2 % of the row's operations)
      u4 d4;
      asm volatile("" : "=v"(d4));
      u4 w4 = {0u, 0u, 0u, 0u}, c4 = {0u, 0u, 0u, 0u};
      uint32_t k4 = 0u;
      // TSDF_EARLY_VOXEL_LOADS == 2 (round 5): only the quads PREDICTED to be observed ask early -- the predictor is the quad's
      // own outcome in the previous row of the block (surfaces are coherent from row to row: the prediction fails where a
      // surface begins or ends along y), a mispredicted observed quad asks late as before, a mispredicted unobserved one has
      // read its words for nothing: the one-round-trip row without the fifth more traffic.
      // [phase: voxel loads]
      const bool ask_early = EARLY && (TSDF_EARLY_VOXEL_LOADS != 2 || pred_obs);
      if (ask_early) {
        if (d_read) d4 = bload128(rsD, voff, soff);
        if (COLOR) c4 = bload128(rsC, voff, soff);
        if (!COLOR) k4 = bload32(rsK, voff >> 2, soff >> 2);
        if (COUNT) rdb += (d_read ? 16u : 0u) + (COLOR ? 16u : 4u);  // plane bytes requested
      }
#endif
      // [phase: frame gather]
      // ---- gather the frame (L2-resident); pixel -1 is out of the descriptor's range and reads 0 ----------
      float zs[4];
      uint32_t cs[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // a valid pixel offset is < npix*4; -1 becomes 0xfffffffc, beyond the descriptor with or without the
        // colour image's scalar offset
        if (TSDF_EXP_NO_GATHER) {
          zs[j] = 1e3f + (float)(pix[j] & 1);
          cs[j] = 0x00406080u + (unsigned)pix[j];
        } else if (ALLIN) {
          zs[j] = __uint_as_float(tsdf_struct_buffer_load_u32(rsFi, pix[j], 0, 0, TSDF_GATHER_AUX));
          if (COLOR) cs[j] = tsdf_struct_buffer_load_u32(rsFi, pix[j], 0, (int)a.bgra_off, TSDF_GATHER_AUX);
        } else {
          zs[j] = __uint_as_float(bload32(rsF, (unsigned)pix[j] << 2, 0u));
          if (COLOR) cs[j] = bload32(rsF, (unsigned)pix[j] << 2, a.bgra_off);
        }
      }
      PT_MARK(1);  // early voxel loads + frame gather ISSUED
      PT_DEP4(zs[0], zs[1], zs[2], zs[3]);
      PT_MARK(2);  // ... and the gathered depths HAVE ARRIVED (the wait also covers the early voxel words: in-order return)
      // [phase: hinge / normalise (hpp:159-198)]
      // ---- hpp:152-198: NaN test, projective SDF, hinge, normalisation ----------------------------------
      // raw / neg through the scale-free ladder: a surviving raw is 0 or, because g.z >= 2^-14 (else `lowz`),
      // at least 2^-39 in magnitude (difference of two floats one of which is >= 2^-14), and at most
      // max(pos, neg); the host checks pos/neg against the window.
      float dn[4], raw[4];
      bool act[4];
      any = false;
      bool any_div = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        raw[j] = zs[j] - gzs[j];                                        // hpp:159
        // hpp:152, :193-196: pcl_isnan(pt.z) and raw < -neg both reject; gz is finite here, so a NaN depth is a NaN
        // raw, and "raw >= -neg" is false for it and for every raw below -neg: one compare for both tests
        act[j] = (ALLIN || pix[j] >= 0) && raw[j] >= -a.neg;
        dn[j] = a.pos_over_neg;                                         // hpp:189-192: raw > pos clamps
        any |= act[j];
        if (!TSDF_LEAN_BAND) any_div |= act[j] && !(raw[j] > a.pos);
      }
      if (TSDF_LEAN_BAND) {
        // "some observed voxel lies inside the truncation band" (any_div) decides nothing but which path a wave takes, so the
        // hot path asks a cheaper question first -- is the least of the four raw values (NaN ignored by v_min) at most pos? --
        // true for every quad any_div is true for, and otherwise only for a quad with a voxel BEHIND the surface next to
        // free-space ones (a surface row, where the wave takes that path anyway); the exact test runs only then
        if (__builtin_fminf(__builtin_fminf(raw[0], raw[1]), __builtin_fminf(raw[2], raw[3])) <= a.pos) {
#pragma unroll
          for (int j = 0; j < 4; ++j) any_div |= act[j] && !(raw[j] > a.pos);
        }
      }
      // [phase: leave the row if nothing is observed; late voxel loads]
      // The colour gathers are retired HERE, with the depth gathers they were issued behind (they return in order, a few
      // cycles later), not wherever their first use falls: a quad none of whose voxels is observed leaves the row without
      // ever using them, and a gather still in flight across the back edge makes the compiler guard the next row's first
      // write to its register with an s_waitcnt vmcnt -- an in-order counter, so that wait also covers the previous row's
      // voxel STORES, in every row, whether the path was taken or not (the third memory round trip of a row, found in
      // round 5: profiles/r05_isa_phase_mix.txt)
      if (COLOR) asm volatile("" ::"v"(cs[0]), "v"(cs[1]), "v"(cs[2]), "v"(cs[3]));
#if TSDF_EARLY_VOXEL_LOADS
      if (EARLY) {  // ... and so are the early voxel words (a quad that turns out unobserved never looks at them)
        if (COLOR)
          asm volatile("" ::"v"(d4), "v"(c4));
        else
          asm volatile("" ::"v"(d4), "v"(k4));
      }
      const bool asked_early = ask_early;
      pred_obs = any;
#endif
      PT_MARK(3);  // raw distances, observed / in-band tests
      if (!any) continue;
      PT_COUNT(8);
#if TSDF_EARLY_VOXEL_LOADS == 2
      if (EARLY && !asked_early) {  // an observed quad the predictor missed: its words are requested now
        if (d_read) d4 = bload128(rsD, voff, soff);
        if (COLOR) c4 = bload128(rsC, voff, soff);
        if (!COLOR) k4 = bload32(rsK, voff >> 2, soff >> 2);
        if (COUNT) rdb += (d_read ? 16u : 0u) + (COLOR ? 16u : 4u);
      }
#endif
      // [phase: normalise: raw / neg ladder (rows with an in-band voxel)]
      if (any_div) {  // free space (every observed voxel of the wave beyond the hinge) skips all four ladders
#pragma unroll
        for (int j = 0; j < 4; ++j) dn[j] = raw[j] > a.pos ? a.pos_over_neg : div32_fast(raw[j], rneg);  // hpp:198
      }
      // [phase: IEEE fallback (rare)]
      if (lowz || !a.neg_in_window) {  // operands outside the scale-free window: the compiler's IEEE division
        // An fdiv is ONE cheap-looking IR instruction, so LLVM would if-convert this rare block into the hot path
        // (and the backend then expands every division into ~10 VALU ops there); the empty volatile asm keeps
        // the block from being speculated.
        asm volatile("");
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (act[j] && !(raw[j] > a.pos)) dn[j] = raw[j] / a.neg;
      }
      // [phase: voxel loads]
      // ---- read-modify-write -----------------------------------------------------------------------------
#if TSDF_EXP_NO_VLOAD
      const uint32_t hb_ = __float_as_uint(a.pos_over_neg);
      const u4 d4 = {hb_, hb_, hb_, hb_};
      u4 w4 = {0u, 0u, 0u, 0u}, c4 = {(unsigned)r << 24, (unsigned)r << 24, (unsigned)r << 24, (unsigned)r << 24};
      uint32_t k4 = 0u;
      const bool d_read = true;
#else
#if !TSDF_EARLY_VOXEL_LOADS
      // (PACKED) the distance words are only read where the cell's flag says they cannot be told from the counts
      constexpr bool EARLY = false;
      const bool d_read = !PACKED || !(r < 64 && (quiet >> r & 1ull));
      u4 d4 = {0u, 0u, 0u, 0u};
      u4 w4 = {0u, 0u, 0u, 0u}, c4 = {0u, 0u, 0u, 0u};
      uint32_t k4 = 0u;
#endif
      if (!EARLY) {
        if (d_read) d4 = bload128(rsD, voff, soff);
        if (!PACKED) w4 = bload128(rsW, voff, soff);
        if (COLOR) c4 = bload128(rsC, voff, soff);
        if (PACKED && !COLOR) k4 = bload32(rsK, voff >> 2, soff >> 2);
        if (COUNT) rdb += (d_read ? 16u : 0u) + (!PACKED ? 16u : 0u) + (COLOR ? 16u : 0u) + (PACKED && !COLOR ? 4u : 0u);
      }
#endif
      PT_MARK(4);  // normalise ladder (in-band rows), late loads issued
      if (COLOR) PT_DEP4(c4.x, c4.y, c4.z, c4.w); else PT_DEP1(k4);
      PT_MARK(5);  // the voxel words HAVE ARRIVED (late askers wait here; early askers' words came with the gather)
      // [phase: decode count / weight (PACKED)]
      uint32_t d0u[4] = {d4.x, d4.y, d4.z, d4.w};
      const uint32_t c0[4] = {c4.x, c4.y, c4.z, c4.w};
      const uint32_t w0u[4] = {w4.x, w4.y, w4.z, w4.w};
      float d0[4], w0[4];
      uint32_t kw[4];  // PACKED: the count in byte 3 of a word (colour word, or the k8 byte moved there)
      float dv[4], wv[4];
      uint32_t cv[4], k1[4];
#if TSDF_KTAB
      float ky[4];
#endif
      float khy[4] = {0.f, 0.f, 0.f, 0.f};  // PACKED: the colour rounding offset of each voxel's divisor (with Rcp32(k + 1).y from LDS)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d0[j] = __uint_as_float(d0u[j]);
        w0[j] = __uint_as_float(w0u[j]);
        kw[j] = 0u;
        k1[j] = 0u;
        if (PACKED) {
          kw[j] = COLOR ? c0[j] : (k4 << (24 - 8 * j));
#if TSDF_KTAB
          const KEntry e = s_tab[kw[j] >> 24];  // one ds_read_b128
          w0[j] = e.w, ky[j] = e.y, khy[j] = e.hy, k1[j] = e.k1;
#else
          // tsdf_decode_w (neither is NaN here).  With an integer max_weight the min is the identity -- and the ALLIN
          // instance is only launched with one, and with kmax == max_weight (launch_integrate) -- but leaving it out lets
          // LLVM turn the colour sums into integer multiplies and byte shuffles (+70 instructions, measured): TSDF_LEAN_K
          // hides the value behind an empty asm instead of paying a v_min per voxel for that.
          if (TSDF_LEAN_K && ALLIN) {
            w0[j] = (float)(kw[j] >> 24);
            if (COLOR) asm volatile("" : "+v"(w0[j]));  // (without colour only the rare distance ladder reads it)
          } else {
            w0[j] = __builtin_fminf((float)(kw[j] >> 24), a.wmax);
          }
          // the count after this observation, k' = min(k + 1, kmax), as byte 3 of a word (saturate BEFORE adding: with
          // kmax == 255 an in-place add would wrap byte 3 to zero; kcap = (kmax - 1) << 24 and kinc = 1 << 24, both 0
          // when kmax == 0).  LEAN_K1: v_cvt_pk_u8_f32 rewrites bytes 0-2 of the word it is handed, so there the old
          // colour bytes may ride along under the clamp (kcap | 0xffffff) and the mask goes away
          k1[j] = LEAN_K1 ? min(kw[j], a.kcap | 0xffffffu) + a.kinc : min(kw[j] & 0xff000000u, a.kcap) + a.kinc;  // == min(k + 1, kmax) << 24 (| old rgb)
#endif
        }
      }
      // F32W: the fast update is exact if update_is_safe().  PACKED: the divisor k + 1 is an integer in
      // [1, 256] (unless the weight sits at a non-integer max_weight), for which the scale-free ladder is
      // exact whenever its RESULT is a normal number (residuals of a normal numerator against an integer
      // divisor are multiples of 2^-149, so they are exact; only a subnormal quotient can double-round); a
      // result that is zero, subnormal or non-finite sends the quad to the IEEE path instead.
      // [phase: band / implied-distance flags, hinge rest test]
      bool safe = true;
      if (!PACKED) {
#pragma unroll
        for (int j = 0; j < 4; ++j) safe &= !act[j] || update_is_safe(d0[j], w0[j], dn[j]);
      }
#if !TSDF_GUARD_ON_RESULT
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          safe &= !act[j] || ((a.wmax_is_int || kw[j] < (a.kmax << 24)) && numerator_ok(d0[j] * w0[j] + dn[j]));
      }
#endif
      // PACKED, free space: a wave none of whose observed voxels is inside the truncation band (no `any_div`) and all of
      // whose observed voxels already sit at the hinge value p sees (p*w + p)/(w + 1), which the host has checked to
      // be p for every weight (a.hinge_fixed): d keeps its bits and the four ladders and their guards are skipped.
      bool d_moves = true;
#if TSDF_SKIP_FIXED_HINGE
      if (PACKED && a.hinge_fixed) {
        bool off_hinge = any_div;
        // (ALLIN: without the `observed` mask -- a quad with an unobserved voxel off the hinge just takes the general
        // path, which is always right)
        if (d_read) {
#pragma unroll
          for (int j = 0; j < 4; ++j) off_hinge |= (ALLIN || act[j]) && d0u[j] != __float_as_uint(a.pos_over_neg);
        } else {  // distances not read (see s_bin): off the hinge value <=> never observed <=> count 0
#pragma unroll
          for (int j = 0; j < 4; ++j) off_hinge |= (ALLIN || act[j]) && kw[j] < 0x01000000u;
          // (TSDF_SWAR_K, ALLIN without colour: the same question of the four count bytes at once -- is one of them zero?)
          if (TSDF_SWAR_K && ALLIN && !COLOR) off_hinge = any_div || ((k4 - 0x01010101u) & ~k4 & 0x80808080u) != 0u;
        }
        d_moves = __builtin_amdgcn_ballot_w64(off_hinge) != 0ull;  // wave-uniform: one scalar branch
      }
#endif
      if (PACKED && !d_read && d_moves) {
        // ... and only a wave in which a distance can move needs them at all: never observed = the reset value, else the
        // hinge value
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          d0u[j] = kw[j] < 0x01000000u ? 0xbf800000u : __float_as_uint(a.pos_over_neg);
          d0[j] = __uint_as_float(d0u[j]);
        }
      }
      bool d_touched = false;  // this lane's distances went through an update (else dv == d0 and nothing is compared or stored)
      // [phase: colour update (octree.cpp:328-337)]
      if (safe) {
        Rcp32 rs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dv[j] = d0[j], wv[j] = w0[j], cv[j] = c0[j];
        // PACKED without colour: the reciprocal of k + 1 is only wanted where a distance moves (wave-uniform: d_moves is a ballot);
        // in resting free space the count bytes are all there is to update
        constexpr bool QUAD_COLOR = TSDF_COLOR_F2 && COLOR && PACKED && TSDF_COLOR_PK == 2 && !TSDF_KTAB;  // colour_quad_pk
        if constexpr (QUAD_COLOR) {
#pragma unroll
          for (int j = 0; j < 4; j += 2) {  // voxels (0, 1), then (2, 3)
            const f2 yh0 = s_rcp[kw[j] >> 24], yh1 = s_rcp[kw[j + 1] >> 24];  // w0 + 1 == k + 1 here (w0 is an integer)
            rs[j].nb = -(w0[j] + 1.f), rs[j + 1].nb = -(w0[j + 1] + 1.f);
            rs[j].y = yh0.x, rs[j + 1].y = yh1.x;
            colour_pair_pk((f2){w0[j], w0[j + 1]}, c0[j], c0[j + 1], cs[j], cs[j + 1], (f2){yh0.x, yh1.x}, (f2){yh0.y, yh1.y}, k1[j], k1[j + 1],
                           cv[j], cv[j + 1]);
          }
        } else
        if (COLOR || !PACKED || d_moves)
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // colour and weight: every observed voxel
          if (PACKED) {
            rs[j].nb = -(w0[j] + 1.f);
#if TSDF_KTAB
            rs[j].y = ky[j];  // w0 + 1 == k + 1 here (w0 is an integer)
#else
            const f2 yh = s_rcp[kw[j] >> 24];  // w0 + 1 == k + 1 here (w0 is an integer)
            rs[j].y = yh.x, khy[j] = yh.y;
#endif
          } else {
            rs[j] = rcp32_prepare(w0[j] + 1.f);
          }
          add_observation_fast<COLOR, false>(dv[j], wv[j], cv[j], dn[j], cs[j], a.wmax, rs[j], COLOR ? k1[j] : 0u, PACKED ? &khy[j] : nullptr);
        }
        // [phase: d update (octree.cpp:152-163)]
        if (d_moves) {  // distance
          d_touched = true;
          // a voxel of this quad was observed inside the truncation band: its distance may turn negative, which is what
          // marching cubes looks for (see s_band)
          // (the host only passes `band` when every block's first row is a multiple of 4: local row groups == global ones)
          if (!TSDF_NO_BAND && any_div) TSDF_BAND(((ty + r * a.TY) >> 2) * max(1, a.TX >> 4) + (tx >> 4)) = 1;
#pragma unroll
          for (int j = 0; j < 4; ++j) dv[j] = div32_fast(d0[j] * w0[j] + dn[j], rs[j]);
#if TSDF_GUARD_ON_RESULT
          if (PACKED) {
            // the guard, on the RESULT (one v_cmp_class per voxel): a normal quotient is the correctly rounded one
            // (tests/test_div_gpu.py::test_count_divider_*); zero, subnormal, infinite or NaN -- or a weight sitting
            // at a non-integer max_weight, whose divisor is no count -- sends the quad through the IEEE path below
#pragma unroll
            for (int j = 0; j < 4; ++j)
              safe &= !act[j] || ((ALLIN || a.wmax_is_int || kw[j] < (a.kmax << 24)) && __builtin_amdgcn_classf(dv[j], 0x108));
          }
#endif
        }
      }
      // [phase: IEEE fallback (rare)]
      if (!safe) {
        d_touched = true;
        if (!TSDF_NO_BAND && any_div) TSDF_BAND(((ty + r * a.TY) >> 2) * max(1, a.TX >> 4) + (tx >> 4)) = 1;
        asm volatile("");  // rare: keep the 16 IEEE divisions out of the hot path's schedule
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dv[j] = d0[j];
          wv[j] = w0[j];
          cv[j] = c0[j] & 0xffffffu;
          add_observation_ieee<COLOR>(dv[j], wv[j], cv[j], dn[j], cs[j], a.wmax);
          if (COLOR) cv[j] |= k1[j] & 0xff000000u;  // both flavours return the colour with the new count in byte 3
        }
      }
      PT_DEP4(dv[0], dv[1], cv[0], cv[1]);
      PT_MARK(6);  // decode, flags, hinge rest test, colour + distance update
      // [phase: select / change detection / store]
      uint32_t diff_w = 0u, diff_c = 0u, k4n = 0u;
      uint32_t wn_u[4];
      const uint32_t c_before[4] = {c0[0], c0[1], c0[2], c0[3]};
      if (d_touched) {  // (free space resting at the hinge never gets here: wave-uniform skip)
        uint32_t diff_d = 0u, dn_u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dn_u[j] = act[j] ? __float_as_uint(dv[j]) : d0u[j];
          diff_d |= dn_u[j] ^ d0u[j];
          if (COUNT) chg += dn_u[j] != d0u[j] ? 4u : 0u;
        }
        if (diff_d && !TSDF_EXP_NO_STORE) bstore128(rsD, voff, soff, (u4){dn_u[0], dn_u[1], dn_u[2], dn_u[3]});
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (PACKED && !COLOR && !TSDF_SWAR_K) k4n |= (act[j] ? k1[j] : (kw[j] & 0xff000000u)) >> (24 - 8 * j);
        wn_u[j] = act[j] ? __float_as_uint(wv[j]) : w0u[j];
        cv[j] = act[j] ? cv[j] : c0[j];
        diff_w |= wn_u[j] ^ w0u[j];
        diff_c |= cv[j] ^ c0[j];
        cnt += act[j] ? 1u : 0u;
        if (COUNT && PACKED) imp += act[j] && !d_read ? 1u : 0u;  // observed voxels whose distance word was not read
        if (COUNT)  // bytes of voxel words whose VALUE changed: what any layout-preserving kernel has to write
          chg += (!PACKED && wn_u[j] != w0u[j] ? 4u : 0u) + (COLOR && cv[j] != c_before[j] ? 4u : 0u);
      }
      if (PACKED && !COLOR && TSDF_SWAR_K) {
        // The four counts of the quad in one word (TSDF_SWAR_K): k' = k + (observed && k < kmax) per byte, == min(k + 1, kmax)
        // for every count the layout can hold (k <= kmax).  "k < kmax" for all four at once: the even and the odd bytes sit
        // in 16-bit fields with bit 8 set on top; after subtracting kmax from each field that bit survives exactly where
        // k >= kmax (256 + k - kmax stays within the field: no borrow crosses it).  ~13 operations instead of 28 per quad.
        const uint32_t km2 = a.kmax * 0x00010001u;
        const uint32_t te = ((k4 & 0x00ff00ffu) | 0x01000100u) - km2, to = (((k4 >> 8) & 0x00ff00ffu) | 0x01000100u) - km2;
        const uint32_t lt = ((~te >> 8) & 0x00010001u) | (((~to >> 8) & 0x00010001u) << 8);  // 0x01 in byte j: k_j < kmax
        const uint32_t actb = (act[0] ? 0x00000001u : 0u) | (act[1] ? 0x00000100u : 0u) | (act[2] ? 0x00010000u : 0u) | (act[3] ? 0x01000000u : 0u);
        k4n = k4 + (lt & actb);
      }
      if (COUNT && PACKED && !COLOR) chg += (unsigned)__popc(((k4n ^ k4) | ((k4n ^ k4) >> 1) | ((k4n ^ k4) >> 2) | ((k4n ^ k4) >> 3) |
                                                            ((k4n ^ k4) >> 4) | ((k4n ^ k4) >> 5) | ((k4n ^ k4) >> 6) | ((k4n ^ k4) >> 7)) & 0x01010101u);
      if (!PACKED && diff_w) bstore128(rsW, voff, soff, (u4){wn_u[0], wn_u[1], wn_u[2], wn_u[3]});
      if (TSDF_EXP_NO_STORE) {  // keep the results alive without touching memory
        if ((cv[0] ^ cv[1] ^ cv[2] ^ cv[3]) == 0x12345678u) bstore32(rsD, voff, soff, cv[0] ^ diff_w);
      } else
      if (COLOR && diff_c) bstore128(rsC, voff, soff, (u4){cv[0], cv[1], cv[2], cv[3]});
      if (PACKED && !COLOR && k4n != k4) bstore32(rsK, voff >> 2, soff >> 2, k4n);
      PT_MARK(7);  // select, change detection, stores issued
    }
  }
  PT_MARK_OUT(10);  // what is left of the loop (rows that left early are in 0-3), exit
  if (!TSDF_NO_BAND && band) {  // the block's flags -> the volume's flag array (every writer stores the same 1: no atomics)
    TSDF_BAND_SYNC();  // (per-wave sets: none -- a wave reads back what it wrote itself)
    const int lf = max(0, a.log2TX - 4), fxb = 1 << lf;
    const int yg0 = (a.y_abs0 + row0) >> 2, yg1 = (a.y_abs0 + row0 + max(rows, 1) - 1) >> 2;
    const int xc0 = (a.x_abs0 + (int)bc.bx * a.TX * 4) >> 6;
    const int n_fl = (yg1 - yg0 + 1) << lf;
    for (int i = TSDF_BAND_FIRST; i < n_fl; i += TSDF_BAND_STEP) {
      const int yg = yg0 + (i >> lf), xc = xc0 + (i & (fxb - 1));
      if (TSDF_BAND(i) && yg < a.band_fy && xc < a.band_fx) band[((int64_t)(a.zl0 + zl) * a.band_fy + yg) * a.band_fx + xc] = 1;
    }
  }
  PT_MARK_OUT(12);  // epilogue: barrier + flag write-out
  PT_FLUSH();
  if (COUNT) {  // block reduction, then one of 1024 striped counters (summed by the host); slots 1024.. = changed bytes,
                // slots 2048.. = observed voxels whose distance word was not read
    __shared__ unsigned s_cnt, s_chg, s_imp, s_rdb;
    if (tid == 0) s_cnt = s_chg = s_imp = s_rdb = 0;
    __syncthreads();
    if (cnt) atomicAdd(&s_cnt, cnt);
    if (chg) atomicAdd(&s_chg, chg);
    if (imp) atomicAdd(&s_imp, imp);
    if (rdb) atomicAdd(&s_rdb, rdb);
    __syncthreads();
    if (tid == 0 && (s_cnt || s_rdb)) {  // (slots 3072..: bytes of the voxel planes the launch requested, observed voxel or not)
      const unsigned b = bc.bx + bc.by * bc.gdx + bc.bz * bc.gdx * bc.gdy;
      if (s_cnt) atomicAdd(n_obs + (b & 1023u), (unsigned long long)s_cnt);
      if (s_chg) atomicAdd(n_obs + 1024u + (b & 1023u), (unsigned long long)s_chg);
      if (s_imp) atomicAdd(n_obs + 2048u + (b & 1023u), (unsigned long long)s_imp);
      if (s_rdb) atomicAdd(n_obs + 3072u + (b & 1023u), (unsigned long long)s_rdb);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_integrate_p (round 6): the ALLIN PACKED instance WITHOUT colour -- BASELINE configs[1] / [2], the reference's default
// (integrate_color_(false), src/lib/tsdf_volume_octree.cpp:78) -- as a two-stage SOFTWARE PIPELINE over the block's rows.
//
// What bound k_integrate's colourless instance (profiles/r06_phase_c0.txt, r06_c0_summary_pmc_*.json):
// a wave's row is ONE dependent chain -- LDS reads -> transform / projection / certificate -> frame gather (an L2 round
// trip) -> observed / hinge tests -> count update -> store -- run back to back; eight waves per SIMD take turns on the
// VALU, and whenever most of them sit in the gather's s_waitcnt at once the SIMD idles: ~85 % VALU-active, with
// neither fewer operations nor fewer bytes moving the time (round 5's four A/Bs).  Here a wave keeps TWO rows in flight:
//
//   stage A(r)  "issue":   LDS reads, pcl::transformPoint + certified projection of row r (hpp:143-149), then the row's
//                          loads all at once: [distance words] + count bytes + four depth gathers
//   stage B(r)  "consume": hpp:152-198 on the gathered depths, addObservation (octree.cpp:152-163), stores
//
//   A(0);  loop:  A(r+1)  B(r)  A(r+2)  B(r+1)  ...          (two row-register sets taken in turn: no register copies)
//
// B(r)'s wait for its loads is `s_waitcnt vmcnt(N)` with N = the loads A(r+1) has just issued BEHIND them (the counter
// is in order), so a row's gather has a whole A + B of other work to come back in and the wave hardly ever stalls on it.
// For the compiler to count N the steady-state loop issues a STATIC number of loads per stage: no predicted / late voxel
// loads here -- the count bytes are requested for every quad (1 B per voxel: unobserved quads cost 2 % more traffic) and
// the distance words per WAVE and pass (the implied-distance decision of k_integrate, a scalar branch whose two paths
// differ by one load: the wait is conservative by that one load at most).
// (Tried on top, round 6: stage A requesting the PLANE words BEFORE it projects -- they come from HBM, the gathers from L2, and
// depend on nothing the projection works out.  No effect, here or in k_integrate_pc: 7.47 against 7.47 ms, 13.16 against 13.20
// (profiles/r06_ab_planefirst_call10.txt): the stage the pipeline gives every load already covers them.  Not kept.)
// Same arithmetic as k_integrate<ORDER, false, true, COUNT, true, true, false>, operation for operation; the host
// launches it instead of that instance when, additionally, every row step of a block is whole (ny a multiple of TY),
// the hinge value rests (hinge_fixed) and max_dist_neg lies in the scale-free divider's window -- else the old instance.
#ifndef TSDF_WPE_PIPE
#define TSDF_WPE_PIPE 8  // 63 VGPRs, no scratch.  (LLVM grants 8 waves 80 SGPRs, 7 waves 96+: at 8 the loop's rare blocks carry scalar spills
                         // in VGPR lanes, at 7 -- 65 VGPRs, seven waves -- hardly any; measured by alternation, 2048^3 without colour:
                         // 7.61 ms at 8 against 7.81 at 7, 8.98 for k_integrate's own row loop: profiles/r06_ab_pipe_c0_call02.txt)
#endif
struct PipeRow {      // what stage A hands to stage B
  float gz[4];        // g.z of the quad's four voxels
  uint32_t z[4];      // gathered depths (bits; in flight)
  uint32_t k4;        // the quad's four count bytes (in flight)
  u4 d4;              // its distance words, where the pass reads them (in flight)
};

template <int ORDER, bool COUNT>
static __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TSDF_WPE_PIPE < TSDF_WPE_MAX ? TSDF_WPE_PIPE : TSDF_WPE_MAX, TSDF_WPE_MAX)))
k_integrate_p(const IntegrateArgs a, float *__restrict__ D, uint8_t *__restrict__ K8, const float *__restrict__ depth,
              const double *__restrict__ cam, const float *__restrict__ ctrx, const float *__restrict__ ctry,
              const float *__restrict__ ctrz, unsigned long long *__restrict__ n_obs, uint8_t *__restrict__ band) {
  const BlockCoords bc = tsdf_block_coords(a.zfast);
  const unsigned tid = threadIdx.x;
  __shared__ f2 s_rcp[256];  // s_rcp[k] = {Rcp32(k + 1).y, unused here}
  __shared__ f4 s_yt[256];   // the rows' part of pcl::transformPoint (see k_integrate)
  __shared__ f4 s_cx[256];   // every thread's own four x centres
  __shared__ uint8_t s_bin[1024];
  TSDF_BAND_DECL;
  {
    const KEntry e = tsdf_ktab_entry(a, tid);
    s_rcp[tid] = (f2){e.y, e.hy};
    const int yy = (int)bc.by * a.rpb * a.TY + (int)tid;
    const float cy_ = ctry[yy < a.ny ? yy : a.ny - 1], cz_ = ctrz[a.z_global0 + (int)bc.bz];
    f4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 3; ++q)
      t[q] = ORDER == TSDF_XFORM_PCL_SSE ? cy_ * a.m[4 * q + 1] + (cz_ * a.m[4 * q + 2] + a.m[4 * q + 3]) : a.m[4 * q + 1] * cy_;
    s_yt[tid] = t;
    const int xq_ = (int)bc.bx * a.TX + (int)(tid & (unsigned)(a.TX - 1));
    if (xq_ < a.qpr) s_cx[tid] = *reinterpret_cast<const f4 *>(ctrx + xq_ * 4);
  }
  if (a.implied_d) tsdf_flags_before(a, bc, band, s_bin, tid);
  __syncthreads();
  const uint64_t quiet = a.implied_d ? tsdf_quiet_passes(a, s_bin, tid) : 0ull;
  const int tx = (int)(tid & (unsigned)(a.TX - 1));
  const int ty = (int)(tid >> a.log2TX);
  const int xq = (int)bc.bx * a.TX + tx;
  const int zl = (int)bc.bz;
  const Rcp32 rneg = rcp32_prepare(a.neg);
  unsigned cnt = 0, chg = 0, imp = 0, rdb = 0;
  const int row0 = (int)bc.by * a.rpb * a.TY;
  const int rows = min(a.rpb * a.TY, a.ny - row0);  // a multiple of TY (the host checked)
  const int64_t e0 = ((int64_t)(a.zl0 + zl) * a.plane_rows + row0) * a.pitch;
  const unsigned span = (unsigned)rows * (unsigned)a.pitch;
  const rsrc_t rsD = make_rsrc(D + e0, span * 4u);
  const rsrc_t rsK = make_rsrc(K8 + e0, span);
  const i4_rsrc rsFi = make_rsrc_2d(depth, 4u, 0xffffffffu);
  const uint32_t pbits = __float_as_uint(a.pos_over_neg);
  if (xq < a.qpr) {
    const int x4 = xq * 4;
    const float cz = ctrz[a.z_global0 + zl];
    float zt[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) zt[q] = ORDER == TSDF_XFORM_PCL_SSE ? cz * a.m[4 * q + 2] + a.m[4 * q + 3] : a.m[4 * q + 2] * cz;
    const unsigned voff = (unsigned)(ty * (int)a.pitch + x4) * 4u;
    const unsigned row_step = (unsigned)a.TY * (unsigned)a.pitch * 4u;

    // ---- stage A: transform + certified projection (project_quad_allin), then every load of the row ----------------
    auto issue = [&](const int r, PipeRow &R) {
      const f4 ytv = s_yt[ty + r * a.TY];
      asm volatile("" ::: "memory");  // (the centres are re-read each row: neither they nor their products live across the loop)
      int pix[4];
      project_quad_allin<ORDER>(a, a.m, cam, s_cx[tid], ytv, zt, pix, R.gz);
      const unsigned soff = (unsigned)r * row_step;
      const bool d_read = !(r < 64 && (quiet >> r & 1ull));  // wave-uniform (see k_integrate: implied distances)
      // [phase: voxel loads]
      if (d_read)
        R.d4 = bload128(rsD, voff, soff);
      else
        asm volatile("" : "=v"(R.d4));  // (no value: rebuilt from the counts where a distance can move, never looked at otherwise)
      R.k4 = bload32(rsK, voff >> 2, soff >> 2);
      // [phase: frame gather]
#pragma unroll
      for (int j = 0; j < 4; ++j) R.z[j] = tsdf_struct_buffer_load_u32(rsFi, pix[j], 0, 0, TSDF_GATHER_AUX);
      if (COUNT) rdb += (d_read ? 16u : 0u) + 4u;
    };

    // ---- stage B: hpp:152-198 + addObservation on what stage A requested -------------------------------------------
    auto consume = [&](const int r, const PipeRow &R) {
      const unsigned soff = (unsigned)r * row_step;
      const bool d_read = !(r < 64 && (quiet >> r & 1ull));
      // [phase: hinge / normalise (hpp:159-198)]
      float raw[4];
      bool act[4];
      bool any = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        raw[j] = __uint_as_float(R.z[j]) - R.gz[j];  // hpp:159
        act[j] = raw[j] >= -a.neg;                   // hpp:152 (a NaN depth is a NaN raw) and :193-196 in one compare
        any |= act[j];
      }
      if (__builtin_amdgcn_ballot_w64(any) == 0ull) return;  // nothing of this wave's row is observed
      bool any_div = false;  // an observed voxel inside the truncation band (k_integrate: TSDF_LEAN_BAND)
      if (__builtin_fminf(__builtin_fminf(raw[0], raw[1]), __builtin_fminf(raw[2], raw[3])) <= a.pos) {
#pragma unroll
        for (int j = 0; j < 4; ++j) any_div |= act[j] && !(raw[j] > a.pos);
      }
      const uint32_t k4 = R.k4;
      uint32_t d0u[4] = {R.d4.x, R.d4.y, R.d4.z, R.d4.w};
      // [phase: band / implied-distance flags, hinge rest test]
      // free space rests: a wave none of whose observed quads holds an in-band voxel or a voxel off the hinge value p sees
      // (p*w + p)/(w + 1) == p (a.hinge_fixed, host-checked for every weight): the counts are all there is to update
      bool off_hinge = any_div;
      if (d_read) {
#pragma unroll
        for (int j = 0; j < 4; ++j) off_hinge |= d0u[j] != pbits;
      } else {  // distances not read: off the hinge value <=> never observed <=> count 0: is one of the four bytes zero?
        off_hinge |= ((k4 - 0x01010101u) & ~k4 & 0x80808080u) != 0u;
      }
      const bool d_moves = __builtin_amdgcn_ballot_w64(any && off_hinge) != 0ull;  // wave-uniform
      // [phase: select / change detection / store]
      // the four counts in one word (k_integrate: TSDF_SWAR_K): k' = k + (observed && k < kmax) per byte
      const uint32_t km2 = a.kmax * 0x00010001u;
      const uint32_t te = ((k4 & 0x00ff00ffu) | 0x01000100u) - km2, to = (((k4 >> 8) & 0x00ff00ffu) | 0x01000100u) - km2;
      const uint32_t lt = ((~te >> 8) & 0x00010001u) | (((~to >> 8) & 0x00010001u) << 8);  // 0x01 in byte j: k_j < kmax
      const uint32_t actb = (act[0] ? 0x00000001u : 0u) | (act[1] ? 0x00000100u : 0u) | (act[2] ? 0x00010000u : 0u) | (act[3] ? 0x01000000u : 0u);
      const uint32_t k4n = k4 + (lt & actb);
      if (d_moves) {  // (a tenth of the observed wave-rows: surfaces, first observations)
        // [phase: normalise: raw / neg ladder (rows with an in-band voxel)]
        float dn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dn[j] = a.pos_over_neg;  // hpp:189-192
        if (any_div) {
#pragma unroll
          for (int j = 0; j < 4; ++j) dn[j] = raw[j] > a.pos ? a.pos_over_neg : div32_fast(raw[j], rneg);  // hpp:198
        }
        // [phase: decode count / weight (PACKED)]
        float d0[4], w0[4], dv[4];
        uint32_t kw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          kw[j] = k4 << (24 - 8 * j);
          w0[j] = (float)(kw[j] >> 24);  // tsdf_decode_w: max_weight is an integer equal to kmax here (ALLIN), so min() is the identity
          if (!d_read) d0u[j] = kw[j] < 0x01000000u ? 0xbf800000u : pbits;  // never observed: the reset value; else the hinge value
          d0[j] = __uint_as_float(d0u[j]);
        }
        // [phase: d update (octree.cpp:152-163)]
        if (any_div) TSDF_BAND(((ty + r * a.TY) >> 2) * max(1, a.TX >> 4) + (tx >> 4)) = 1;
        bool safe = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Rcp32 rs;
          rs.nb = -(w0[j] + 1.f);
          rs.y = s_rcp[kw[j] >> 24].x;
          dv[j] = div32_fast(d0[j] * w0[j] + dn[j], rs);
          // the guard on the RESULT (k_integrate: TSDF_GUARD_ON_RESULT): a normal quotient is the correctly rounded one
          safe &= !act[j] || __builtin_amdgcn_classf(dv[j], 0x108);
        }
        // [phase: IEEE fallback (rare)]
        if (!safe) {
          asm volatile("");  // rare: keep the IEEE divisions out of the hot path's schedule
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float wv = w0[j];
            uint32_t cv = 0u;
            dv[j] = d0[j];
            add_observation_ieee<false>(dv[j], wv, cv, dn[j], 0u, a.wmax);
          }
        }
        uint32_t diff_d = 0u, dn_u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dn_u[j] = act[j] ? __float_as_uint(dv[j]) : d0u[j];
          diff_d |= dn_u[j] ^ d0u[j];
          if (COUNT) chg += dn_u[j] != d0u[j] ? 4u : 0u;
        }
        if (diff_d) bstore128(rsD, voff, soff, (u4){dn_u[0], dn_u[1], dn_u[2], dn_u[3]});
      }
      if (COUNT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          cnt += act[j] ? 1u : 0u;
          imp += act[j] && !d_read ? 1u : 0u;
        }
        const uint32_t x = k4n ^ k4;
        chg += (unsigned)__popc((x | (x >> 1) | (x >> 2) | (x >> 3) | (x >> 4) | (x >> 5) | (x >> 6) | (x >> 7)) & 0x01010101u);
      }
      if (k4n != k4) bstore32(rsK, voff >> 2, soff >> 2, k4n);
    };

    PipeRow A, B;
    const int nr = rows / a.TY;  // (the grid's last block may be shorter: the host checked that ny is a multiple of TY)
    issue(0, A);
    int r = 0;
    for (; r + 2 < nr; r += 2) {  // steady state: every trip issues two rows and consumes two
      issue(r + 1, B);
      consume(r, A);
      issue(r + 2, A);
      consume(r + 1, B);
    }
    if (r + 1 < nr) {
      issue(r + 1, B);
      consume(r, A);
      consume(r + 1, B);
    } else {
      consume(r, A);
    }
  }
  if (band) {  // the block's flags -> the volume's flag array (every writer stores the same 1: no atomics)
    TSDF_BAND_SYNC();  // (per-wave sets: none -- a wave reads back what it wrote itself)
    const int lf = max(0, a.log2TX - 4), fxb = 1 << lf;
    const int yg0 = (a.y_abs0 + row0) >> 2, yg1 = (a.y_abs0 + row0 + max(rows, 1) - 1) >> 2;
    const int xc0 = (a.x_abs0 + (int)bc.bx * a.TX * 4) >> 6;
    const int n_fl = (yg1 - yg0 + 1) << lf;
    for (int i = TSDF_BAND_FIRST; i < n_fl; i += TSDF_BAND_STEP) {
      const int yg = yg0 + (i >> lf), xc = xc0 + (i & (fxb - 1));
      if (TSDF_BAND(i) && yg < a.band_fy && xc < a.band_fx) band[((int64_t)(a.zl0 + zl) * a.band_fy + yg) * a.band_fx + xc] = 1;
    }
  }
  if (COUNT) {  // the same four striped counters as k_integrate
    __shared__ unsigned s_cnt, s_chg, s_imp, s_rdb;
    if (tid == 0) s_cnt = s_chg = s_imp = s_rdb = 0;
    __syncthreads();
    if (cnt) atomicAdd(&s_cnt, cnt);
    if (chg) atomicAdd(&s_chg, chg);
    if (imp) atomicAdd(&s_imp, imp);
    if (rdb) atomicAdd(&s_rdb, rdb);
    __syncthreads();
    if (tid == 0 && (s_cnt || s_rdb)) {
      const unsigned b = bc.bx + bc.by * bc.gdx + bc.bz * bc.gdx * bc.gdy;
      if (s_cnt) atomicAdd(n_obs + (b & 1023u), (unsigned long long)s_cnt);
      if (s_chg) atomicAdd(n_obs + 1024u + (b & 1023u), (unsigned long long)s_chg);
      if (s_imp) atomicAdd(n_obs + 2048u + (b & 1023u), (unsigned long long)s_imp);
      if (s_rdb) atomicAdd(n_obs + 3072u + (b & 1023u), (unsigned long long)s_rdb);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_integrate_pc (round 6): the HEADLINE instance -- ALLIN, PACKED, integrate_color -- on k_integrate_p's two-stage pipeline.
// A row in flight holds more here (four colour gathers and the colour|count words besides the depths: 20 registers per row,
// two rows), so this kernel runs FEWER waves per SIMD than k_integrate's 64-register instance; what it buys is that a wave
// never sits in the frame gather's s_waitcnt with nothing else to issue (k_integrate: ~80 % VALU-active at eight waves).
// The voxel words are requested in stage A by the quads PREDICTED to be observed -- the quad's own outcome in the last row
// whose stage B has run, i.e. two rows back -- through an exec-free device: an unpredicted quad's byte offset is poisoned
// (all ones: beyond every descriptor's range, the hardware returns zero and fetches nothing), so stage A always issues the
// same TEN loads and the compiler can count them behind the row that waits.  An observed quad the predictor missed asks in
// stage B and waits there (in order: also for the next row's gathers -- rare: where a surface begins along y).
// Arithmetic: k_integrate<ORDER, true, true, COUNT, true, true, false>'s, operation for operation.
// MEASURED (profiles/r06_ab_pipec_call04.txt, five alternations, 2048^3 + colour): 12.80 ms at five waves (91 VGPRs), 12.78-13.46 at
// four, against 12.59-12.64 for k_integrate's own row loop at eight waves -- no gain, nor on a configs[4] slab (12.21 against
// 12.14): with colour the row moves 58.6 GB per launch, 4.6 TB/s = 82 % of what this box's plain read-modify-write sweep
// reaches, while the VALU is 80 % busy -- two limiters at once, neither of which the pipeline touches.  OFF by default (knob
// pipe, bit 1); kept, oracle-gated like every instance (tests/test_integrate_gpu.py), for hardware where the balance differs.
#ifndef TSDF_WPE_PIPEC
#define TSDF_WPE_PIPEC 5  // 91 VGPRs, no scratch (six waves: 80 VGPRs + ten spill operations in the row loop, whose waits undo the pipeline)
#endif
struct PipeRowC {
  float gz[4];
  uint32_t z[4];   // gathered depths (in flight)
  uint32_t cs[4];  // gathered bgra (in flight)
  u4 c4;           // the quad's colour|count words (in flight; zero where not asked)
  u4 d4;           // its distance words (in flight)
};

template <int ORDER, bool COUNT>
static __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TSDF_WPE_PIPEC < TSDF_WPE_MAX ? TSDF_WPE_PIPEC : TSDF_WPE_MAX, TSDF_WPE_MAX)))
k_integrate_pc(const IntegrateArgs a, float *__restrict__ D, uint32_t *__restrict__ RGB, const float *__restrict__ depth,
               const double *__restrict__ cam, const float *__restrict__ ctrx, const float *__restrict__ ctry,
               const float *__restrict__ ctrz, unsigned long long *__restrict__ n_obs, uint8_t *__restrict__ band) {
  const BlockCoords bc = tsdf_block_coords(a.zfast);
  const unsigned tid = threadIdx.x;
  __shared__ f2 s_rcp[256];  // s_rcp[k] = {Rcp32(k + 1).y, the colour average's rounding offset for that divisor}
  __shared__ f4 s_yt[256];
  __shared__ f4 s_cx[256];
  __shared__ uint8_t s_bin[1024];
  TSDF_BAND_DECL;
  {
    const KEntry e = tsdf_ktab_entry(a, tid);
    s_rcp[tid] = (f2){e.y, e.hy};
    const int yy = (int)bc.by * a.rpb * a.TY + (int)tid;
    const float cy_ = ctry[yy < a.ny ? yy : a.ny - 1], cz_ = ctrz[a.z_global0 + (int)bc.bz];
    f4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 3; ++q)
      t[q] = ORDER == TSDF_XFORM_PCL_SSE ? cy_ * a.m[4 * q + 1] + (cz_ * a.m[4 * q + 2] + a.m[4 * q + 3]) : a.m[4 * q + 1] * cy_;
    s_yt[tid] = t;
    const int xq_ = (int)bc.bx * a.TX + (int)(tid & (unsigned)(a.TX - 1));
    if (xq_ < a.qpr) s_cx[tid] = *reinterpret_cast<const f4 *>(ctrx + xq_ * 4);
  }
  if (a.implied_d) tsdf_flags_before(a, bc, band, s_bin, tid);
  __syncthreads();
  const uint64_t quiet = a.implied_d ? tsdf_quiet_passes(a, s_bin, tid) : 0ull;
  const int tx = (int)(tid & (unsigned)(a.TX - 1));
  const int ty = (int)(tid >> a.log2TX);
  const int xq = (int)bc.bx * a.TX + tx;
  const int zl = (int)bc.bz;
  const Rcp32 rneg = rcp32_prepare(a.neg);
  unsigned cnt = 0, chg = 0, imp = 0, rdb = 0;
  const int row0 = (int)bc.by * a.rpb * a.TY;
  const int rows = min(a.rpb * a.TY, a.ny - row0);  // a multiple of TY (the host checked)
  const int64_t e0 = ((int64_t)(a.zl0 + zl) * a.plane_rows + row0) * a.pitch;
  const unsigned span = (unsigned)rows * (unsigned)a.pitch;
  const rsrc_t rsD = make_rsrc(D + e0, span * 4u);
  const rsrc_t rsC = make_rsrc(RGB + e0, span * 4u);
  const i4_rsrc rsFi = make_rsrc_2d(depth, 4u, 0xffffffffu);
  const uint32_t pbits = __float_as_uint(a.pos_over_neg);
  if (xq < a.qpr) {
    const int x4 = xq * 4;
    const float cz = ctrz[a.z_global0 + zl];
    float zt[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) zt[q] = ORDER == TSDF_XFORM_PCL_SSE ? cz * a.m[4 * q + 2] + a.m[4 * q + 3] : a.m[4 * q + 2] * cz;
    const unsigned voff = (unsigned)(ty * (int)a.pitch + x4) * 4u;
    const unsigned row_step = (unsigned)a.TY * (unsigned)a.pitch * 4u;
    bool obs_a = true, obs_b = true;  // this quad's outcome in the last even / odd row whose stage B has run (the predictor)

    // ---- stage A ----------------------------------------------------------------------------------------------------
    auto issue = [&](const int r, PipeRowC &R, const bool pred) {
      const f4 ytv = s_yt[ty + r * a.TY];
      asm volatile("" ::: "memory");
      int pix[4];
      project_quad_allin<ORDER>(a, a.m, cam, s_cx[tid], ytv, zt, pix, R.gz);
      const unsigned soff = (unsigned)r * row_step;
      const bool d_read = !(r < 64 && (quiet >> r & 1ull));  // wave-uniform (implied distances, see k_integrate)
      // [phase: voxel loads]
      const unsigned offp = pred ? voff : 0x7ffffff0u;  // an unpredicted quad asks for nothing: its offset lies beyond the descriptor
      R.d4 = bload128(rsD, d_read ? offp : 0x7ffffff0u, soff);
      R.c4 = bload128(rsC, offp, soff);
      // [phase: frame gather]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        R.z[j] = tsdf_struct_buffer_load_u32(rsFi, pix[j], 0, 0, TSDF_GATHER_AUX);
        R.cs[j] = tsdf_struct_buffer_load_u32(rsFi, pix[j], 0, (int)a.bgra_off, TSDF_GATHER_AUX);
      }
      if (COUNT) rdb += pred ? (d_read ? 32u : 16u) : 0u;
    };

    // ---- stage B ----------------------------------------------------------------------------------------------------
    auto consume = [&](const int r, PipeRowC &R, const bool pred, bool &obs_out) {
      const unsigned soff = (unsigned)r * row_step;
      const bool d_read = !(r < 64 && (quiet >> r & 1ull));
      // [phase: hinge / normalise (hpp:159-198)]
      float raw[4];
      bool act[4];
      bool any = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        raw[j] = __uint_as_float(R.z[j]) - R.gz[j];  // hpp:159
        act[j] = raw[j] >= -a.neg;                   // hpp:152 and :193-196 in one compare (see k_integrate)
        any |= act[j];
      }
      obs_out = any;
      if (__builtin_amdgcn_ballot_w64(any) == 0ull) return;  // nothing of this wave's row is observed
      // [phase: leave the row if nothing is observed; late voxel loads]
      if (any && !pred) {  // an observed quad the predictor missed: its words are requested now, and retired here
        if (d_read) R.d4 = bload128(rsD, voff, soff);
        R.c4 = bload128(rsC, voff, soff);
        if (COUNT) rdb += d_read ? 32u : 16u;
        asm volatile("" ::"v"(R.d4), "v"(R.c4));
      }
      bool any_div = false;
      if (__builtin_fminf(__builtin_fminf(raw[0], raw[1]), __builtin_fminf(raw[2], raw[3])) <= a.pos) {
#pragma unroll
        for (int j = 0; j < 4; ++j) any_div |= act[j] && !(raw[j] > a.pos);
      }
      // [phase: decode count / weight (PACKED)]
      const uint32_t c0[4] = {R.c4.x, R.c4.y, R.c4.z, R.c4.w};
      uint32_t d0u[4] = {R.d4.x, R.d4.y, R.d4.z, R.d4.w};
      // [phase: band / implied-distance flags, hinge rest test]
      bool off_hinge = any_div;
      if (d_read) {
#pragma unroll
        for (int j = 0; j < 4; ++j) off_hinge |= d0u[j] != pbits;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) off_hinge |= c0[j] < 0x01000000u;
      }
      const bool d_moves = __builtin_amdgcn_ballot_w64(any && off_hinge) != 0ull;  // wave-uniform
      // [phase: colour update (octree.cpp:328-337)]
      // (voxel by voxel, nothing kept: the rare distance block below works its operands out again -- registers are what
      // this kernel is short of)
      uint32_t cv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // colour and count: every observed voxel
        float w = (float)(c0[j] >> 24);  // tsdf_decode_w: max_weight is an integer equal to kmax here (ALLIN)
        asm volatile("" : "+v"(w));      // (keeps LLVM from turning the colour sums into integer multiplies: k_integrate, TSDF_LEAN_K)
        const uint32_t k1 = min(c0[j], a.kcap | 0xffffffu) + a.kinc;  // min(k + 1, kmax) << 24 | old rgb (v_cvt_pk_u8 rewrites bytes 0-2)
        const f2 yh = s_rcp[c0[j] >> 24];
        Rcp32 rs;
        rs.nb = -(w + 1.f);
        rs.y = yh.x;
        const float hy = yh.y;
        float dd = 0.f;
        cv[j] = c0[j];
        add_observation_fast<true, false>(dd, w, cv[j], 0.f, R.cs[j], a.wmax, rs, k1, &hy);
      }
      // [phase: d update (octree.cpp:152-163)]
      if (d_moves) {  // (a tenth of the observed wave-rows: surfaces, first observations)
        float d0[4], dn[4], dv[4], w0[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!d_read) d0u[j] = c0[j] < 0x01000000u ? 0xbf800000u : pbits;  // never observed: the reset value; else the hinge value
          d0[j] = __uint_as_float(d0u[j]);
          dn[j] = a.pos_over_neg;  // hpp:189-192
          w0[j] = (float)(c0[j] >> 24);
        }
        if (any_div) {
          TSDF_BAND(((ty + r * a.TY) >> 2) * max(1, a.TX >> 4) + (tx >> 4)) = 1;
#pragma unroll
          for (int j = 0; j < 4; ++j) dn[j] = raw[j] > a.pos ? a.pos_over_neg : div32_fast(raw[j], rneg);  // hpp:198
        }
        bool safe = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Rcp32 rs;
          rs.nb = -(w0[j] + 1.f);
          rs.y = s_rcp[c0[j] >> 24].x;
          dv[j] = div32_fast(d0[j] * w0[j] + dn[j], rs);
          safe &= !act[j] || __builtin_amdgcn_classf(dv[j], 0x108);  // the guard on the RESULT (k_integrate)
        }
        // [phase: IEEE fallback (rare)]
        if (!safe) {
          asm volatile("");
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float wv = w0[j];
            dv[j] = d0[j];
            cv[j] = c0[j] & 0xffffffu;
            add_observation_ieee<true>(dv[j], wv, cv[j], dn[j], R.cs[j], a.wmax);
            cv[j] |= (min(c0[j], a.kcap | 0xffffffu) + a.kinc) & 0xff000000u;
          }
        }
        uint32_t diff_d = 0u, dn_u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dn_u[j] = act[j] ? __float_as_uint(dv[j]) : d0u[j];
          diff_d |= dn_u[j] ^ d0u[j];
          if (COUNT) chg += dn_u[j] != d0u[j] ? 4u : 0u;
        }
        if (diff_d) bstore128(rsD, voff, soff, (u4){dn_u[0], dn_u[1], dn_u[2], dn_u[3]});
      }
      // [phase: select / change detection / store]
      uint32_t diff_c = 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        cv[j] = act[j] ? cv[j] : c0[j];
        diff_c |= cv[j] ^ c0[j];
        if (COUNT) {
          cnt += act[j] ? 1u : 0u;
          imp += act[j] && !d_read ? 1u : 0u;
          chg += cv[j] != c0[j] ? 4u : 0u;
        }
      }
      if (diff_c) bstore128(rsC, voff, soff, (u4){cv[0], cv[1], cv[2], cv[3]});
    };

    PipeRowC A, B;
    const int nr = rows / a.TY;  // (the grid's last block may be shorter: the host checked that ny is a multiple of TY)
    issue(0, A, true);
    int r = 0;
    for (; r + 2 < nr; r += 2) {  // steady state: every trip issues two rows and consumes two
      const bool pb = obs_b;      // row r + 1 asks by the outcome of row r - 1
      issue(r + 1, B, pb);
      const bool pa = obs_a;      // what row r asked by: row r - 2's outcome (the first two rows ask)
      consume(r, A, pa, obs_a);
      const bool pa2 = obs_a;     // row r + 2 asks by the outcome of row r
      issue(r + 2, A, pa2);
      consume(r + 1, B, pb, obs_b);
    }
    if (r + 1 < nr) {
      const bool pb = obs_b;
      issue(r + 1, B, pb);
      const bool pa = obs_a;
      consume(r, A, pa, obs_a);
      consume(r + 1, B, pb, obs_b);
    } else {
      const bool pa = obs_a;
      consume(r, A, pa, obs_a);
    }
  }
  if (band) {
    TSDF_BAND_SYNC();
    const int lf = max(0, a.log2TX - 4), fxb = 1 << lf;
    const int yg0 = (a.y_abs0 + row0) >> 2, yg1 = (a.y_abs0 + row0 + max(rows, 1) - 1) >> 2;
    const int xc0 = (a.x_abs0 + (int)bc.bx * a.TX * 4) >> 6;
    const int n_fl = (yg1 - yg0 + 1) << lf;
    for (int i = TSDF_BAND_FIRST; i < n_fl; i += TSDF_BAND_STEP) {
      const int yg = yg0 + (i >> lf), xc = xc0 + (i & (fxb - 1));
      if (TSDF_BAND(i) && yg < a.band_fy && xc < a.band_fx) band[((int64_t)(a.zl0 + zl) * a.band_fy + yg) * a.band_fx + xc] = 1;
    }
  }
  if (COUNT) {
    __shared__ unsigned s_cnt, s_chg, s_imp, s_rdb;
    if (tid == 0) s_cnt = s_chg = s_imp = s_rdb = 0;
    __syncthreads();
    if (cnt) atomicAdd(&s_cnt, cnt);
    if (chg) atomicAdd(&s_chg, chg);
    if (imp) atomicAdd(&s_imp, imp);
    if (rdb) atomicAdd(&s_rdb, rdb);
    __syncthreads();
    if (tid == 0 && (s_cnt || s_rdb)) {
      const unsigned b = bc.bx + bc.by * bc.gdx + bc.bz * bc.gdx * bc.gdy;
      if (s_cnt) atomicAdd(n_obs + (b & 1023u), (unsigned long long)s_cnt);
      if (s_chg) atomicAdd(n_obs + 1024u + (b & 1023u), (unsigned long long)s_chg);
      if (s_imp) atomicAdd(n_obs + 2048u + (b & 1023u), (unsigned long long)s_imp);
      if (s_rdb) atomicAdd(n_obs + 3072u + (b & 1023u), (unsigned long long)s_rdb);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Two frames per sweep.  k_integrate waits for HBM with all the requests the register file lets it
// keep in flight: half of its time is arithmetic, the other half latency it cannot cover.  When TWO frames
// are at hand, this kernel reads a quad's voxel words once, applies frame A's observation and then frame B's to the
// values in registers, and writes the words back once: the bytes per frame halve and there is twice the arithmetic to
// cover each memory round trip.  updateVoxel (hpp:113-218) is applied frame by frame in order -- the second update sees
// exactly the words the first would have stored -- so the planes are bit-identical to two k_integrate launches.
// Restricted to what the headline runs on: PACKED layout, certified fp32 projection, BOTH poses proved ALLIN by the
// host (every voxel of the slab in sensor range and a pixel inside the image, see k_integrate), nx a multiple of 4.
// Anything else takes two ordinary launches (tsdf_integrate_launch2).
struct Frame2 {
  float m[12];        // frame B's cam_from_vol
  unsigned bgra_off;  // ... and the byte offset of its colour image from its depth image
};
#ifndef TSDF_K2_PIPE
#define TSDF_K2_PIPE 0  // 1 = two-row software pipeline (ISSUE(r + 1) before RETIRE(r)): built and measured in round 4 -- 32 more
                        // registers of row state: at 4 waves 32 VGPRs spill (27.9 ms per frame), at 3 waves none (16.9 ms), against
                        // 13.3 ms for issue + retire back to back at 5 waves (profiles/r04_ab_k_integrate2.txt): occupancy wins again
#endif
#ifndef TSDF_WPE_K2
#define TSDF_WPE_K2 (TSDF_K2_PIPE ? 4 : 5)
#endif

template <int ORDER, bool COLOR, bool COUNT>
static __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TSDF_WPE_K2, TSDF_WPE_MAX)))
k_integrate2(const IntegrateArgs a, const Frame2 fb, float *__restrict__ D, uint32_t *__restrict__ RGB, uint8_t *__restrict__ K8,
             const float *__restrict__ depthA, const float *__restrict__ depthB, const double *__restrict__ cam,
             const float *__restrict__ ctrx, const float *__restrict__ ctry, const float *__restrict__ ctrz,
             unsigned long long *__restrict__ n_obs, uint8_t *__restrict__ band) {
  const unsigned tid = threadIdx.x;
  const BlockCoords bc = tsdf_block_coords(a.zfast);
  __shared__ KEntry s_tab[256];        // per count k: decoded weight, Rcp32(k + 1).y, colour rounding offset, next count (k_integrate)
  __shared__ f4 s_ytA[256], s_ytB[256];  // per row of the block: the row's part of each frame's transform (k_integrate's s_yt)
  __shared__ f4 s_cx[256];             // every thread's own four x centres, re-read each row
  TSDF_BAND_DECL;
  s_tab[tid] = tsdf_ktab_entry(a, tid);
  {
    const int yy = (int)bc.by * a.rpb * a.TY + (int)tid;
    const float cy_ = ctry[yy < a.ny ? yy : a.ny - 1], cz_ = ctrz[a.z_global0 + (int)bc.bz];
    f4 tA = {0.f, 0.f, 0.f, 0.f}, tB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      tA[q] = ORDER == TSDF_XFORM_PCL_SSE ? cy_ * a.m[4 * q + 1] + (cz_ * a.m[4 * q + 2] + a.m[4 * q + 3]) : a.m[4 * q + 1] * cy_;
      tB[q] = ORDER == TSDF_XFORM_PCL_SSE ? cy_ * fb.m[4 * q + 1] + (cz_ * fb.m[4 * q + 2] + fb.m[4 * q + 3]) : fb.m[4 * q + 1] * cy_;
    }
    s_ytA[tid] = tA, s_ytB[tid] = tB;
    const int xq_ = (int)bc.bx * a.TX + (int)(tid & (unsigned)(a.TX - 1));
    if (xq_ < a.qpr) s_cx[tid] = *reinterpret_cast<const f4 *>(ctrx + xq_ * 4);  // (nx is a multiple of 4: every quad is whole)
  }
  // (No implied distances here -- k_integrate's s_bin: built into this kernel in round 4, they cost it 0.5 ms per frame of
  // scalar work and nineteen spilled scalar registers for bytes an issue-bound kernel gains nothing from, DESIGN 3.1b;
  // compiled out in round 5.  The host still keeps the record, and this kernel keeps the flags, for the launches that use them.)
  __syncthreads();
  const int tx = (int)(tid & (unsigned)(a.TX - 1));
  const int ty = (int)(tid >> a.log2TX);
  const int xq = (int)bc.bx * a.TX + tx;
  const int zl = (int)bc.bz;
  const Rcp32 rneg = rcp32_prepare(a.neg);
  unsigned cnt = 0, chg = 0, cntA = 0, cntB = 0, rdb = 0;
  const int row0 = (int)bc.by * a.rpb * a.TY;
  const int rows = min(a.rpb * a.TY, a.ny - row0);
  const int64_t e0 = ((int64_t)(a.zl0 + zl) * a.plane_rows + row0) * a.pitch;
  const unsigned span = (unsigned)rows * (unsigned)a.pitch;
  const rsrc_t rsD = make_rsrc(D + e0, span * 4u);
  const rsrc_t rsC = make_rsrc(COLOR ? RGB + e0 : (uint32_t *)D, COLOR ? span * 4u : 0u);
  const rsrc_t rsK = make_rsrc(!COLOR ? K8 + e0 : (uint8_t *)D, !COLOR ? span : 0u);
  const i4_rsrc rsFA = make_rsrc_2d(depthA, 4u, 0xffffffffu), rsFB = make_rsrc_2d(depthB, 4u, 0xffffffffu);
  const uint32_t hinge_bits = __float_as_uint(a.pos_over_neg);
  if (xq < a.qpr) {
    const int x4 = xq * 4;
    const float cz = ctrz[a.z_global0 + zl];
    float ztA[3], ztB[3];  // the plane's part of the transform: only the left-to-right order still needs it per thread
#pragma unroll
    for (int q = 0; q < 3; ++q) ztA[q] = a.m[4 * q + 2] * cz, ztB[q] = fb.m[4 * q + 2] * cz;
    const unsigned voff = (unsigned)(ty * (int)a.pitch + x4) * 4u;
    const unsigned row_step = (unsigned)a.TY * (unsigned)a.pitch * 4u;
    // A row's work in two stages: ISSUE(r) = project both frames and request everything the row needs at once -- the two
    // frames' depth / colour gathers and the quad's voxel words (read whether or not a voxel turns out to be observed: in
    // this regime -- the whole slab in view -- three quarters are) -- so that a row costs ONE memory round trip per PAIR of
    // frames (issuing frame B's gathers only after frame A's had returned: 16.1 instead of 13.3 ms per frame); RETIRE(r) =
    // finish updateVoxel for both frames on the loaded values and write back what changed.
    struct RowLoads {
      float zsA[4], zsB[4], gzA[4], gzB[4];
      uint32_t csA[4], csB[4];
      u4 d4, c4;
      uint32_t k4;
    };
    auto issue = [&](int r, RowLoads &L) {
      const unsigned soff = (unsigned)r * row_step;
      int pixA[4], pixB[4];
      asm volatile("" ::: "memory");  // (compiler barrier: the LDS reads below are redone each row, see k_integrate)
      project_quad_allin<ORDER>(a, a.m, cam, s_cx[tid], s_ytA[ty + r * a.TY], ztA, pixA, L.gzA);
      project_quad_allin<ORDER>(a, fb.m, cam, s_cx[tid], s_ytB[ty + r * a.TY], ztB, pixB, L.gzB);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        L.zsA[j] = __uint_as_float(tsdf_struct_buffer_load_u32(rsFA, pixA[j], 0, 0, TSDF_GATHER_AUX));
        L.csA[j] = COLOR ? tsdf_struct_buffer_load_u32(rsFA, pixA[j], 0, (int)a.bgra_off, TSDF_GATHER_AUX) : 0u;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        L.zsB[j] = __uint_as_float(tsdf_struct_buffer_load_u32(rsFB, pixB[j], 0, 0, TSDF_GATHER_AUX));
        L.csB[j] = COLOR ? tsdf_struct_buffer_load_u32(rsFB, pixB[j], 0, (int)fb.bgra_off, TSDF_GATHER_AUX) : 0u;
      }
      L.d4 = bload128(rsD, voff, soff);
      L.c4 = (u4){0u, 0u, 0u, 0u};
      L.k4 = 0u;
      if (COLOR) L.c4 = bload128(rsC, voff, soff);
      if (!COLOR) L.k4 = bload32(rsK, voff >> 2, soff >> 2);
      if (COUNT) rdb += 16u + (COLOR ? 16u : 4u);  // plane bytes requested
    };
    auto retire = [&](int r, const RowLoads &L) {
      const unsigned soff = (unsigned)r * row_step;
      // everything the row requested is retired HERE, together (one wait: the voxel words were issued last): a quad none of
      // whose voxels is observed leaves without touching most of it, and a load still in flight across the back edge makes
      // the next row wait, in order, for the previous row's voxel STORES as well (see k_integrate)
      if (COLOR)
        asm volatile("" ::"v"(L.csA[0]), "v"(L.csA[1]), "v"(L.csA[2]), "v"(L.csA[3]), "v"(L.csB[0]), "v"(L.csB[1]), "v"(L.csB[2]),
                     "v"(L.csB[3]), "v"(L.d4), "v"(L.c4));
      else
        asm volatile("" ::"v"(L.d4), "v"(L.k4));
      // ---- hpp:152-198 for one frame: NaN test, projective distance, hinge, normalisation ---------------------------
      // returns bit j = voxel j reaches addObservation; bit 4 = one of them lies inside the truncation band
      auto finish = [&](const float (&zs)[4], const float (&gzs)[4], float (&dn)[4]) -> unsigned {
        unsigned obs = 0;
        float raw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          raw[j] = zs[j] - gzs[j];                 // hpp:159
          const bool act = raw[j] >= -a.neg;       // hpp:152, :193-196 in one compare (a NaN depth is a NaN raw)
          dn[j] = a.pos_over_neg;                  // hpp:189-192
          obs |= act ? 1u << j : 0u;
          obs |= act && !(raw[j] > a.pos) ? 16u : 0u;
        }
        if (obs & 16u) {
#pragma unroll
          for (int j = 0; j < 4; ++j) dn[j] = raw[j] > a.pos ? a.pos_over_neg : div32_fast(raw[j], rneg);  // hpp:198
        }
        if (!a.neg_in_window) {  // truncation limits outside the scale-free divider's window: the compiler's IEEE division
          asm volatile("");
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((obs >> j & 1u) && !(raw[j] > a.pos)) dn[j] = raw[j] / a.neg;
        }
        return obs;
      };
      float dnA[4], dnB[4];
      const unsigned obsA = finish(L.zsA, L.gzA, dnA);
      const unsigned obsB = finish(L.zsB, L.gzB, dnB);
      if (!((obsA | obsB) & 15u)) return;
      const u4 d4 = L.d4, c4 = L.c4;
      const uint32_t k4 = L.k4;
      const uint32_t d0u[4] = {d4.x, d4.y, d4.z, d4.w};
      const uint32_t c0[4] = {c4.x, c4.y, c4.z, c4.w};
      uint32_t du[4], kw[4];  // the state both updates work on: distance bits; colour | count << 24 (or only the count there)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kw[j] = COLOR ? c0[j] : ((k4 << (24 - 8 * j)) & 0xff000000u);
        du[j] = d0u[j];
      }
      // ---- OctreeNode / RGBNode::addObservation (octree.cpp:152-163, 328-337) of one frame on that state ----
      auto apply = [&](unsigned obs, const float (&dn)[4], const uint32_t (&cs)[4]) {
        if (!(obs & 15u)) return;
        const bool any_div = (obs & 16u) != 0u;
        float d0[4], w0[4], dv[4], wv[4], ky[4], khy[4];
        uint32_t cv[4], k1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          d0[j] = __uint_as_float(du[j]);
          const KEntry e = s_tab[kw[j] >> 24];  // tsdf_decode_w, Rcp32(k + 1).y, colour offset, min(k + 1, kmax) << 24
          w0[j] = e.w, ky[j] = e.y, khy[j] = e.hy, k1[j] = e.k1;
        }
        bool d_moves = true;
        if (a.hinge_fixed) {  // free space resting at the hinge value stays there (host-checked identity): skip the d ladder
          bool off_hinge = any_div;
#pragma unroll
          for (int j = 0; j < 4; ++j) off_hinge |= du[j] != hinge_bits;
          d_moves = __builtin_amdgcn_ballot_w64(off_hinge) != 0ull;
        }
        bool safe = true, d_touched = false;
        {
          Rcp32 rs[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            dv[j] = d0[j];
            wv[j] = w0[j];
            cv[j] = kw[j];
            rs[j].nb = -(w0[j] + 1.f);
            rs[j].y = ky[j];
            add_observation_fast<COLOR, false>(dv[j], wv[j], cv[j], dn[j], cs[j], a.wmax, rs[j], COLOR ? k1[j] : 0u, &khy[j]);
          }
          if (d_moves) {
            d_touched = true;
            if (any_div) TSDF_BAND(((ty + r * a.TY) >> 2) * max(1, a.TX >> 4) + (tx >> 4)) = 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) dv[j] = div32_fast(d0[j] * w0[j] + dn[j], rs[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) safe &= !(obs >> j & 1u) || __builtin_amdgcn_classf(dv[j], 0x108);
          }
        }
        if (!safe) {  // a zero / subnormal / non-finite quotient: the compiler's IEEE divisions for the quad
          asm volatile("");
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            dv[j] = d0[j];
            wv[j] = w0[j];
            cv[j] = kw[j] & 0xffffffu;
            add_observation_ieee<COLOR>(dv[j], wv[j], cv[j], dn[j], cs[j], a.wmax);
            if (COLOR) cv[j] |= k1[j];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool act = (obs >> j & 1u) != 0u;
          if (d_touched) du[j] = act ? __float_as_uint(dv[j]) : du[j];
          kw[j] = act ? (COLOR ? cv[j] : k1[j]) : kw[j];
        }
      };
      apply(obsA, dnA, L.csA);
      apply(obsB, dnB, L.csB);
      // ---- write back what changed ----------------------------------------------------------------------------
      uint32_t diff_d = 0u, diff_c = 0u, k4n = 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        diff_d |= du[j] ^ d0u[j];
        if (COLOR) diff_c |= kw[j] ^ c0[j];
        if (!COLOR) k4n |= (kw[j] & 0xff000000u) >> (24 - 8 * j);
        cnt += ((obsA | obsB) >> j & 1u);
        if (COUNT) cntA += (obsA >> j & 1u), cntB += (obsB >> j & 1u);
        if (COUNT) chg += (du[j] != d0u[j] ? 4u : 0u) + (COLOR && kw[j] != c0[j] ? 4u : 0u);
      }
      if (COUNT && !COLOR) chg += (unsigned)__popc(((k4n ^ k4) | ((k4n ^ k4) >> 1) | ((k4n ^ k4) >> 2) | ((k4n ^ k4) >> 3) |
                                                   ((k4n ^ k4) >> 4) | ((k4n ^ k4) >> 5) | ((k4n ^ k4) >> 6) | ((k4n ^ k4) >> 7)) & 0x01010101u);
      if (diff_d) bstore128(rsD, voff, soff, (u4){du[0], du[1], du[2], du[3]});
      if (COLOR && diff_c) bstore128(rsC, voff, soff, (u4){kw[0], kw[1], kw[2], kw[3]});
      if (!COLOR && k4n != k4) bstore32(rsK, voff >> 2, soff >> 2, k4n);
    };
    // rows this thread walks: r = 0 .. nrows - 1 (row0 + ty + r * TY < ny)
    const int left = a.ny - row0 - ty;
    const int nrows = left <= 0 ? 0 : min(a.rpb, (left + a.TY - 1) / a.TY);
#if TSDF_K2_PIPE
    RowLoads L0, L1;
    if (nrows > 0) issue(0, L0);
    for (int r = 0; r < nrows; r += 2) {
      if (r + 1 < nrows) issue(r + 1, L1);
      retire(r, L0);
      if (r + 1 < nrows) {
        if (r + 2 < nrows) issue(r + 2, L0);
        retire(r + 1, L1);
      }
    }
#else
    for (int r = 0; r < nrows; ++r) {
      RowLoads L;
      issue(r, L);
      retire(r, L);
    }
#endif
  }
  if (band) {
    TSDF_BAND_SYNC();  // (per-wave sets: none -- a wave reads back what it wrote itself)
    const int lf = max(0, a.log2TX - 4), fxb = 1 << lf;
    const int yg0 = (a.y_abs0 + row0) >> 2, yg1 = (a.y_abs0 + row0 + max(rows, 1) - 1) >> 2;
    const int xc0 = (a.x_abs0 + (int)bc.bx * a.TX * 4) >> 6;
    const int n_fl = (yg1 - yg0 + 1) << lf;
    for (int i = TSDF_BAND_FIRST; i < n_fl; i += TSDF_BAND_STEP) {
      const int yg = yg0 + (i >> lf), xc = xc0 + (i & (fxb - 1));
      if (TSDF_BAND(i) && yg < a.band_fy && xc < a.band_fx) band[((int64_t)(a.zl0 + zl) * a.band_fy + yg) * a.band_fx + xc] = 1;
    }
  }
  if (COUNT) {
    __shared__ unsigned s_cnt, s_chg, s_cntA, s_cntB, s_rdb;
    if (tid == 0) s_cnt = s_chg = s_cntA = s_cntB = s_rdb = 0;
    __syncthreads();
    if (cnt) atomicAdd(&s_cnt, cnt);
    if (chg) atomicAdd(&s_chg, chg);
    if (cntA) atomicAdd(&s_cntA, cntA);
    if (cntB) atomicAdd(&s_cntB, cntB);
    if (rdb) atomicAdd(&s_rdb, rdb);
    __syncthreads();
    if (tid == 0 && s_rdb) atomicAdd(n_obs + 2560u + ((bc.bx + bc.by * bc.gdx + bc.bz * bc.gdx * bc.gdy) & 511u), (unsigned long long)s_rdb);
    if (tid == 0 && s_cnt) {  // 512 striped slots each: frame A, frame B, either, changed bytes, (slots 2048.. stay 0: voxels
                              // whose distance word was not read -- none in this kernel), plane bytes requested (tsdf_integrate_collect2)
      const unsigned b = (bc.bx + bc.by * bc.gdx + bc.bz * bc.gdx * bc.gdy) & 511u;
      if (s_cntA) atomicAdd(n_obs + b, (unsigned long long)s_cntA);
      if (s_cntB) atomicAdd(n_obs + 512u + b, (unsigned long long)s_cntB);
      atomicAdd(n_obs + 1024u + b, (unsigned long long)s_cnt);
      if (s_chg) atomicAdd(n_obs + 1536u + b, (unsigned long long)s_chg);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Brick-level frustum cull -- the dense counterpart of getFrustumCulledVoxels (tsdf_volume_octree.cpp:
// 619-652), which lets the reference skip coarse octree cells outside a 1.1x FOV frustum.  Here the unit is
// one k_integrate block (up to 1024 voxels of x by rpb*TY rows of one plane) and the test is CONSERVATIVE:
// a block is dropped only if no voxel in it can pass updateVoxel's own tests (hpp:146 sensor range,
// .cpp:616 pixel inside the image), so results are identical with and without it.  The block's rectangle
// of voxel centres is transformed in double; the per-voxel float transform differs from that by at most
// ~4 roundings of magnitude |m||c| + |t|, covered by eps (8x margin); a rectangle in front of the camera
// projects to a convex quadrilateral, so the pixel extremes are at its corners, widened by the projection's
// sensitivity to eps plus one pixel for the truncation toward zero.
struct CullArgs {
  double m[12], fx, fy, cx, cy;
  double zlo, zmax;  // a voxel needs g.z >= zlo (= max(min_sensor_dist, 0), with g.z > 0) and g.z <= zmax
  int W, H, nx, ny, z_global0;
  int bx_vox, by_rows;  // voxels along x / rows along y per block
  int gx, gy, gz;
};

static __host__ __device__ inline bool box_may_be_observed(const CullArgs &c, double x0, double x1, double y0, double y1,
                                                           double z) {
  double gz_min = 1e300, gz_max = -1e300, eps = 0, u_min = 1e300, u_max = -1e300, v_min = 1e300, v_max = -1e300;
  double g[4][3];
  for (int k = 0; k < 4; ++k) {
    const double x = (k & 1) ? x1 : x0, y = (k & 2) ? y1 : y0;
    for (int r = 0; r < 3; ++r) {
      g[k][r] = c.m[4 * r] * x + c.m[4 * r + 1] * y + c.m[4 * r + 2] * z + c.m[4 * r + 3];
      const double mag = fabs(c.m[4 * r] * x) + fabs(c.m[4 * r + 1] * y) + fabs(c.m[4 * r + 2] * z) + fabs(c.m[4 * r + 3]);
      eps = fmax(eps, 8.0 * 4.0 * 5.97e-8 * mag);
    }
    gz_min = fmin(gz_min, g[k][2]);
    gz_max = fmax(gz_max, g[k][2]);
  }
  if (!(eps < 1e300)) return true;                          // non-finite: no claim
  if (gz_max + eps < c.zlo || gz_max + eps <= 0) return false;  // every voxel behind the near bound / camera
  if (gz_min - eps > c.zmax) return false;                  // every voxel beyond the far bound
  const double zc = gz_min - eps;
  if (!(zc > 1e-6 * (fabs(c.m[11]) + fabs(x1 - x0) + fabs(y1 - y0) + 1e-30))) return true;  // touches the camera plane
  double mu_u = 0, mu_v = 0;
  for (int k = 0; k < 4; ++k) {
    const double iz = 1.0 / g[k][2];
    const double u = c.fx * g[k][0] * iz + c.cx, v = c.fy * g[k][1] * iz + c.cy;
    u_min = fmin(u_min, u), u_max = fmax(u_max, u);
    v_min = fmin(v_min, v), v_max = fmax(v_max, v);
    mu_u = fmax(mu_u, fabs(c.fx) * eps * (1.0 + fabs(g[k][0]) / zc) / zc);
    mu_v = fmax(mu_v, fabs(c.fy) * eps * (1.0 + fabs(g[k][1]) / zc) / zc);
  }
  mu_u = 1.0 + 2.0 * mu_u;
  mu_v = 1.0 + 2.0 * mu_v;
  if (u_max < -1.0 - mu_u || u_min > c.W + mu_u) return false;  // (int)R in [0, W) needs -1 < R < W
  if (v_max < -1.0 - mu_v || v_min > c.H + mu_v) return false;
  return true;
}

// Row intervals.  Along a voxel row (y, z fixed) both sets that decide whether a voxel is integrated are INTERVALS of x:
//  * updateVoxel's own tests (sensor range hpp:146, pixel inside the image .cpp:616) cut the row's line g(x) = g0 + x m0
//    with a convex pyramid.  Here conservatively: each test is a linear inequality in x once the division by g.z is
//    multiplied out (g.z > 0), widened by the float transform's error bound eps (the same 8x-margin bound as
//    box_may_be_observed) -- nothing outside [ob_lo, ob_hi] can be observed, so masking it changes no result.
//  * the reference's frustum cull (getFrustumCulledVoxels, tsdf_volume_octree.cpp:619-652, replicated when `rc`): PCL's
//    verdict `(x p0 + y p1) + (z p2 + p3) <= 0` is, for finite operands, a MONOTONE function of the float x (every
//    rounding is monotone), and the centre table increases with the index, so each plane keeps a prefix or a suffix of
//    the row and the six planes keep an interval.  Its ends are found by bisection on the very float expression the
//    per-voxel test evaluates (reference_cull_keeps): EXACT, voxel for voxel.
// One thread per row of the launch; the word is lo | len << 16 in launch-relative x (an empty row: lo = 0xffff).
struct RowArgs {
  double m[12], fx, fy, cx, cy;
  double zlo, zmax;
  int W, H, nx, ny, nz, z_global0;  // the launch box: nx voxels of ctrx, ny rows of ctry, nz planes from z_global0
  int rc;                           // replicate the reference's cull with these planes
  float cull[24];
};

static __host__ __device__ inline bool row_plane_keeps(const float *pl, float x, float y, float z) {
  return (x * pl[0] + y * pl[1]) + (z * pl[2] + 1.0f * pl[3]) <= 0.f;  // == reference_cull_keeps, one plane
}

static __host__ __device__ inline uint32_t row_interval(const RowArgs &c, const float *ctrx, float cyf, float czf) {
  int lo = 0, hi = c.nx - 1;
  // ---- conservative: what updateVoxel can accept at all ----
  {
    const double y = cyf, z = czf, xa = ctrx[0], xb = ctrx[c.nx - 1];
    double g0[3], m0[3], eps = 0;
    for (int r = 0; r < 3; ++r) {
      m0[r] = c.m[4 * r];
      g0[r] = c.m[4 * r + 1] * y + c.m[4 * r + 2] * z + c.m[4 * r + 3];
      const double mag = fmax(fabs(c.m[4 * r] * xa), fabs(c.m[4 * r] * xb)) + fabs(c.m[4 * r + 1] * y) + fabs(c.m[4 * r + 2] * z) + fabs(c.m[4 * r + 3]);
      eps = fmax(eps, 8.0 * 4.0 * 5.97e-8 * mag);
    }
    // a voxel needs  A x + B <= C  for each of:  -g.z <= -zlo + eps,  g.z <= zmax + eps,  and with R = fx g.x / g.z + cx
    // (evaluated on float coordinates within eps of these) -1 < R < W, i.e. fx g.x + (cx + 1) g.z > 0 and
    // fx g.x + (cx - W) g.z < 0, likewise for v -- each with the margin its coefficients give eps, relaxed by 1e-9 relative
    double xlo = -1e300, xhi = 1e300;
    bool none = !(eps < 1e300);
    const double zlo = c.zlo > 0 ? c.zlo : 0;
    const double A[6] = {-m0[2], m0[2], -(c.fx * m0[0] + (c.cx + 1.0) * m0[2]), c.fx * m0[0] + (c.cx - c.W) * m0[2],
                         -(c.fy * m0[1] + (c.cy + 1.0) * m0[2]), c.fy * m0[1] + (c.cy - c.H) * m0[2]};
    const double B[6] = {-g0[2], g0[2], -(c.fx * g0[0] + (c.cx + 1.0) * g0[2]), c.fx * g0[0] + (c.cx - c.W) * g0[2],
                         -(c.fy * g0[1] + (c.cy + 1.0) * g0[2]), c.fy * g0[1] + (c.cy - c.H) * g0[2]};
    const double Cm[6] = {-zlo + eps, c.zmax + eps, (fabs(c.fx) + fabs(c.cx + 1.0)) * eps, (fabs(c.fx) + fabs(c.cx - c.W)) * eps,
                          (fabs(c.fy) + fabs(c.cy + 1.0)) * eps, (fabs(c.fy) + fabs(c.cy - c.H)) * eps};
    bool claim = !none;
    for (int k = 0; k < 6 && claim; ++k) {
      const double slack = 1e-9 * (fabs(B[k]) + fabs(A[k]) * fmax(fabs(xa), fabs(xb)) + fabs(Cm[k])) + 1e-300;
      const double rhs = Cm[k] - B[k] + slack;
      if (!(fabs(A[k]) < 1e300) || !(fabs(rhs) < 1e300)) {
        continue;  // non-finite (an unbounded sensor range): this test bounds nothing
      } else if (A[k] > 0) {
        xhi = fmin(xhi, rhs / A[k]);
      } else if (A[k] < 0) {
        xlo = fmax(xlo, rhs / A[k]);
      } else if (!(0 <= rhs)) {
        xhi = -1e300;  // the row fails this test for every x
      }
    }
    if (claim && xhi < xlo) {
      lo = 1, hi = 0;  // both bounds are conservative: nothing in this row can be observed
    } else if (claim) {
      // widen by the bisection's own granularity: a relative 1e-9 on the bounds and one voxel on each side
      const double w = 1e-9 * (fabs(xlo) < 1e299 ? fabs(xlo) : 0) + 1e-9 * (fabs(xhi) < 1e299 ? fabs(xhi) : 0);
      xlo -= w, xhi += w;
      // first index with ctrx >= xlo, minus one; last index with ctrx <= xhi, plus one (the table increases)
      int a = 0, b = c.nx;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if ((double)ctrx[mid] < xlo) a = mid + 1; else b = mid;
      }
      lo = a > 0 ? a - 1 : 0;
      a = 0, b = c.nx;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if ((double)ctrx[mid] <= xhi) a = mid + 1; else b = mid;
      }
      hi = a < c.nx ? a : c.nx - 1;  // a - 1 is the last one inside; plus one
    }
  }
  // ---- exact: what the reference's cull keeps ----
  if (c.rc) {
    for (int k = 0; k < 6 && lo <= hi; ++k) {
      const float *pl = c.cull + 4 * k;
      if (pl[0] > 0.f) {  // keeps a prefix: the first index in [lo, hi] that fails ends it
        int a = lo, b = hi + 1;
        while (a < b) {
          const int mid = (a + b) >> 1;
          if (row_plane_keeps(pl, ctrx[mid], cyf, czf)) a = mid + 1; else b = mid;
        }
        hi = a - 1;
      } else if (pl[0] < 0.f) {  // keeps a suffix: the first index that passes starts it
        int a = lo, b = hi + 1;
        while (a < b) {
          const int mid = (a + b) >> 1;
          if (row_plane_keeps(pl, ctrx[mid], cyf, czf)) b = mid; else a = mid + 1;
        }
        lo = a;
      } else if (!row_plane_keeps(pl, ctrx[lo], cyf, czf)) {  // +-0 (the host rejects NaN planes): the same verdict for every x
        hi = lo - 1;
      }
    }
  }
  if (lo > hi) return 0x0000ffffu;
  return (uint32_t)lo | ((uint32_t)(hi - lo + 1) << 16);
}

#define TSDF_ROWS_LDS_NX 8192  // x centre tables up to this long are staged in LDS for the bisections (32 KB)
template <bool LDSX>
static __global__ void __launch_bounds__(256)
k_rows(const RowArgs c, const float *__restrict__ ctrx, const float *__restrict__ ctry, const float *__restrict__ ctrz,
       uint32_t *__restrict__ row_iv) {
  // a row's bisections are ~90 DEPENDENT reads of the x table: from LDS they cost a tenth of what L2 hits cost
  __shared__ float s_x[LDSX ? TSDF_ROWS_LDS_NX : 1];
  if (LDSX) {
    for (int k = (int)threadIdx.x; k < c.nx; k += 256) s_x[k] = ctrx[k];
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)c.ny * c.nz) return;
  const int y = (int)(i % c.ny), z = (int)(i / c.ny);
  row_iv[i] = row_interval(c, LDSX ? s_x : ctrx, ctry[y], ctrz[c.z_global0 + z]);
}

// One flag per k_integrate block of a LIVE launch: 0 = no row's interval meets the block's x range (or the block's
// rectangle cannot be observed at all), 1 = every row's interval covers it, 2 = neither.
static __global__ void __launch_bounds__(256)
k_cull(const CullArgs c, const float *__restrict__ ctrx, const float *__restrict__ ctry,
       const float *__restrict__ ctrz, const uint32_t *__restrict__ row_iv, uint8_t *__restrict__ live) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= (int64_t)c.gx * c.gy * c.gz) return;
  const int bx = (int)(b % c.gx), by = (int)((b / c.gx) % c.gy), bz = (int)(b / ((int64_t)c.gx * c.gy));
  const int xa = bx * c.bx_vox, xb = min(c.nx, xa + c.bx_vox) - 1;
  const int ya = by * c.by_rows, yb = min(c.ny, ya + c.by_rows) - 1;
  // the centre tables increase with the index, so the first and last voxel bound the block
  bool any = box_may_be_observed(c, ctrx[xa], ctrx[xb], ctry[ya], ctry[yb], ctrz[c.z_global0 + bz]), all = true;
  if (any) {
    any = false;
    for (int y = ya; y <= yb; ++y) {
      const uint32_t iv = row_iv[(int64_t)bz * c.ny + y];
      const int lo = (int)(iv & 0xffffu), len = (int)(iv >> 16);
      any |= len > 0 && lo <= xb && lo + len - 1 >= xa;
      all &= lo <= xa && lo + len - 1 >= xb;
    }
  }
  live[b] = !any ? 0 : all ? 1 : 2;
}

static float f32_ulp(float v) {
  const float a = fabsf(v);
  return nextafterf(a, INFINITY) - a;
}

// Kernel arguments plus what only the host needs to decide the launch.
struct IntegrateHost {
  IntegrateArgs a;
  float band_u, band_v;  // half-width of the "too close to an integer to trust fp32" zone, in pixels
  double fx, fy;
  int planes;            // planes to integrate
};

static IntegrateHost make_args(tsdf_handle h, const float T[12]) {
  const tsdf_params &p = h->p;
  IntegrateHost hh;
  IntegrateArgs &a = hh.a;
  for (int i = 0; i < 12; ++i) a.m[i] = T[i];
  hh.fx = p.fx;
  hh.fy = p.fy;
  a.fxf = (float)p.fx;
  a.fyf = (float)p.fy;
  {
    // |R~ - (R + s)| <= |q|*(2*2^-24 + 2^-22) + (n + 2)*2^-24 + conversion of f and of c + s, for R~ within one
    // pixel of the image (q = g*f/z, |q| <= max(|c| + 1, n + 1 - c)); see project_fast.  1.5x margin on top; s = band.
    auto band = [](double c, int n) {
      const double q = std::max(fabs(c) + 1.0, fabs((double)n + 1.0 - c));
      const double e = q * (2.0 / 16777216.0 + 1.0 / 4194304.0) * 1.0000005 + ((double)n + 2.0) / 16777216.0 +
                       2.0 * f32_ulp((float)c) + 1e-9;
      return nextafterf((float)(1.5 * e), INFINITY);
    };
    hh.band_u = band(p.cx, p.image_width);
    hh.band_v = band(p.cy, p.image_height);
    a.cxf = (float)(p.cx + (double)hh.band_u);
    a.cyf = (float)(p.cy + (double)hh.band_v);
    a.hb_u = nextafterf(2.f * hh.band_u, INFINITY);  // rounded toward the conservative side
    a.hb_v = nextafterf(2.f * hh.band_v, INFINITY);
    a.hb_max = std::max(a.hb_u, a.hb_v);
  }
  a.zmin = p.min_sensor_dist;
  // !(gz < zmin) && gz > 0 for a non-NaN gz: gz >= zmin when zmin > 0 (<=> gz > the float just below zmin), else gz > 0
  // (zmin <= 0, or NaN: `gz < NaN` never rejects)
  a.zlo = p.min_sensor_dist > 0.f ? nextafterf(p.min_sensor_dist, -INFINITY) : 0.f;
  a.zmax = p.max_sensor_dist;
  a.pos = p.max_dist_pos;
  a.neg = p.max_dist_neg;
  a.wmax = p.max_weight;
  a.kmax = h->kmax;
  a.kcap = h->kmax ? (h->kmax - 1u) << 24 : 0u;
  a.kinc = h->kmax ? 0x01000000u : 0u;
  a.wmax_is_int = p.max_weight == floorf(p.max_weight);
  a.pos_over_neg = p.max_dist_pos / p.max_dist_neg;
  {
    // A voxel sitting at the hinge value p that is observed in free space again stays at p if (p*w + p)/(w + 1) == p in
    // fp32 for every weight the PACKED layout can hold (always so for p == 1): then such a quad skips the d ladder.
    const volatile float pv = a.pos_over_neg;
    bool fixed = h->packed && a.wmax_is_int && std::isnormal(a.pos_over_neg);
    for (unsigned k = 0; fixed && k <= h->kmax; ++k) {
      const volatile float w = fminf((float)k, p.max_weight);
      const volatile float num = pv * w;
      const volatile float sum = num + pv;
      const volatile float den = w + 1.f;
      const volatile float q = sum / den;
      fixed = q == pv;
    }
    a.hinge_fixed = fixed ? 1 : 0;
  }
  a.neg_in_window = p.max_dist_neg >= 0x1p-20f && p.max_dist_neg <= 0x1p20f && p.max_dist_pos <= 0x1p20f;
  a.W = p.image_width;
  a.H = p.image_height;
  a.ny = h->ny;
  a.plane_rows = h->ny;
  a.qpr = (h->nx + 3) / 4;
  hh.planes = h->z_end - h->z_begin;
  a.z_global0 = h->z_begin;
  a.zl0 = h->z_begin - h->z_first;
  int l2 = 0;
  while ((1 << l2) < a.qpr && l2 < 8) ++l2;
  a.log2TX = l2;
  a.TX = 1 << l2;
  a.TY = 256 / a.TX;
  a.rpb = std::max(1, std::min(tsdf_tuning().rows_per_block, 256) / a.TY);  // rpb * TY <= 256: the block's row centres sit in LDS
  a.rpb = std::max(1, std::min(a.rpb, (a.ny + a.TY - 1) / a.TY));           // (... and no taller than the grid)
  a.pitch = h->pitch;
  a.expf_fused_r = h->expf_fused_r;
  a.ref_cull = h->ref_cull ? 1 : 0;
  for (int i = 0; i < 24; ++i) a.cull[i] = h->ref_cull ? h->cull_planes[i] : 0.f;
  a.band_fx = h->band_fx;
  a.band_fy = h->band_fy;
  a.x_abs0 = a.y_abs0 = 0;
  a.zfast = 0;
  return hh;
}

// ---------------------------------------------------------------------------------------------
// TSDF_COLOR_RGB_NORMALIZED: updateVoxel with RGBNormalized::addObservation (src/lib/octree.cpp:380-393).
// A plain kernel -- one thread per voxel, exact fp64 projection, the compiler's IEEE divisions and square
// root, every operation in the reference's order -- because this voxel class is a setter away from the
// default and nothing in the reference's programs selects it.  It moves d, w, four float means and the
// cached getRGB() bytes (octree.cpp:396-402) per observed voxel.
template <int ORDER>
static __global__ void __launch_bounds__(256)
k_integrate_rgbn(const IntegrateArgs a, float *__restrict__ D, float *__restrict__ Wt, uint32_t *__restrict__ RGB,
                 float *__restrict__ RN, float *__restrict__ GN, float *__restrict__ BN, float *__restrict__ IN,
                 const float *__restrict__ depth, const uint32_t *__restrict__ bgra, const double *__restrict__ cam,
                 const float *__restrict__ ctrx, const float *__restrict__ ctry, const float *__restrict__ ctrz,
                 unsigned long long *__restrict__ n_obs) {
  const int x = (int)(blockIdx.x * 256u + threadIdx.x);
  const int y = (int)blockIdx.y, zl = (int)blockIdx.z;
  bool observed = false;
  if (x < (int)a.pitch) {  // the x centre table is NaN beyond nx: those lanes fail the range test
    const float cx = ctrx[x], cy = ctry[y], cz = ctrz[a.z_global0 + zl];
    float g[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)  // pcl::transformPoint (hpp:145) in the summation order of this PCL build
      g[q] = ORDER == TSDF_XFORM_PCL_SSE ? cx * a.m[4 * q] + (cy * a.m[4 * q + 1] + (cz * a.m[4 * q + 2] + a.m[4 * q + 3]))
                                         : ((a.m[4 * q] * cx + a.m[4 * q + 1] * cy) + a.m[4 * q + 2] * cz) + a.m[4 * q + 3];
    const bool in = !(g[2] < a.zmin || g[2] > a.zmax) && g[2] > 0.f &&  // hpp:146, .cpp:616
                    (!a.ref_cull || reference_cull_keeps(a, cx, cy, cz));  // hpp:93-94 (replication mode)
    const int pix = in ? project_exact(a, cam, g[0], g[1], g[2]) : -1;
    if (pix >= 0) {
      const float z = depth[pix];
      float dn = z - g[2];  // hpp:159
      if (!isnan(z) && !(dn < -a.neg)) {  // hpp:152, :193-196
        dn = dn > a.pos ? a.pos_over_neg : dn / a.neg;  // hpp:189-198
        const int64_t vi = ((int64_t)(a.zl0 + zl) * a.plane_rows + y) * a.pitch + x;
        const uint32_t c = bgra[pix];  // PCL memory order b, g, r, a
        const float r = (float)((c >> 16) & 255u), gch = (float)((c >> 8) & 255u), b = (float)(c & 255u);
        float w = Wt[vi], d = D[vi];
        const float wn = 1.f;
        const float wsum = w + wn;
        const float i = sqrtf(r * r + gch * gch + b * b);  // octree.cpp:384 (products and sums are exact)
        const float rf = r / i, gf = gch / i, bf = b / i;   // a black pixel makes these NaN, as in the reference
        const float rn = (w * RN[vi] + wn * rf) / wsum;
        const float gn = (w * GN[vi] + wn * gf) / wsum;
        const float bn = (w * BN[vi] + wn * bf) / wsum;
        const float im = (w * IN[vi] + wn * i) / wsum;
        RN[vi] = rn;
        GN[vi] = gn;
        BN[vi] = bn;
        IN[vi] = im;
        // getRGB (octree.cpp:396-402): uint8_t = float, i.e. cvttss2si and the low byte (NaN -> 0)
        RGB[vi] = ((uint32_t)(int)(rn * im) & 255u) | (((uint32_t)(int)(gn * im) & 255u) << 8) |
                  (((uint32_t)(int)(bn * im) & 255u) << 16);
        uint32_t unused = 0;
        add_observation_ieee<false>(d, w, unused, dn, 0u, a.wmax);
        D[vi] = d;
        Wt[vi] = w;
        observed = true;
      }
    }
  }
  if (n_obs) {
    const unsigned long long m = __ballot(observed);
    if ((threadIdx.x & 63u) == 0u && m)
      atomicAdd(n_obs + ((blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y) & 1023u),
                (unsigned long long)__popcll(m));
  }
}

// ---------------------------------------------------------------------------------------------
// TSDF_COLOR_LAB: updateVoxel with LABNode::addObservation (src/lib/octree.cpp:531-545).  RGB2LAB depends on the
// pixel alone, so it runs once per PIXEL of the frame (k_lab_image, 0.3 M conversions) instead of once per voxel
// observation (~100 M at 512^3): the sRGB curve comes from the host-built table (tsdf_lab_curve), the XYZ sums are
// the reference's double expressions rounded to float where it stores them, and the three cube roots are fp64 pow
// on the device rounded to float like `X = std::pow(double, 1/3.)`.  k_integrate_lab then is the RGB_NORMALIZED
// kernel with three float means and LAB2RGB (octree.cpp:483-527) for the cached getRGB() bytes.
static __device__ __forceinline__ float lab_f(float t) {  // octree.cpp:466-477
  return (double)t > 0.008856 ? (float)pow((double)t, 1 / 3.) : (float)(7.787 * (double)t + (16 / 116.));
}

static __global__ void __launch_bounds__(256)
k_lab_image(const uint32_t *__restrict__ bgra, const float *__restrict__ lut, float4 *__restrict__ lab, int n) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= n) return;
  const uint32_t c = bgra[i];  // PCL memory order b, g, r, a
  const double rf = (double)lut[(c >> 16) & 255u], gf = (double)lut[(c >> 8) & 255u], bf = (double)lut[c & 255u];
  float X = (float)(rf * 0.4124 + gf * 0.3576 + bf * 0.1805);  // :459-461
  float Y = (float)(rf * 0.2126 + gf * 0.7152 + bf * 0.0722);
  float Z = (float)(rf * 0.0193 + gf * 0.1192 + bf * 0.9505);
  X = (float)((double)X / 95.047);  // :463-465
  Y = (float)((double)Y / 100.);
  Z = (float)((double)Z / 108.883);
  X = lab_f(X);
  Y = lab_f(Y);
  Z = lab_f(Z);
  lab[i] = make_float4((116.f * Y) - 16.f, 500.f * (X - Y), 200.f * (Y - Z), 0.f);  // :478-480, float arithmetic
}

static __device__ __forceinline__ float lab_cube(float t) {  // octree.cpp:491-502; t * t is exact in fp64
  const double c = ((double)t * (double)t) * (double)t;
  return c > 0.008856 ? (float)c : (float)(((double)t - 16 / 116.) / 7.787);
}
static __device__ __forceinline__ float lab_gamma(float t) {  // octree.cpp:514-525
  return (double)t > 0.0031308 ? (float)(1.055 * pow((double)t, 1. / 2.4) - 0.055) : (float)((double)t * 12.92);
}
// LAB2RGB (octree.cpp:483-527) -> r | g<<8 | b<<16; uint8_t = float is cvttss2si and the low byte, as in getRGB above
static __device__ __forceinline__ uint32_t lab_to_rgb(float L, float A, float B) {
  float Y = (float)((double)(L + 16.f) / 116.);  // :488-490
  float X = (float)((double)A / 500. + (double)Y);
  float Z = (float)((double)Y - ((double)B / 200.));
  X = lab_cube(X);
  Y = lab_cube(Y);
  Z = lab_cube(Z);
  X = (float)((double)X * 95.047);  // :503-505
  Y = (float)((double)Y * 100.);
  Z = (float)((double)Z * 108.883);
  X /= 100.f;  // :507-509: float / int
  Y /= 100.f;
  Z /= 100.f;
  const double x = X, y = Y, z = Z;
  const float rf = lab_gamma((float)(x * +3.2406 + y * -1.5372 + z * -0.4986));  // :511-513
  const float gf = lab_gamma((float)(x * -0.9689 + y * +1.8758 + z * +0.0415));
  const float bf = lab_gamma((float)(x * +0.0557 + y * -0.2040 + z * +1.0570));
  return ((uint32_t)(int)(rf * 255.f) & 255u) | (((uint32_t)(int)(gf * 255.f) & 255u) << 8) |
         (((uint32_t)(int)(bf * 255.f) & 255u) << 16);
}

template <int ORDER>
static __global__ void __launch_bounds__(256)
k_integrate_lab(const IntegrateArgs a, float *__restrict__ D, float *__restrict__ Wt, uint32_t *__restrict__ RGB,
                float *__restrict__ LM, float *__restrict__ AM, float *__restrict__ BM,
                const float *__restrict__ depth, const float4 *__restrict__ lab, const double *__restrict__ cam,
                const float *__restrict__ ctrx, const float *__restrict__ ctry, const float *__restrict__ ctrz,
                unsigned long long *__restrict__ n_obs) {
  const int x = (int)(blockIdx.x * 256u + threadIdx.x);
  const int y = (int)blockIdx.y, zl = (int)blockIdx.z;
  bool observed = false;
  if (x < (int)a.pitch) {  // the x centre table is NaN beyond nx: those lanes fail the range test
    const float cx = ctrx[x], cy = ctry[y], cz = ctrz[a.z_global0 + zl];
    float g[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)  // pcl::transformPoint (hpp:145) in the summation order of this PCL build
      g[q] = ORDER == TSDF_XFORM_PCL_SSE ? cx * a.m[4 * q] + (cy * a.m[4 * q + 1] + (cz * a.m[4 * q + 2] + a.m[4 * q + 3]))
                                         : ((a.m[4 * q] * cx + a.m[4 * q + 1] * cy) + a.m[4 * q + 2] * cz) + a.m[4 * q + 3];
    const bool in = !(g[2] < a.zmin || g[2] > a.zmax) && g[2] > 0.f &&  // hpp:146, .cpp:616
                    (!a.ref_cull || reference_cull_keeps(a, cx, cy, cz));  // hpp:93-94 (replication mode)
    const int pix = in ? project_exact(a, cam, g[0], g[1], g[2]) : -1;
    if (pix >= 0) {
      const float z = depth[pix];
      float dn = z - g[2];  // hpp:159
      if (!isnan(z) && !(dn < -a.neg)) {  // hpp:152, :193-196
        dn = dn > a.pos ? a.pos_over_neg : dn / a.neg;  // hpp:189-198
        const int64_t vi = ((int64_t)(a.zl0 + zl) * a.plane_rows + y) * a.pitch + x;
        const float4 n = lab[pix];  // RGB2LAB of the pixel (octree.cpp:537)
        float w = Wt[vi], d = D[vi];
        const float wn = 1.f;
        const float wsum = w + wn;  // octree.cpp:535
        const float lm = (w * LM[vi] + wn * n.x) / wsum;  // :540-542
        const float am = (w * AM[vi] + wn * n.y) / wsum;
        const float bm = (w * BM[vi] + wn * n.z) / wsum;
        LM[vi] = lm;
        AM[vi] = am;
        BM[vi] = bm;
        RGB[vi] = lab_to_rgb(lm, am, bm);  // getRGB (octree.cpp:547-551)
        uint32_t unused = 0;
        add_observation_ieee<false>(d, w, unused, dn, 0u, a.wmax);
        D[vi] = d;
        Wt[vi] = w;
        observed = true;
      }
    }
  }
  if (n_obs) {
    const unsigned long long m = __ballot(observed);
    if ((threadIdx.x & 63u) == 0u && m)
      atomicAdd(n_obs + ((blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y) & 1023u),
                (unsigned long long)__popcll(m));
  }
}

// ---------------------------------------------------------------------------------------------
// weight_by_depth_ (hpp:200-202): w_new = 1 * (1 - std::min(pt.z / 10., 1.)) -- a float times a double stored back
// into the float -- then OctreeNode / RGBNode::addObservation with that w_new (octree.cpp:152-163, 328-337).  Only a
// loaded .vol can carry the flag (tsdf_volume_octree.cpp:265), so this is the plain form: one thread per voxel, exact
// fp64 projection, the compiler's IEEE divisions, float weight plane.  BY_DEPTH = false is the same kernel with
// w_new = 1 (used by the tests to pin the plain kernel itself against the fast one).
// PACKED (only with BY_DEPTH = false, i.e. w_new = 1): the weight is the count in byte 3 of the colour word or in the K8
// plane, w = min(k, max_weight), k' = min(k + 1, kmax) -- the same state the fast kernel keeps, so the reference-cull
// replication mode (tsdf_hip_set_reference_cull), which integrates through this kernel, works on every layout.
template <int ORDER, bool COLOR, bool BY_DEPTH, bool PACKED = false>
static __global__ void __launch_bounds__(256)
k_integrate_plain(const IntegrateArgs a, float *__restrict__ D, float *__restrict__ Wt, uint32_t *__restrict__ RGB,
                  uint8_t *__restrict__ K8, float *__restrict__ VM, int32_t *__restrict__ VN,
                  const float *__restrict__ depth, const uint32_t *__restrict__ bgra, const double *__restrict__ cam,
                  const float *__restrict__ ctrx, const float *__restrict__ ctry, const float *__restrict__ ctrz,
                  unsigned long long *__restrict__ n_obs) {
  const int x = (int)(blockIdx.x * 256u + threadIdx.x);
  const int y = (int)blockIdx.y, zl = (int)blockIdx.z;
  bool observed = false;
  if (x < (int)a.pitch) {  // the x centre table is NaN beyond nx: those lanes fail the range test
    const float cx = ctrx[x], cy = ctry[y], cz = ctrz[a.z_global0 + zl];
    float g[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)  // pcl::transformPoint (hpp:145) in the summation order of this PCL build
      g[q] = ORDER == TSDF_XFORM_PCL_SSE ? cx * a.m[4 * q] + (cy * a.m[4 * q + 1] + (cz * a.m[4 * q + 2] + a.m[4 * q + 3]))
                                         : ((a.m[4 * q] * cx + a.m[4 * q + 1] * cy) + a.m[4 * q + 2] * cz) + a.m[4 * q + 3];
    const bool in = !(g[2] < a.zmin || g[2] > a.zmax) && g[2] > 0.f &&  // hpp:146, .cpp:616
                    (!a.ref_cull || reference_cull_keeps(a, cx, cy, cz));  // hpp:93-94 (replication mode)
    const int pix = in ? project_exact(a, cam, g[0], g[1], g[2]) : -1;
    if (pix >= 0) {
      const float z = depth[pix];
      float dn = z - g[2];  // hpp:159
      if (!isnan(z) && !(dn < -a.neg)) {  // hpp:152, :193-196
        dn = dn > a.pos ? a.pos_over_neg : dn / a.neg;  // hpp:189-198
        float wn = 1.f;
        if (BY_DEPTH) {  // hpp:201-202; std::min(a, b) = (b < a) ? b : a
          const double q = (double)z / 10.;
          wn = (float)((double)wn * (1. - ((1. < q) ? 1. : q)));
        }
        const int64_t vi = ((int64_t)(a.zl0 + zl) * a.plane_rows + y) * a.pitch + x;
        unsigned k_old = 0u;
        float w, d = D[vi];
        // weight_by_variance_ (hpp:203-204; VM / VN = OctreeNode::M_ / nsample_, non-NULL only then): once a voxel has
        // more than five samples, w_new *= std::exp(logNormal(d_new, d_, getVariance())) with logNormal (hpp:106-110)
        // = -std::pow(x - mean, 2) / (2 * var) -- the double pow of a float difference (an exact square), a double
        // quotient stored in a float -- getVariance (octree.cpp:281-287) = (M_ / w_) * (nsample_ / (nsample_ - 1)) with
        // an INTEGER quotient, and std::exp(float) = the host libm's expf (tsdf_expf_glibc)
        int ns_old = 0;
        if (!PACKED && VM) {
          ns_old = VN[vi];
          if (ns_old > 5) {
            const float var = (VM[vi] / Wt[vi]) * (float)(ns_old / (ns_old - 1));
            const double dx = (double)(dn - d);
            const float ln = (float)(-(dx * dx) / (double)(2 * var));
            wn *= tsdf_expf_glibc(ln, a.expf_fused_r != 0);
          }
        }
        if (PACKED) {
          k_old = COLOR ? RGB[vi] >> 24 : (unsigned)K8[vi];
          w = tsdf_decode_w(k_old, a.wmax);
        } else {
          w = Wt[vi];
        }
        const float wsum = w + wn;  // (wn: after both weightings)
        if (COLOR) {  // RGBNode::addObservation, octree.cpp:331-335: the OLD w, static_cast<uint8_t> = cvttss2si & 255
          const uint32_t c = bgra[pix], old = RGB[vi];  // PCL memory order b, g, r, a (byte 3 of `old`: the PACKED count)
          uint32_t out = 0u;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float c_old = (float)((old >> (8 * ch)) & 255u);
            const float c_new = (float)((c >> (16 - 8 * ch)) & 255u);
            const float qv = (w * c_old + wn * c_new) / wsum;
            // x86: NaN and anything outside int32 give INT_MIN, whose low byte is 0; v_cvt_i32_f32 gives 0 for NaN
            const bool in_range = qv > -2147483904.f && qv < 2147483648.f;
            out |= (in_range ? ((uint32_t)(int)qv & 255u) : 0u) << (8 * ch);
          }
          RGB[vi] = PACKED ? out | (min(k_old + 1u, a.kmax) << 24) : out;
        }
        const float d_old = d;
        d = (d * w + dn * wn) / wsum;  // octree.cpp:156
        w = wsum;                      // :157
        if (w > a.wmax) w = a.wmax;    // :158-159
        D[vi] = d;
        if (!PACKED && VM) {
          VM[vi] += wn * (dn - d) * (dn - d_old);  // octree.cpp:160
          VN[vi] = ns_old + 1;                      // :161
        }
        if (PACKED) {
          if (!COLOR) K8[vi] = (uint8_t)min(k_old + 1u, a.kmax);
        } else {
          Wt[vi] = w;
        }
        observed = true;
      }
    }
  }
  if (n_obs) {
    const unsigned long long m = __ballot(observed);
    if ((threadIdx.x & 63u) == 0u && m)
      atomicAdd(n_obs + ((blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y) & 1023u),
                (unsigned long long)__popcll(m));
  }
}

static bool fast_projection_ok(const IntegrateHost &a, bool color, bool force = false) {
  // The certified fp32 projection needs a sane camera; anything else takes the exact path only.
  const int knob = tsdf_tuning().fast_projection;
  (void)color;
  const bool want = force || knob != 0;  // auto (-1) = on: since v9 it wins with and without colour (tools/tune_sweep.sh)
  return want && std::isfinite(a.band_u) && std::isfinite(a.band_v) &&
         a.band_u < 0.05f && a.band_v < 0.05f && fabs(a.fx) < 1e6 && fabs(a.fy) < 1e6 && a.a.W < (1 << 23) &&
         a.a.H < (1 << 23);
}

// Implied distances (k_integrate's s_bin): the record of what every flag-keeping launch since the reset did to voxels in
// cells whose flag stayed 0, and whether THIS launch may rebuild distances from counts there.  `flags_kept`: the launch
// maintains the "band seen" flags (else band_exact is already false and no later launch asks).
static int implied_distances(tsdf_handle h, const IntegrateArgs &a, bool flags_kept) {
  if (!flags_kept) return 0;
  uint32_t bits;
  memcpy(&bits, &a.pos_over_neg, 4);
  if (!(h->packed && a.hinge_fixed && h->kmax >= 1u))
    h->rest_state = 2;  // free space may now rest anywhere (or a count of 0 no longer means "never observed")
  else if (h->rest_state == 0)
    h->rest_state = 1, h->rest_bits = bits;
  else if (h->rest_state == 1 && h->rest_bits != bits)
    h->rest_state = 2;  // the truncation limits changed: two hinge values in the planes
  return h->rest_state == 1 && tsdf_tuning().implied_d ? 1 : 0;
}


// Index box of the voxels a frame can possibly observe.  updateVoxel only touches a voxel whose centre maps to
// 0 < g.z <= max_sensor_dist with a pixel (int)(g.x*fx/g.z + cx) inside the image (hpp:146, .cpp:611-617), i.e.
// a point of the pyramid spanned by the camera centre and the far corners of the (slightly widened) image --
// a convex set, so its bounding box in the volume frame is the bounding box of those five points mapped back
// by the inverse of cam_from_vol.  The box is widened by two voxels plus the float transform's error before it
// is turned into index ranges [lo, hi] (inclusive) through the centre tables.  Returns false when no claim can
// be made (singular pose, unbounded range, odd intrinsics); *empty when nothing can be observed at all.
static bool observable_index_box(const tsdf_hip_volume *h, const float T[12], int lo[3], int hi[3], bool *empty) {
  const tsdf_params &p = h->p;
  *empty = false;
  if (!(p.max_sensor_dist > 0.f)) {  // also NaN: g.z <= NaN never holds
    *empty = true;
    return true;
  }
  if (!std::isfinite(p.max_sensor_dist) || !(p.fx > 0) || !(p.fy > 0) || !std::isfinite(p.cx) || !std::isfinite(p.cy))
    return false;
  double A[9], t[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) A[3 * r + c] = T[4 * r + c];
    t[r] = T[4 * r + 3];
  }
  const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
  double scale = 0;
  for (int i = 0; i < 9; ++i) scale = std::max(scale, fabs(A[i]));
  if (!std::isfinite(det) || !(fabs(det) > 1e-9 * scale * scale * scale)) return false;
  const double inv[9] = {(A[4] * A[8] - A[5] * A[7]) / det, (A[2] * A[7] - A[1] * A[8]) / det, (A[1] * A[5] - A[2] * A[4]) / det,
                         (A[5] * A[6] - A[3] * A[8]) / det, (A[0] * A[8] - A[2] * A[6]) / det, (A[2] * A[3] - A[0] * A[5]) / det,
                         (A[3] * A[7] - A[4] * A[6]) / det, (A[1] * A[6] - A[0] * A[7]) / det, (A[0] * A[4] - A[1] * A[3]) / det};
  const double zf = (double)p.max_sensor_dist * (1.0 + 1e-6) + 1e-9;
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, big = 0;
  for (int k = 0; k < 5; ++k) {
    double g[3] = {0, 0, 0};
    if (k < 4) {  // trunc toward zero keeps a pixel coordinate in (-1, W): one more pixel of slack on each side
      const double u = (k & 1) ? p.image_width + 1.0 : -2.0, v = (k & 2) ? p.image_height + 1.0 : -2.0;
      g[0] = (u - p.cx) / p.fx * zf;
      g[1] = (v - p.cy) / p.fy * zf;
      g[2] = zf;
    }
    for (int r = 0; r < 3; ++r) {
      const double c = inv[3 * r] * (g[0] - t[0]) + inv[3 * r + 1] * (g[1] - t[1]) + inv[3 * r + 2] * (g[2] - t[2]);
      if (!std::isfinite(c)) return false;
      mn[r] = std::min(mn[r], c);
      mx[r] = std::max(mx[r], c);
      big = std::max(big, fabs(c));
    }
  }
  for (int a = 0; a < 3; ++a) {
    const std::vector<float> &ctr = h->h_ctr[a];
    const int n = (int)ctr.size();
    const double margin = 2.0 * (double)p.size[a] / (double)p.res[a] + 1e-5 * (big + fabs((double)p.size[a]));
    if (mx[a] + margin < (double)ctr.front() || mn[a] - margin > (double)ctr.back()) *empty = true;  // wholly beside the grid
    lo[a] = (int)(std::lower_bound(ctr.begin(), ctr.end(), (float)(mn[a] - margin)) - ctr.begin());
    hi[a] = (int)(std::upper_bound(ctr.begin(), ctr.end(), (float)(mx[a] + margin)) - ctr.begin()) - 1;
    if (lo[a] > 0) --lo[a];  // (the float casts above may have rounded inwards)
    if (hi[a] < n - 1) ++hi[a];
    if (lo[a] > hi[a]) *empty = true;
  }
  return true;
}

// ALLIN's margins (1e-3 m in depth, one pixel in the image) must dwarf the float transform's rounding: |g| <= ~1e3 m
// keeps that below 1e-4 m, and a depth of at least 0.05 m keeps its image below 0.01 * f pixels ... so require both,
// scaled by the focal length.
static bool zlo_margin_ok(const float T[12], const tsdf_hip_volume *h) {
  double big = 0, zmin = 1e300;
  for (int k = 0; k < 8; ++k) {
    const double x = h->h_ctr[0][(k & 1) ? h->nx - 1 : 0], y = h->h_ctr[1][(k & 2) ? h->ny - 1 : 0],
                 z = h->h_ctr[2][(k & 4) ? h->z_end - 1 : h->z_begin];
    for (int r = 0; r < 3; ++r) {
      const double g = fabs((double)T[4 * r] * x) + fabs((double)T[4 * r + 1] * y) + fabs((double)T[4 * r + 2] * z) + fabs((double)T[4 * r + 3]);
      big = std::max(big, g);
    }
    zmin = std::min(zmin, (double)T[8] * x + (double)T[9] * y + (double)T[10] * z + (double)T[11]);
  }
  const double err = 8.0 * 6e-8 * big;                                   // float transform, 8x margin
  const double f = std::max(fabs(h->p.fx), fabs(h->p.fy));
  return err < 1e-4 && zmin > 0.05 && f * err * (1.0 + big / zmin) / zmin < 0.05;  // < 1/20 pixel
}

// Do the six planes of the reference's frustum cull keep EVERY voxel of this handle's slab?  Each plane's verdict is
// `dot <= 0` of an affine function of the centre, so its maximum over the slab's box of centres sits at one of the eight
// corner voxels; evaluated in double, plus a bound on what the per-voxel float evaluation (three products, three sums)
// can add.  True for any ordinary camera looking at a volume inside its sensor range -- then the cull is a no-op for this
// frame and the launch need not know about it.  Non-finite planes: false (the per-voxel test decides).
static bool reference_cull_keeps_whole_slab(const tsdf_hip_volume *h, const float *planes) {
  for (int k = 0; k < 6; ++k) {
    const float *pl = planes + 4 * k;
    double worst = -1e300, mag = 0;
    for (int c = 0; c < 8; ++c) {
      const double x = h->h_ctr[0][(c & 1) ? h->nx - 1 : 0], y = h->h_ctr[1][(c & 2) ? h->ny - 1 : 0],
                   z = h->h_ctr[2][(c & 4) ? h->z_end - 1 : h->z_begin];
      const double v = (double)pl[0] * x + (double)pl[1] * y + (double)pl[2] * z + (double)pl[3];
      if (!std::isfinite(v)) return false;
      worst = std::max(worst, v);
      mag = std::max(mag, fabs((double)pl[0] * x) + fabs((double)pl[1] * y) + fabs((double)pl[2] * z) + fabs((double)pl[3]));
    }
    // (`<= 0` keeps, so <= suffices; with min_sensor_dist = 0 -- the programs' default -- the near plane's corners all
    // coincide with the camera centre and PCL's near plane is the zero vector: its dot is 0 for every voxel, kept)
    if (!(worst + 8.0 * 4.0 * 5.97e-8 * mag <= 0.0)) return false;
  }
  return true;
}

// Can a launch use row intervals (k_rows)?  The interval words hold 16-bit x indices, the bisections need an increasing
// x centre table, and the exact half (`planes`: the reference's cull) needs finite planes of moderate size so that every
// product and sum of the per-voxel test is finite (then each rounding is monotone in x).
static bool row_intervals_usable(const tsdf_hip_volume *h, bool planes) {
  if (h->pitch > 0xfff0 || !h->ctr_increasing[0]) return false;
  if (planes)
    for (int i = 0; i < 24; ++i)
      if (!(fabsf(h->cull_planes[i]) < 1e30f)) return false;
  return true;
}

// The host's ALLIN proof: the slab's eight corner voxels inside {sensor range, image minus a border} with 1e-3 m / one
// pixel to spare.  The set is convex, so every voxel centre lies inside in exact arithmetic, and the margins cover the
// float transform (~1e-7 relative) and the projection's sensitivity to it while the coordinates stay moderate
// (zlo_margin_ok, checked separately).
static bool slab_all_inside(const tsdf_hip_volume *h, const float T[12]) {
  const tsdf_params &p = h->p;
  bool all_inside = true;
  for (int k = 0; k < 8 && all_inside; ++k) {
    const double x = h->h_ctr[0][(k & 1) ? h->nx - 1 : 0], y = h->h_ctr[1][(k & 2) ? h->ny - 1 : 0],
                 z = h->h_ctr[2][(k & 4) ? h->z_end - 1 : h->z_begin];
    double g[3];
    for (int r = 0; r < 3; ++r) g[r] = (double)T[4 * r] * x + (double)T[4 * r + 1] * y + (double)T[4 * r + 2] * z + (double)T[4 * r + 3];
    const double u = p.fx * g[0] / g[2] + p.cx, v = p.fy * g[1] / g[2] + p.cy;
    const double zlo = p.min_sensor_dist > 0 ? p.min_sensor_dist : 0;
    all_inside = g[2] > zlo + 1e-3 && g[2] > 1e-3 && g[2] < p.max_sensor_dist - 1e-3 && u > 1 && u < p.image_width - 2 && v > 1 &&
                 v < p.image_height - 2;
  }
  return all_inside;
}

// Asynchronous half: queues the launch on the handle's stream.  `count` selects the counting instance, whose striped
// counters stay in h->counter until tsdf_integrate_collect reads them (a multi-GPU set launches every slab first and
// collects afterwards, so the slabs count concurrently).
int tsdf_integrate_launch(tsdf_handle h, const float *d_depth, const uint32_t *d_bgra, const float T[12], bool count) {
  const tsdf_params &p = h->p;
  h->count_slots = 0;
  h->count_ran = false;
  const IntegrateHost hh = make_args(h, T);
  IntegrateArgs a = hh.a;
  unsigned gx = (unsigned)((a.qpr + a.TX - 1) / a.TX);
  unsigned gy = (unsigned)((a.ny + a.rpb * a.TY - 1) / (a.rpb * a.TY));
  unsigned gz = (unsigned)hh.planes;
  if (gy > 65535u || gz > 65535u) {
    tsdf_set_error("grid too large for one launch");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  const bool color = p.integrate_color != 0;
  if (color && !d_bgra) {
    tsdf_set_error("integrate_color is set but no colour image was given");
    return TSDF_HIP_E_INVALID;
  }
  if (h->weight_by_variance && (!h->vm || !h->vn || h->packed || h->cn[0])) {
    tsdf_set_error("weight_by_variance_ (hpp:203-204) needs the per-voxel M_ / nsample_ planes: F32W layout, TSDF_COLOR_RGB");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  // the plain per-voxel kernel: weight_by_depth_ (2), the test knob "plain_kernel" (1), or the reference-cull replication
  // mode (3: any layout; the fast kernels know nothing about the six planes)
  // The reference's frustum cull (tsdf_hip_set_reference_cull): nothing to do when the six planes provably keep every
  // voxel of this slab (ordinary cameras: the cull is a no-op and the launch is the usual one); otherwise the fast kernel
  // applies it through the row intervals (k_rows), or -- planes that are not finite, a grid too wide for the interval
  // words, the `refcull_plain` knob -- the plain per-voxel kernel tests the six planes itself.
  const bool rc = h->ref_cull && !reference_cull_keeps_whole_slab(h, h->cull_planes);
  const bool rc_rows = rc && !h->cn[0] && !tsdf_tuning().refcull_plain && row_intervals_usable(h, true);
  const int plain_mode = h->weight_by_depth ? 2 : h->weight_by_variance ? 4 : (rc && !rc_rows && !h->cn[0]) ? 3 : (tsdf_tuning().plain_kernel && !h->packed && !h->cn[0] ? 1 : 0);
  a.ref_cull = rc ? 1 : 0;  // (the plain kernels test the planes per voxel only when they can bite)
  if (plain_mode || h->cn[0]) h->band_exact = false;  // the plain kernels keep no "band seen" flags: marching cubes reads everything
  if (plain_mode) {
    if ((h->packed && plain_mode != 3) || h->cn[0]) {  // (2 and 4 need float weights)
      tsdf_set_error("weight_by_depth needs the F32W layout and TSDF_COLOR_RGB");
      return TSDF_HIP_E_UNSUPPORTED;
    }
    if (count) TSDF_HIP_TRY(hipMemsetAsync(h->counter, 0, 1024 * sizeof(unsigned long long), h->stream));
    bool pose_ok = true;
    for (int i = 0; i < 12; ++i) pose_ok &= std::isfinite(T[i]) && fabsf(T[i]) <= 1e15f;
    if (pose_ok) {
      const dim3 grid((unsigned)((h->pitch + 255) / 256), (unsigned)h->ny, gz), block(256);
      if (grid.y > 65535u) {
        tsdf_set_error("grid too large for one launch");
        return TSDF_HIP_E_UNSUPPORTED;
      }
#define LAUNCH_PLAIN(ORDER, COLOR, BYD, PK)                                                                             \
  hipLaunchKernelGGL((k_integrate_plain<ORDER, COLOR, BYD, PK>), grid, block, 0, h->stream, a, h->d, h->w, h->rgb, h->k8, \
                     h->weight_by_variance ? h->vm : nullptr, h->weight_by_variance ? h->vn : nullptr, d_depth, d_bgra,   \
                     h->cam64, h->ctr[0], h->ctr[1], h->ctr[2], count ? h->counter : nullptr)
#define LP2(ORDER, COLOR)                      \
  do {                                         \
    if (plain_mode == 2)                       \
      LAUNCH_PLAIN(ORDER, COLOR, true, false); \
    else if (h->packed)                        \
      LAUNCH_PLAIN(ORDER, COLOR, false, true); \
    else                                       \
      LAUNCH_PLAIN(ORDER, COLOR, false, false); \
  } while (0)
      if (p.xform_order == TSDF_XFORM_PCL_SSE) {
        if (color)
          LP2(TSDF_XFORM_PCL_SSE, true);
        else
          LP2(TSDF_XFORM_PCL_SSE, false);
      } else {
        if (color)
          LP2(TSDF_XFORM_LEFT_TO_RIGHT, true);
        else
          LP2(TSDF_XFORM_LEFT_TO_RIGHT, false);
      }
#undef LP2
#undef LAUNCH_PLAIN
      TSDF_HIP_TRY(hipGetLastError());
    }
    h->count_slots = count ? 1024 : 0;  // these kernels count observations only (no changed-byte slots)
    h->count_ran = pose_ok;
    return TSDF_HIP_OK;
  }
  if (h->cn[0]) {  // TSDF_COLOR_RGB_NORMALIZED / TSDF_COLOR_LAB: their own plain kernels
    if (count) TSDF_HIP_TRY(hipMemsetAsync(h->counter, 0, 1024 * sizeof(unsigned long long), h->stream));
    bool pose_ok = true;
    for (int i = 0; i < 12; ++i) pose_ok &= std::isfinite(T[i]) && fabsf(T[i]) <= 1e15f;
    if (pose_ok) {
      const dim3 grid((unsigned)((h->pitch + 255) / 256), (unsigned)h->ny, gz), block(256);
      if (grid.y > 65535u) {
        tsdf_set_error("grid too large for one launch");
        return TSDF_HIP_E_UNSUPPORTED;
      }
      if (h->lab_img) {  // TSDF_COLOR_LAB: convert the frame's pixels once, then the per-voxel kernel
        const int npx = p.image_width * p.image_height;
        hipLaunchKernelGGL(k_lab_image, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, h->stream, d_bgra,
                           h->lab_lut, h->lab_img, npx);
      }
#define LAUNCH_RGBN(ORDER)                                                                                        \
  do {                                                                                                            \
    if (h->lab_img)                                                                                               \
      hipLaunchKernelGGL((k_integrate_lab<ORDER>), grid, block, 0, h->stream, a, h->d, h->w, h->rgb, h->cn[0],    \
                         h->cn[1], h->cn[2], d_depth, h->lab_img, h->cam64, h->ctr[0], h->ctr[1], h->ctr[2],      \
                         count ? h->counter : nullptr);                                                           \
    else                                                                                                          \
      hipLaunchKernelGGL((k_integrate_rgbn<ORDER>), grid, block, 0, h->stream, a, h->d, h->w, h->rgb, h->cn[0],   \
                         h->cn[1], h->cn[2], h->cn[3], d_depth, d_bgra, h->cam64, h->ctr[0], h->ctr[1], h->ctr[2], \
                         count ? h->counter : nullptr);                                                           \
  } while (0)
      if (p.xform_order == TSDF_XFORM_PCL_SSE)
        LAUNCH_RGBN(TSDF_XFORM_PCL_SSE);
      else
        LAUNCH_RGBN(TSDF_XFORM_LEFT_TO_RIGHT);
#undef LAUNCH_RGBN
      TSDF_HIP_TRY(hipGetLastError());
    }
    h->count_slots = count ? 1024 : 0;  // these kernels count observations only (no changed-byte slots)
    h->count_ran = pose_ok;
    return TSDF_HIP_OK;
  }
  // The kernel reads the frame through ONE buffer descriptor based at the depth image, with the colour
  // image at a 32-bit byte offset from it.  Caller buffers laid out otherwise are staged into the handle's
  // [depth | bgra] allocation first (two device copies of 1.2 MB each at 640x480).
  const size_t npx = (size_t)p.image_width * p.image_height;
  a.bgra_off = 0;
  if (color) {
    const char *zd = (const char *)d_depth, *cb = (const char *)d_bgra;
    if (cb >= zd + npx * 4 && (size_t)(cb - zd) + npx * 4 < (1ull << 31)) {
      a.bgra_off = (unsigned)(cb - zd);
    } else {
      if (d_depth != h->frame_depth)
        TSDF_HIP_TRY(hipMemcpyAsync(h->frame_depth, d_depth, npx * 4, hipMemcpyDeviceToDevice, h->stream));
      if (d_bgra != h->frame_bgra)
        TSDF_HIP_TRY(hipMemcpyAsync(h->frame_bgra, d_bgra, npx * 4, hipMemcpyDeviceToDevice, h->stream));
      d_depth = h->frame_depth;
      a.bgra_off = (unsigned)(npx * 4);
    }
  }
  // Cull (TSDF_HIP_CULL: 1 on when useful, 0 off, 2 always): skipped when the whole slab is provably inside the
  // frustum and sensor range (convex frustum: test the slab's 8 corners), the turntable case.  Otherwise
  //  (1) the launch shrinks to the blocks that meet the index box of the observable pyramid
  //      (observable_index_box; host only -- pointers, centre tables and limits are offset to the box's first
  //      block, the kernel does not know), and
  //  (2) k_cull flags, inside that box, the blocks none of whose voxels can be observed.
  // Neither changes results: both are conservative.
  const uint8_t *live = nullptr;
  float *D = h->d, *Wt = h->w;
  uint32_t *RGB = h->rgb;
  uint8_t *K8 = h->k8;
  const float *ctrx = h->ctr[0], *ctry = h->ctr[1];
  bool nothing_observable = false;
  bool allin = false;  // every voxel of the launch in sensor range and a pixel inside the image: the ALLIN instance
  bool all_inside = false;
  if (tsdf_tuning().cull) {
    all_inside = tsdf_tuning().cull != 2 && slab_all_inside(h, T);
    allin = all_inside && tsdf_tuning().allin && (h->nx & 3) == 0 && zlo_margin_ok(T, h) &&
            (!h->packed || (a.wmax_is_int && (float)h->kmax == p.max_weight));
  }
  if (rc) allin = false;  // (the ALLIN instance knows no row intervals; measured: an ALLIN pass over the blocks the cull leaves
                          // whole + an interval pass over the rest took 16.0 + 3.8 ms where ONE interval pass takes 18 ms)
  // LIVE launch: the frame cannot see the whole slab (or the reference's cull bites): row intervals + block flags
  const bool want_live = (tsdf_tuning().cull && !all_inside && row_intervals_usable(h, false)) || rc_rows;
  if (want_live) {
    // narrow blocks: 32 quads (128 voxels) by 8 rows per pass, 64 rows per block, so that the flags and a wave's row skip follow the frustum's outline (a block
    // of a whole 1024-voxel row group is mostly outside it when the camera sits inside the volume)
    // (a slab that is wholly in view -- only the reference's cull decides anything -- keeps the streaming shape)
    const int ltx = std::max(4, std::min(8, tsdf_tuning().live_log2tx));  // 32 quads by default (a knob for A/B runs: 16 .. 256)
    if (a.TX > (1 << ltx) && !all_inside) {
      a.TX = 1 << ltx, a.log2TX = ltx, a.TY = 256 >> ltx;
      // (twice the rows of a full-width block: 128 voxels x 64 rows, eight passes -- Scene B at 2048^3: 0.43 / 0.36 / 0.34 /
      // 0.36 ms per frame with 16 / 32 / 64 / 128 rows)
      a.rpb = std::max(1, std::min(2 * tsdf_tuning().rows_per_block, 64) / a.TY);  // (64 rows: the default knob's own value since round 6)
      gx = (unsigned)((a.qpr + a.TX - 1) / a.TX);
      gy = (unsigned)((a.ny + a.rpb * a.TY - 1) / (a.rpb * a.TY));
    }
    int lo[3], hi[3];
    bool empty = false;
    if (tsdf_tuning().cull && !all_inside && observable_index_box(h, T, lo, hi, &empty)) {
      lo[2] = std::max(lo[2], h->z_begin);
      hi[2] = std::min(hi[2], h->z_end - 1);
      if (empty || lo[2] > hi[2]) {
        nothing_observable = true;
      } else {
        const int bxv = a.TX * 4, byr = a.rpb * a.TY;  // voxels / rows per block
        const int bx0 = lo[0] / bxv, bx1 = hi[0] / bxv, by0 = lo[1] / byr, by1 = hi[1] / byr;
        const int x_off = bx0 * bxv, y_off = by0 * byr, z_off = lo[2] - h->z_begin;
        const int64_t e_off = (int64_t)y_off * h->pitch + x_off;
        D += e_off;
        if (Wt) Wt += e_off;
        if (RGB) RGB += e_off;
        if (K8) K8 += e_off;
        ctrx += x_off;
        ctry += y_off;
        a.qpr -= x_off / 4;
        a.ny -= y_off;
        a.x_abs0 = x_off;
        a.y_abs0 = y_off;
        a.z_global0 += z_off;
        a.zl0 += z_off;
        gx = (unsigned)(bx1 - bx0 + 1);
        gy = (unsigned)(by1 - by0 + 1);
        gz = (unsigned)(hi[2] - lo[2] + 1);
      }
    }
    if (gy > 65535u) {
      tsdf_set_error("grid too large for one launch");
      return TSDF_HIP_E_UNSUPPORTED;
    }
    if (!nothing_observable) {
      CullArgs c;
      for (int i = 0; i < 12; ++i) c.m[i] = T[i];
      c.fx = p.fx, c.fy = p.fy, c.cx = p.cx, c.cy = p.cy;
      c.zlo = p.min_sensor_dist > 0 ? p.min_sensor_dist : 0;
      c.zmax = p.max_sensor_dist;
      c.W = p.image_width, c.H = p.image_height;
      c.nx = h->nx - (int)(ctrx - h->ctr[0]), c.ny = a.ny, c.z_global0 = a.z_global0;
      c.bx_vox = a.TX * 4, c.by_rows = a.rpb * a.TY;
      c.gx = (int)gx, c.gy = (int)gy, c.gz = (int)gz;
      // the rows the launch's blocks cover (the last row group may reach past the box, never past the grid)
      const int rows = std::min(a.ny, (int)gy * a.rpb * a.TY);
      a.ny = rows;
      c.ny = rows;
      RowArgs ra;
      for (int i = 0; i < 12; ++i) ra.m[i] = T[i];
      ra.fx = p.fx, ra.fy = p.fy, ra.cx = p.cx, ra.cy = p.cy;
      ra.zlo = c.zlo, ra.zmax = c.zmax;
      ra.W = c.W, ra.H = c.H;
      ra.nx = std::min(c.nx, (int)gx * a.TX * 4), ra.ny = rows, ra.nz = (int)gz, ra.z_global0 = a.z_global0;
      ra.rc = rc_rows ? 1 : 0;
      for (int i = 0; i < 24; ++i) ra.cull[i] = rc_rows ? h->cull_planes[i] : 0.f;
      const size_t nb = (size_t)gx * gy * gz, nrows = (size_t)rows * gz;
      if (nb > h->live_cap || nrows > h->row_iv_cap) {
        TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
        if (nb > h->live_cap) {
          if (h->live) TSDF_HIP_TRY(hipFree(h->live));
          h->live = nullptr, h->live_cap = 0;
          TSDF_HIP_TRY(hipMalloc(&h->live, nb));
          h->live_cap = nb;
        }
        if (nrows > h->row_iv_cap) {
          if (h->row_iv) TSDF_HIP_TRY(hipFree(h->row_iv));
          h->row_iv = nullptr, h->row_iv_cap = 0;
          TSDF_HIP_TRY(hipMalloc(&h->row_iv, nrows * sizeof(uint32_t)));
          h->row_iv_cap = nrows;
        }
      }
      if (ra.nx <= TSDF_ROWS_LDS_NX)
        hipLaunchKernelGGL(k_rows<true>, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, h->stream, ra, ctrx, ctry, h->ctr[2], h->row_iv);
      else
        hipLaunchKernelGGL(k_rows<false>, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, h->stream, ra, ctrx, ctry, h->ctr[2], h->row_iv);
      hipLaunchKernelGGL(k_cull, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, h->stream, c, ctrx, ctry, h->ctr[2], h->row_iv, h->live);
      TSDF_HIP_TRY(hipGetLastError());
      live = h->live;
    }
  }
  if (count) TSDF_HIP_TRY(hipMemsetAsync(h->counter, 0, 4096 * sizeof(unsigned long long), h->stream));
  // A pose with a non-finite (or absurdly large) entry makes g.x/g.y/g.z non-finite or out of sensor
  // range for every voxel, and the reference then observes nothing (u/v become INT_MIN or g.z fails
  // hpp:146 / .cpp:616).  Same here, without launching: the kernel may assume finite arithmetic.
  bool pose_ok = true;
  for (int i = 0; i < 12; ++i) pose_ok &= std::isfinite(T[i]) && fabsf(T[i]) <= 1e15f;
  const bool fastproj = fast_projection_ok(hh, p.integrate_color != 0);
  h->last_launch[0] = h->last_launch[1] = h->last_launch[2] = h->last_launch[3] = 0;
  // the kernel keeps the "band seen" flags per block in LDS with block-local row groups: they are the volume's row groups
  // only if every block starts on a multiple of 4 rows (always, unless the rows_per_block knob was turned below 4);
  // otherwise this launch keeps no flags and marching cubes reads everything until the next reset
  uint8_t *band_arg = h->band_exact && ((a.rpb * a.TY) & 3) == 0 && (a.y_abs0 & 3) == 0 ? h->band : nullptr;
  if (!band_arg) h->band_exact = false;
  a.implied_d = implied_distances(h, a, band_arg != nullptr);
  h->last_implied_on = a.implied_d != 0;
  if (pose_ok && !nothing_observable) {
    // planes fastest when the frame outgrows an XCD's L2 (knob zfast: -1 auto, 0 / 1 force)
    const size_t frame_bytes = npx * (color ? 8 : 4);
    a.zfast = tsdf_tuning().zfast < 0 ? (frame_bytes > (3u << 20) && gx <= 65535u) : (tsdf_tuning().zfast != 0 && gx <= 65535u);
    const dim3 grid(a.zfast ? gz : gx, gy, a.zfast ? gx : gz), block(256);
    h->last_launch[0] = fastproj && allin && !live;
    h->last_launch[1] = fastproj;
    h->last_launch[2] = live ? (rc_rows ? 2 : 1) : 0;
    h->last_launch[3] = (int)std::min<uint64_t>((uint64_t)gx * gy * gz, 0x7fffffffu);
#define LAUNCH(ORDER, COLOR, FP, COUNT, PK, AI, LV)                                                                  \
  hipLaunchKernelGGL((k_integrate<ORDER, COLOR, FP, COUNT, PK, AI, LV>), grid, block, 0, h->stream, a, D, Wt, RGB, K8, \
                     d_depth, h->cam64, ctrx, ctry, h->ctr[2], h->counter, live, band_arg, h->row_iv)
#define L6(ORDER, COLOR, FP, COUNT, PK)                  \
  do {                                                   \
    if (live)                                            \
      LAUNCH(ORDER, COLOR, FP, COUNT, PK, false, true);  \
    else if (FP && allin)                                \
      LAUNCH(ORDER, COLOR, FP, COUNT, PK, FP, false);    \
    else                                                 \
      LAUNCH(ORDER, COLOR, FP, COUNT, PK, false, false); \
  } while (0)
#define L5(ORDER, COLOR, FP, COUNT)     \
  do {                                  \
    if (h->packed)                      \
      L6(ORDER, COLOR, FP, COUNT, true);  \
    else                                \
      L6(ORDER, COLOR, FP, COUNT, false); \
  } while (0)
#define L4(ORDER, COLOR, FP) \
  do {                       \
    if (count)               \
      L5(ORDER, COLOR, FP, true);  \
    else                     \
      L5(ORDER, COLOR, FP, false); \
  } while (0)
#define L2(ORDER, COLOR) \
  do {                   \
    if (fastproj)        \
      L4(ORDER, COLOR, true);  \
    else                 \
      L4(ORDER, COLOR, false); \
  } while (0)
    // the software-pipelined instance (k_integrate_p): ALLIN, PACKED, no colour, every block's rows present, resting hinge
    // (knob pipe: bit 0 = without colour, k_integrate_p; bit 1 = with colour, k_integrate_pc)
    const bool pipe = (tsdf_tuning().pipe & (color ? 2 : 1)) && fastproj && allin && !live && h->packed && a.hinge_fixed && a.neg_in_window &&
                      a.ny % a.TY == 0;
    if (pipe) {
      h->last_launch[0] |= 0x100;  // bit 8: the pipelined row loop
#define LAUNCH_P(ORDER, COUNT)                                                                                                           \
  do {                                                                                                                                   \
    if (color)                                                                                                                           \
      hipLaunchKernelGGL((k_integrate_pc<ORDER, COUNT>), grid, block, 0, h->stream, a, D, RGB, d_depth, h->cam64, ctrx, ctry, h->ctr[2], \
                         h->counter, band_arg);                                                                                          \
    else                                                                                                                                 \
      hipLaunchKernelGGL((k_integrate_p<ORDER, COUNT>), grid, block, 0, h->stream, a, D, K8, d_depth, h->cam64, ctrx, ctry, h->ctr[2],   \
                         h->counter, band_arg);                                                                                          \
  } while (0)
      if (p.xform_order == TSDF_XFORM_PCL_SSE) {
        if (count)
          LAUNCH_P(TSDF_XFORM_PCL_SSE, true);
        else
          LAUNCH_P(TSDF_XFORM_PCL_SSE, false);
      } else {
        if (count)
          LAUNCH_P(TSDF_XFORM_LEFT_TO_RIGHT, true);
        else
          LAUNCH_P(TSDF_XFORM_LEFT_TO_RIGHT, false);
      }
#undef LAUNCH_P
    } else if (p.xform_order == TSDF_XFORM_PCL_SSE) {
      if (color)
        L2(TSDF_XFORM_PCL_SSE, true);
      else
        L2(TSDF_XFORM_PCL_SSE, false);
    } else {
      if (color)
        L2(TSDF_XFORM_LEFT_TO_RIGHT, true);
      else
        L2(TSDF_XFORM_LEFT_TO_RIGHT, false);
    }
#undef L2
#undef L4
#undef L5
#undef L6
#undef LAUNCH
    TSDF_HIP_TRY(hipGetLastError());
#if TSDF_PHASE_TIMER
    if (const char *pf = getenv("TSDF_HIP_PHASE_FILE")) {  // diagnostic build: the phase accumulators of THIS launch
      TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
      static unsigned long long host_ph[1024 * TSDF_NPHASE];
      TSDF_HIP_TRY(hipMemcpyFromSymbol(host_ph, HIP_SYMBOL(g_phase), sizeof host_ph));
      unsigned long long sum[TSDF_NPHASE] = {0};
      for (int i = 0; i < 1024; ++i)
        for (int k = 0; k < TSDF_NPHASE; ++k) sum[k] += host_ph[i * TSDF_NPHASE + k];
      memset(host_ph, 0, sizeof host_ph);
      TSDF_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), host_ph, sizeof host_ph));
      if (FILE *f = fopen(pf, "a")) {
        fprintf(f, "{\"color\": %d, \"count\": %d, \"allin\": %d, \"live\": %d, \"packed\": %d, \"blocks\": %d, \"phase\": [", (int)color, (int)count,
                (int)(fastproj && allin && !live), (int)(live != nullptr), (int)h->packed, h->last_launch[3]);
        for (int k = 0; k < TSDF_NPHASE; ++k) fprintf(f, "%llu%s", sum[k], k + 1 < TSDF_NPHASE ? ", " : "]}\n");
        fclose(f);
      }
    }
#endif
  }
  h->count_slots = count ? 4096 : 0;  // slots 1024.. hold the bytes of voxel words whose value changed, 2048.. the observed
                                      // voxels whose distance word was not read, 3072.. the plane bytes requested
  h->count_ran = pose_ok && !nothing_observable;
  return TSDF_HIP_OK;
}

// Synchronising half: reads the counters of the last tsdf_integrate_launch(count = true) on this handle.  The plain
// kernels (weight_by_depth, RGB_NORMALIZED, LAB) do not count changed bytes: last_changed_bytes is then 0, never a
// stale figure of an earlier fast-kernel call.
int tsdf_integrate_collect(tsdf_handle h, uint64_t *n_observed) {
  if (!h->count_slots) {
    tsdf_set_error("tsdf_integrate_collect without a counting launch");
    return TSDF_HIP_E_INVALID;
  }
  unsigned long long c[4096];
  const int slots = h->count_slots;
  TSDF_HIP_TRY(hipMemcpyAsync(c, h->counter, (size_t)slots * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  unsigned long long sum = 0, changed = 0, implied = 0, read_bytes = 0;
  for (int i = 0; i < 1024; ++i) sum += c[i];
  for (int i = 1024; i < slots && i < 2048; ++i) changed += c[i];
  for (int i = 2048; i < slots && i < 3072; ++i) implied += c[i];
  for (int i = 3072; i < slots; ++i) read_bytes += c[i];
  h->last_observed = h->count_ran ? sum : 0;
  h->last_changed_bytes = h->count_ran ? changed : 0;
  h->last_implied = h->count_ran ? implied : 0;
  h->last_read_bytes = h->count_ran ? read_bytes : 0;
  h->count_slots = 0;
  if (n_observed) *n_observed = h->last_observed;
  return TSDF_HIP_OK;
}

static int launch_integrate(tsdf_handle h, const float *d_depth, const uint32_t *d_bgra, const float T[12],
                            uint64_t *n_observed) {
  const int rc = tsdf_integrate_launch(h, d_depth, d_bgra, T, n_observed != nullptr);
  if (rc || !n_observed) return rc;
  return tsdf_integrate_collect(h, n_observed);
}

// Two frames in one sweep (k_integrate2) when the handle and BOTH poses qualify, else two ordinary launches in order.
// planesA / planesB: the reference cull's planes of each frame (NULL = none), as tsdf_hip_set_reference_cull takes them;
// the handle keeps frame B's afterwards.  *fused says which way it went.  With `count`, tsdf_integrate_collect2 reads
// the per-frame observation counts (identical to two separate launches) and the detail of the pair.
static bool fusable_pose(tsdf_handle h, const IntegrateHost &hh, const float T[12], const float *planes) {
  for (int i = 0; i < 12; ++i)
    if (!(std::isfinite(T[i]) && fabsf(T[i]) <= 1e15f)) return false;
  if (planes && !reference_cull_keeps_whole_slab(h, planes)) return false;
  return slab_all_inside(h, T) && zlo_margin_ok(T, h) && fast_projection_ok(hh, h->p.integrate_color != 0);
}

int tsdf_integrate_launch2(tsdf_handle h, const float *dA, const uint32_t *cA, const float TA[12], const float *planesA_,
                           const float *dB, const uint32_t *cB, const float TB[12], const float *planesB_, bool count, bool *fused) {
  // The callers' plane arrays may BE the handle's own (frame pairing hands frame B's planes over as h->cull_planes), and
  // the two-launch path below rewrites that array for frame A before frame B is launched: work on copies.
  float planes_copy[2][24];
  const float *planesA = nullptr, *planesB = nullptr;
  if (planesA_) memcpy(planes_copy[0], planesA_, sizeof planes_copy[0]), planesA = planes_copy[0];
  if (planesB_) memcpy(planes_copy[1], planesB_, sizeof planes_copy[1]), planesB = planes_copy[1];
  const tsdf_params &p = h->p;
  const bool color = p.integrate_color != 0;
  if (color && (!cA || !cB)) {
    tsdf_set_error("integrate_color is set but no colour image was given");
    return TSDF_HIP_E_INVALID;
  }
  const IntegrateHost hh = make_args(h, TA);
  IntegrateArgs a = hh.a;
  const size_t npx = (size_t)p.image_width * p.image_height;
  auto bgra_offset = [&](const float *d, const uint32_t *c, unsigned *off) {  // the colour image behind the depth image, within 2 GB
    *off = 0;
    if (!color) return true;
    const char *zd = (const char *)d, *cb = (const char *)c;
    if (!(cb >= zd + npx * 4 && (size_t)(cb - zd) + npx * 4 < (1ull << 31))) return false;
    *off = (unsigned)(cb - zd);
    return true;
  };
  Frame2 fb;
  for (int i = 0; i < 12; ++i) fb.m[i] = TB[i];
  const unsigned gx = (unsigned)((a.qpr + a.TX - 1) / a.TX), gy = (unsigned)((a.ny + a.rpb * a.TY - 1) / (a.rpb * a.TY)),
                 gz = (unsigned)hh.planes;
  // Without colour two frames through the pipelined single-frame kernel beat one shared sweep since round 6 (k_integrate_p 7.1 ms per
  // frame against k_integrate2's 7.6 at 2048^3: profiles/r06_cpp_path_timing.json), so knob fuse2 = 1 (the default) pairs only where
  // the sweep wins: with colour, or where k_integrate_p does not apply; fuse2 = 2 always shares the sweep (the tests' k_integrate2
  // coverage), 0 never.
  const bool pipe_wins = !color && (tsdf_tuning().pipe & 1) && tsdf_tuning().fuse2 == 1 && a.hinge_fixed && a.neg_in_window && a.ny % a.TY == 0;
  const bool ok = tsdf_tuning().fuse2 && !pipe_wins && tsdf_tuning().cull == 1 && tsdf_tuning().allin && h->packed && !h->cn[0] && !h->weight_by_depth &&
                  !h->weight_by_variance && !tsdf_tuning().plain_kernel && (h->nx & 3) == 0 && a.wmax_is_int && (float)h->kmax == p.max_weight &&
                  gy <= 65535u && gz <= 65535u && gz > 0 && bgra_offset(dA, cA, &a.bgra_off) && bgra_offset(dB, cB, &fb.bgra_off) &&
                  fusable_pose(h, hh, TA, planesA) && fusable_pose(h, hh, TB, planesB);
  if (fused) *fused = ok;
  if (!ok) {  // two launches, each with its own planes; with `count` the first one's counters are read before the second runs
    int rc = tsdf_hip_set_reference_cull(h, planesA);
    if (!rc) rc = tsdf_integrate_launch(h, dA, cA, TA, count);
    if (!rc && count) {
      uint64_t n = 0;
      rc = tsdf_integrate_collect(h, &n);
      h->pair_first_observed = n, h->pair_first_changed = h->last_changed_bytes, h->pair_first_implied = h->last_implied;
      h->pair_first_read = h->last_read_bytes;
    }
    if (!rc) rc = tsdf_hip_set_reference_cull(h, planesB);
    if (!rc) rc = tsdf_integrate_launch(h, dB, cB, TB, count);
    h->pair_fused = false;
    return rc;
  }
  h->ref_cull = planesB != nullptr;
  for (int i = 0; i < 24; ++i) h->cull_planes[i] = planesB ? planesB[i] : 0.f;
  h->count_slots = 0;
  h->count_ran = false;
  if (count) TSDF_HIP_TRY(hipMemsetAsync(h->counter, 0, 3072 * sizeof(unsigned long long), h->stream));
  uint8_t *band_arg = h->band_exact && ((a.rpb * a.TY) & 3) == 0 ? h->band : nullptr;
  if (!band_arg) h->band_exact = false;
  (void)implied_distances(h, a, band_arg != nullptr);  // the record every flag-keeping launch keeps (rest_state); k_integrate2
  a.implied_d = 0;                                      // itself reads every distance word (see the kernel)
  h->last_implied_on = false;
  h->last_launch[0] = 2, h->last_launch[1] = 1, h->last_launch[2] = 0;
  h->last_launch[3] = (int)std::min<uint64_t>((uint64_t)gx * gy * gz, 0x7fffffffu);
  a.zfast = tsdf_tuning().zfast < 0 ? (npx * (color ? 8 : 4) > (3u << 20) && gx <= 65535u) : (tsdf_tuning().zfast != 0 && gx <= 65535u);
  const dim3 grid(a.zfast ? gz : gx, gy, a.zfast ? gx : gz), block(256);
#define LAUNCH2(ORDER, COLOR, COUNT)                                                                                    \
  hipLaunchKernelGGL((k_integrate2<ORDER, COLOR, COUNT>), grid, block, 0, h->stream, a, fb, h->d, h->rgb, h->k8, dA, dB, \
                     h->cam64, h->ctr[0], h->ctr[1], h->ctr[2], h->counter, band_arg)
#define L2B(ORDER, COLOR)         \
  do {                            \
    if (count)                    \
      LAUNCH2(ORDER, COLOR, true);  \
    else                          \
      LAUNCH2(ORDER, COLOR, false); \
  } while (0)
  if (p.xform_order == TSDF_XFORM_PCL_SSE) {
    if (color)
      L2B(TSDF_XFORM_PCL_SSE, true);
    else
      L2B(TSDF_XFORM_PCL_SSE, false);
  } else {
    if (color)
      L2B(TSDF_XFORM_LEFT_TO_RIGHT, true);
    else
      L2B(TSDF_XFORM_LEFT_TO_RIGHT, false);
  }
#undef L2B
#undef LAUNCH2
  TSDF_HIP_TRY(hipGetLastError());
  h->count_slots = count ? 3072 : 0;
  h->count_ran = true;
  h->pair_fused = true;
  return TSDF_HIP_OK;
}

// Counters of the last tsdf_integrate_launch2(count = true): n_observed[0 / 1] = voxels frame A / B brought to
// addObservation (what two separate launches report); the handle's last_count_detail becomes the PAIR's: voxels
// observed by at least one of the two frames, bytes of voxel words whose value changed over the pair.
int tsdf_integrate_collect2(tsdf_handle h, uint64_t n_observed[2]) {
  if (!h->pair_fused) {  // two launches: the second one's counters are still pending
    uint64_t nb = 0;
    const int rc = tsdf_integrate_collect(h, &nb);
    if (rc) return rc;
    n_observed[0] = h->pair_first_observed, n_observed[1] = nb;
    h->last_observed = h->pair_first_observed + nb;  // (an upper bound of the union; the words were read twice anyway)
    h->last_changed_bytes += h->pair_first_changed;
    h->last_implied += h->pair_first_implied;
    h->last_read_bytes += h->pair_first_read;
    return TSDF_HIP_OK;
  }
  if (!h->count_slots) {
    tsdf_set_error("tsdf_integrate_collect2 without a counting launch");
    return TSDF_HIP_E_INVALID;
  }
  unsigned long long c[3072];
  TSDF_HIP_TRY(hipMemcpyAsync(c, h->counter, sizeof c, hipMemcpyDeviceToHost, h->stream));
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  unsigned long long sum[6] = {0, 0, 0, 0, 0, 0};  // A, B, union, changed bytes, distance words not read, plane bytes requested: 512 striped slots each
  for (int i = 0; i < 3072; ++i) sum[i >> 9] += c[i];
  n_observed[0] = sum[0], n_observed[1] = sum[1];
  h->last_observed = sum[2];
  h->last_changed_bytes = sum[3];
  h->last_implied = sum[4];
  h->last_read_bytes = sum[5];
  h->count_slots = 0;
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_integrate_device2(tsdf_handle h, const float *d_depth_a, const uint32_t *d_bgra_a, const float cam_from_vol_a[12],
                                          const float *planes_a, const float *d_depth_b, const uint32_t *d_bgra_b,
                                          const float cam_from_vol_b[12], const float *planes_b, uint64_t *n_observed, int32_t *fused) {
  if (!h || !d_depth_a || !d_depth_b || !cam_from_vol_a || !cam_from_vol_b) return TSDF_HIP_E_INVALID;
  if (fused) *fused = 0;
  if (h->multi)  // a multi-GPU set: every slab takes both frames and sweeps ONCE where both poses see all of it (round 6)
    return tsdf_multi_integrate_device2(h, d_depth_a, d_bgra_a, cam_from_vol_a, planes_a, d_depth_b, d_bgra_b, cam_from_vol_b, planes_b,
                                        n_observed, fused);
  TSDF_ENTER(h);
  bool f = false;
  const int rc = tsdf_integrate_launch2(h, d_depth_a, d_bgra_a, cam_from_vol_a, planes_a, d_depth_b, d_bgra_b, cam_from_vol_b, planes_b,
                                        n_observed != nullptr, &f);
  if (fused) *fused = f ? 1 : 0;
  if (rc || !n_observed) return rc;
  return tsdf_integrate_collect2(h, n_observed);
}

// The measured side of the roofline's algorithmic bytes (bench.py): of the last integrate call that asked for
// n_observed (fast kernel only), out[0] = voxels that reached addObservation, out[1] = bytes of voxel words whose
// value changed (4 per d / w / colour word, 1 per count byte).
extern "C" int tsdf_hip_last_count_detail(tsdf_handle h, uint64_t out[2]) {
  if (!h || !out) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_last_count_detail(h, out);
  out[0] = h->last_observed;
  out[1] = h->last_changed_bytes;
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_last_read_detail(tsdf_handle h, uint64_t out[3]) {
  if (!h || !out) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_last_read_detail(h, out);
  out[0] = h->last_implied;
  out[1] = h->last_implied_on ? 1 : 0;
  out[2] = h->last_read_bytes;
  return TSDF_HIP_OK;
}

// Replication mode for the reference's frustum cull (VERDICT r02 missing #3).  The six planes are PCL / Eigen arithmetic
// on the forward pose, so the caller computes them (the C++ shell and the Python binding both do, exactly as
// pcl::FrustumCulling::applyFilter); they apply to every integrate call until cleared with NULL.
extern "C" int tsdf_hip_set_reference_cull(tsdf_handle h, const float planes[24]) {
  if (!h) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_set_reference_cull(h, planes);
  h->ref_cull = planes != nullptr;
  for (int i = 0; i < 24; ++i) h->cull_planes[i] = planes ? planes[i] : 0.f;
  return TSDF_HIP_OK;
}

// Test / report hook: which instance the last integrate launch on this handle (slab 0 of a set) went through.
extern "C" int tsdf_hip_last_launch_info(tsdf_handle h, int32_t out[4]) {
  if (!h || !out) return TSDF_HIP_E_INVALID;
  if (h->multi) h = tsdf_multi_first(h);
  for (int i = 0; i < 4; ++i) out[i] = h->last_launch[i];
  return TSDF_HIP_OK;
}

// Which expf does this host's libm run?  glibc's FMA build (picked by ifunc on CPUs with FMA) fuses r = InvLn2N * x - k;
// the two builds disagree on exactly one float of +-(2^-26 .. 104) (exhaustive search on the build host), which is the probe.
static int host_expf_fuses_r() {
  volatile float probe = -0x1.f8cbb2p+5f;
  return expf(probe) == 0x1.f45326p-92f ? 1 : 0;  // the unfused form (and the correctly rounded value) is 0x1.f45324p-92
}

// tsdf_expf_glibc on n host floats (the current device).
static __global__ void k_expf(const float *__restrict__ in, float *__restrict__ out, size_t n, int fused_r) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = tsdf_expf_glibc(in[i], fused_r != 0);
}
static int device_expf(const float *in, size_t n, float *out, int fused_r) {
  float *di = nullptr, *dout = nullptr;
  TSDF_HIP_TRY(hipMalloc(&di, n * 4));
  TSDF_HIP_TRY(hipMalloc(&dout, n * 4));
  TSDF_HIP_TRY(hipMemcpy(di, in, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_expf, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, di, dout, n, fused_r);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(di);
  (void)hipFree(dout);
  return TSDF_HIP_OK;
}

// The device's expf is a restatement of ONE libm (glibc >= 2.27's table algorithm, in the flavour the probe above selects).
// On another libm -- musl, an older glibc, a vectorised build -- the reference's std::exp(float) is a different function and
// weight_by_variance_ would silently drift from the host's results.  So the first volume that switches the weighting on
// checks: 4096 floats spread over the whole argument range, the device against THIS host's expf, bit for bit.
static int expf_matches_this_host(int fused_r) {
  static int verdict = -1;  // -1 unknown, 0 differs, 1 equal
  if (verdict >= 0) return verdict;
  const size_t n = 4096;
  std::vector<float> in(n), dev(n);
  uint32_t s = 0x9e3779b9u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    const float u = (float)(s >> 8) * (1.0f / 16777216.0f);                  // [0, 1)
    const float mag = ldexpf(1.0f + u, (int)(i % 34) - 26);                    // 2^-26 .. 2^7.99: the weighting's exponents are <= 0,
    in[i] = (i & 1) ? mag : -mag;                                              // the positive half guards the shared code path
  }
  if (device_expf(in.data(), n, dev.data(), fused_r) != TSDF_HIP_OK) return -1;
  verdict = 1;
  for (size_t i = 0; i < n; ++i) {
    const volatile float x = in[i];
    const float want = expf(x);
    if (memcmp(&want, &dev[i], 4) != 0 && !(std::isnan(want) && std::isnan(dev[i]))) verdict = 0;
  }
  return verdict;
}

extern "C" int tsdf_hip_set_weighting(tsdf_handle h, int weight_by_depth, int weight_by_variance) {
  if (!h) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_set_weighting(h, weight_by_depth, weight_by_variance);
  if (weight_by_depth && (h->packed || h->cn[0])) {
    tsdf_set_error("weight_by_depth makes weights non-integer: it needs the F32W layout (and TSDF_COLOR_RGB)");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  if (weight_by_variance && (h->packed || h->cn[0])) {
    tsdf_set_error("weight_by_variance needs float weights and TSDF_COLOR_RGB: create / load the volume with TSDF_LAYOUT_F32W");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  TSDF_ENTER(h);  // a frame that pairing holds back was committed under the OLD weighting: it is launched first
  if (weight_by_variance && !h->vm) {  // OctreeNode::M_ / nsample_ per voxel, zero like a fresh octree's
    const size_t n = (size_t)(h->pitch * h->ny * h->nz_alloc);
    TSDF_HIP_TRY(hipMalloc(&h->vm, n * sizeof(float)));
    TSDF_HIP_TRY(hipMalloc(&h->vn, n * sizeof(int32_t)));
    TSDF_HIP_TRY(hipMemsetAsync(h->vm, 0, n * sizeof(float), h->stream));
    TSDF_HIP_TRY(hipMemsetAsync(h->vn, 0, n * sizeof(int32_t), h->stream));
  }
  h->expf_fused_r = host_expf_fuses_r();
  if (weight_by_variance) {
    TSDF_ON_DEVICE(h->device);
    if (expf_matches_this_host(h->expf_fused_r) == 0) {
      tsdf_set_error("weight_by_variance_: this host's expf is not the glibc algorithm the device restates (4096-float spot check "
                     "differs): integrating would not reproduce the reference's std::exp(float) on this machine");
      return TSDF_HIP_E_UNSUPPORTED;
    }
  }
  h->weight_by_depth = weight_by_depth != 0;
  h->weight_by_variance = weight_by_variance != 0;
  return TSDF_HIP_OK;
}

#ifdef TSDF_HIP_TEST_HOOKS
// Test hook: the device's std::exp(float) of the variance weighting (tsdf_expf_glibc in this host's flavour) on n floats.
extern "C" int tsdf_hip_selftest_expf(const float *in, size_t n, float *out) {
  if (!in || !out || !n) return TSDF_HIP_E_INVALID;
  return device_expf(in, n, out, host_expf_fuses_r());
}
#endif  // TSDF_HIP_TEST_HOOKS

extern "C" int tsdf_hip_integrate_device(tsdf_handle h, const float *d_depth, const uint32_t *d_bgra,
                                         const float cam_from_vol[12], uint64_t *n_observed) {
  if (!h || !d_depth || !cam_from_vol) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_integrate_device(h, d_depth, d_bgra, cam_from_vol, n_observed);
  TSDF_ENTER(h);
  return launch_integrate(h, d_depth, d_bgra, cam_from_vol, n_observed);
}

extern "C" int tsdf_hip_integrate(tsdf_handle h, const float *depth, const uint8_t *bgra,
                                  const float cam_from_vol[12], uint64_t *n_observed) {
  if (!h || !depth || !cam_from_vol) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_integrate(h, depth, bgra, cam_from_vol, n_observed, false);
  TSDF_ENTER(h);
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  const bool color = h->p.integrate_color != 0;
  if (color && !bgra) {
    tsdf_set_error("integrate_color is set but no colour image was given");
    return TSDF_HIP_E_INVALID;
  }
  // through the pinned bounce buffer (tsdf_to_device): the same cost whatever memory the caller hands over
  int rc = tsdf_to_device(h, h->frame_depth, depth, npx * sizeof(float));
  if (rc) return rc;
  if (color && (rc = tsdf_to_device(h, h->frame_bgra, bgra, npx * 4))) return rc;
  rc = launch_integrate(h, h->frame_depth, color ? h->frame_bgra : nullptr, cam_from_vol, n_observed);
  if (rc) return rc;
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  return TSDF_HIP_OK;
}

// Pipelined host entry point: the caller's frame is copied into one of two pinned staging slots and the call
// returns; the host-to-device copy runs on a private copy stream while the previous frame's kernel is still
// busy, and the kernel of this frame waits for the copy through an event.  A stream of host frames then runs
// at the kernel's rate instead of kernel + upload + synchronise.  Results are identical to tsdf_hip_integrate;
// every other entry point is ordered after it on the handle's stream; errors of the asynchronous part surface
// at the next synchronising call (tsdf_hip_synchronize, download, march, ...).
struct tsdf_hip_pipeline {
  static const int SLOTS = 4;  // two frames may wait for their shared sweep while the next two are being staged
  hipStream_t copy_stream = nullptr;
  float *pinned[SLOTS] = {nullptr, nullptr, nullptr, nullptr};   // host, [depth | bgra]
  float *device[SLOTS] = {nullptr, nullptr, nullptr, nullptr};   // device, [depth | bgra]
  hipEvent_t copied[SLOTS] = {nullptr, nullptr, nullptr, nullptr}, consumed[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long frames = 0;
  // frame pairing (tsdf_hip_set_frame_pairing): a committed frame whose kernel launch waits for a partner
  bool pairing = false;
  int pend_slot = 0;
  float pend_T[12];
  bool pend_has_planes = false;
  float pend_planes[24];
};

void tsdf_pipeline_destroy(tsdf_hip_volume *v) {
  tsdf_hip_pipeline *p = v->pipe;
  if (!p) return;
  for (int i = 0; i < tsdf_hip_pipeline::SLOTS; ++i) {
    if (p->consumed[i]) (void)hipEventSynchronize(p->consumed[i]);
    if (p->pinned[i]) (void)hipHostFree(p->pinned[i]);
    if (p->device[i]) (void)hipFree(p->device[i]);
    if (p->copied[i]) (void)hipEventDestroy(p->copied[i]);
    if (p->consumed[i]) (void)hipEventDestroy(p->consumed[i]);
  }
  if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
  delete p;
  v->pipe = nullptr;
  v->pair_pending = false;
}

// The ring itself: tsdf_hip_frame_begin hands out the next pinned slot (waiting until the kernel that last read it
// has finished), tsdf_hip_frame_commit uploads it on the private copy stream and queues the integrate launch behind
// the upload.  A caller that can write its frame straight into the slot (the C++ integrateCloud template stripping a
// PCL cloud) saves the intermediate copy; tsdf_hip_integrate_async is begin + memcpy + commit.
static int pipeline_ready(tsdf_handle h) {
  if (h->pipe) return TSDF_HIP_OK;
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  tsdf_hip_pipeline *p = new tsdf_hip_pipeline;
  // the ring is attached to the handle only once every allocation has succeeded: a half-built one would hand NULL
  // slot pointers to the next tsdf_hip_frame_begin
  auto build = [&]() -> int {
    TSDF_HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < tsdf_hip_pipeline::SLOTS; ++i) {
      TSDF_HIP_TRY(hipHostMalloc((void **)&p->pinned[i], npx * 8, hipHostMallocPortable));
      TSDF_HIP_TRY(hipMalloc((void **)&p->device[i], npx * 8));
      TSDF_HIP_TRY(hipEventCreateWithFlags(&p->copied[i], hipEventDisableTiming));
      TSDF_HIP_TRY(hipEventCreateWithFlags(&p->consumed[i], hipEventDisableTiming));
    }
    return TSDF_HIP_OK;
  };
  const int rc = build();
  h->pipe = p;
  if (rc) tsdf_pipeline_destroy(h);  // frees what was built and detaches it
  return rc;
}

int tsdf_multi_frame_begin(tsdf_handle h, float **depth, uint8_t **bgra);
int tsdf_multi_frame_commit(tsdf_handle h, const float T[12]);

// Frame pairing launches the frame it has been holding back on its own: with that frame's pose and cull planes, the
// handle's current planes put back afterwards.  Called by every entry point that reads or writes the volume (TSDF_ENTER).
int tsdf_pipeline_flush(tsdf_hip_volume *h) {
  tsdf_hip_pipeline *p = h->pipe;
  if (!h->pair_pending || !p) {
    h->pair_pending = false;
    return TSDF_HIP_OK;
  }
  h->pair_pending = false;
  const bool color = h->p.integrate_color != 0;
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  const bool cur_has = h->ref_cull;
  float cur[24];
  for (int i = 0; i < 24; ++i) cur[i] = h->cull_planes[i];
  h->ref_cull = p->pend_has_planes;
  for (int i = 0; i < 24; ++i) h->cull_planes[i] = p->pend_has_planes ? p->pend_planes[i] : 0.f;
  const int s = p->pend_slot;
  const int rc = tsdf_integrate_launch(h, p->device[s], color ? reinterpret_cast<const uint32_t *>(p->device[s] + npx) : nullptr, p->pend_T, false);
  h->ref_cull = cur_has;
  for (int i = 0; i < 24; ++i) h->cull_planes[i] = cur[i];
  if (rc) return rc;
  TSDF_HIP_TRY(hipEventRecord(p->consumed[s], h->stream));
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_set_frame_pairing(tsdf_handle h, int on) {
  if (!h) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_set_frame_pairing(h, on);
  TSDF_ENTER(h);  // (switching it off launches what was waiting)
  const int rc = pipeline_ready(h);
  if (rc) return rc;
  h->pipe->pairing = on != 0;
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_frame_begin(tsdf_handle h, float **depth, uint8_t **bgra) {
  if (!h || !depth) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_frame_begin(h, depth, bgra);
  TSDF_ON_DEVICE(h->device);
  const int rc = pipeline_ready(h);
  if (rc) return rc;
  tsdf_hip_pipeline *p = h->pipe;
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  const int slot = (int)(p->frames % tsdf_hip_pipeline::SLOTS);
  // the slot was last used SLOTS frames ago: its kernel must be done before the staging buffers are reused
  if (p->frames >= (unsigned)tsdf_hip_pipeline::SLOTS) TSDF_HIP_TRY(hipEventSynchronize(p->consumed[slot]));
  *depth = p->pinned[slot];
  if (bgra) *bgra = h->p.integrate_color ? reinterpret_cast<uint8_t *>(p->pinned[slot] + npx) : nullptr;
  return TSDF_HIP_OK;
}

// The ring's second half, once frame `slot` is on its way into p->device[slot] and the handle's stream waits for it: launch
// it, or -- frame pairing -- hold it back for a partner / launch it together with the frame that was waiting.
static int pipeline_commit_uploaded(tsdf_handle h, int slot, const float cam_from_vol[12]) {
  tsdf_hip_pipeline *p = h->pipe;
  const bool color = h->p.integrate_color != 0;
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  auto bgra_of = [&](int s) { return color ? reinterpret_cast<const uint32_t *>(p->device[s] + npx) : nullptr; };
  if (h->pair_pending) {  // the partner has arrived: one sweep for both where the poses allow it, two launches otherwise
    h->pair_pending = false;
    const int sa = p->pend_slot;
    const int rc = tsdf_integrate_launch2(h, p->device[sa], bgra_of(sa), p->pend_T, p->pend_has_planes ? p->pend_planes : nullptr,
                                          p->device[slot], bgra_of(slot), cam_from_vol, h->ref_cull ? h->cull_planes : nullptr, false, nullptr);
    // Whatever the launch did, both ring slots are handed back in order (their `consumed` events recorded, the frame
    // counter advanced): a failure must not leave a slot whose upload or launch state the next frame_begin cannot know
    // (ADVICE r04).  A failed launch integrated NEITHER frame or only the first: the error text says a frame was lost.
    (void)hipEventRecord(p->consumed[sa], h->stream);
    (void)hipEventRecord(p->consumed[slot], h->stream);
    if (rc) {
      p->frames++;
      const std::string why = tsdf_hip_last_error();
      tsdf_set_error("paired commit failed -- the frame held back for pairing and this one are LOST (not integrated): " + why);
      return rc;
    }
  } else if (p->pairing) {  // uploaded, but its kernel waits for the next frame (or for any other call on the volume)
    p->pend_slot = slot;
    for (int i = 0; i < 12; ++i) p->pend_T[i] = cam_from_vol[i];
    p->pend_has_planes = h->ref_cull;
    for (int i = 0; i < 24; ++i) p->pend_planes[i] = h->cull_planes[i];
    h->pair_pending = true;
  } else {
    const int rc = launch_integrate(h, p->device[slot], bgra_of(slot), cam_from_vol, nullptr);
    if (rc) return rc;
    TSDF_HIP_TRY(hipEventRecord(p->consumed[slot], h->stream));
  }
  p->frames++;
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_frame_commit(tsdf_handle h, const float cam_from_vol[12]) {
  if (!h || !cam_from_vol) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_frame_commit(h, cam_from_vol);
  if (!h->pipe) {
    tsdf_set_error("tsdf_hip_frame_commit without tsdf_hip_frame_begin");
    return TSDF_HIP_E_INVALID;
  }
  TSDF_ON_DEVICE(h->device);
  tsdf_hip_pipeline *p = h->pipe;
  const bool color = h->p.integrate_color != 0;
  const size_t npx = (size_t)h->p.image_width * h->p.image_height, bytes = npx * 4 * (color ? 2 : 1);
  const int slot = (int)(p->frames % tsdf_hip_pipeline::SLOTS);
  TSDF_HIP_TRY(hipMemcpyAsync(p->device[slot], p->pinned[slot], bytes, hipMemcpyHostToDevice, p->copy_stream));
  TSDF_HIP_TRY(hipEventRecord(p->copied[slot], p->copy_stream));
  TSDF_HIP_TRY(hipStreamWaitEvent(h->stream, p->copied[slot], 0));
  return pipeline_commit_uploaded(h, slot, cam_from_vol);
}

// A SLAB of a multi-GPU set takes a frame through its own ring (tsdf_multi.hip, frame pairing on a set): the frame already
// sits in `src` -- the set's pinned host slot (src_dev < 0: one upload over this GPU's own PCIe link) or a device buffer on
// GPU src_dev (a peer copy, or the pinned relay where the driver refuses peer access: `copy`) -- and goes into the slab's next
// ring slot on the slab's copy stream; then exactly what tsdf_hip_frame_commit does.  *uploaded, when given, is recorded on
// the copy stream behind the copy (the set waits for it before it reuses its pinned slot).
int tsdf_pipeline_commit_from(tsdf_handle s, const void *src, int src_dev, const float T[12], bool pairing, hipEvent_t uploaded,
                              int (*copy)(void *ctx, void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t st), void *ctx) {
  TSDF_ON_DEVICE(s->device);
  int rc = pipeline_ready(s);
  if (rc) return rc;
  tsdf_hip_pipeline *p = s->pipe;
  if (p->pairing != pairing) {
    if (s->pair_pending && (rc = tsdf_pipeline_flush(s))) return rc;
    p->pairing = pairing;
  }
  const bool color = s->p.integrate_color != 0;
  const size_t npx = (size_t)s->p.image_width * s->p.image_height, bytes = npx * 4 * (color ? 2 : 1);
  const int slot = (int)(p->frames % tsdf_hip_pipeline::SLOTS);
  if (p->frames >= (unsigned)tsdf_hip_pipeline::SLOTS) TSDF_HIP_TRY(hipEventSynchronize(p->consumed[slot]));
  if (src_dev < 0) {
    TSDF_HIP_TRY(hipMemcpyAsync(p->device[slot], src, bytes, hipMemcpyHostToDevice, p->copy_stream));
  } else if ((rc = copy(ctx, p->device[slot], s->device, src, src_dev, bytes, p->copy_stream))) {
    return rc;
  }
  TSDF_HIP_TRY(hipEventRecord(p->copied[slot], p->copy_stream));
  if (uploaded) TSDF_HIP_TRY(hipEventRecord(uploaded, p->copy_stream));
  TSDF_HIP_TRY(hipStreamWaitEvent(s->stream, p->copied[slot], 0));
  return pipeline_commit_uploaded(s, slot, T);
}

// ... and two device frames at once (tsdf_hip_integrate_device2 on a set): both into ring slots, one tsdf_integrate_launch2.
int tsdf_pipeline_pair_from(tsdf_handle s, const void *src_a, const void *src_b, int src_dev, const float TA[12], const float *planes_a,
                            const float TB[12], const float *planes_b, bool count, bool *fused,
                            int (*copy)(void *ctx, void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t st), void *ctx) {
  TSDF_ENTER(s);  // (a frame this slab was holding back goes first)
  int rc = pipeline_ready(s);
  if (rc) return rc;
  tsdf_hip_pipeline *p = s->pipe;
  const bool color = s->p.integrate_color != 0;
  const size_t npx = (size_t)s->p.image_width * s->p.image_height, bytes = npx * 4 * (color ? 2 : 1);
  int slot[2];
  const void *src[2] = {src_a, src_b};
  for (int k = 0; k < 2; ++k) {
    slot[k] = (int)((p->frames + k) % tsdf_hip_pipeline::SLOTS);
    if (p->frames + k >= (unsigned)tsdf_hip_pipeline::SLOTS) TSDF_HIP_TRY(hipEventSynchronize(p->consumed[slot[k]]));
    if ((rc = copy(ctx, p->device[slot[k]], s->device, src[k], src_dev, bytes, p->copy_stream))) return rc;
    TSDF_HIP_TRY(hipEventRecord(p->copied[slot[k]], p->copy_stream));
    TSDF_HIP_TRY(hipStreamWaitEvent(s->stream, p->copied[slot[k]], 0));
  }
  auto bgra_of = [&](int q) { return color ? reinterpret_cast<const uint32_t *>(p->device[q] + npx) : nullptr; };
  rc = tsdf_integrate_launch2(s, p->device[slot[0]], bgra_of(slot[0]), TA, planes_a, p->device[slot[1]], bgra_of(slot[1]), TB, planes_b, count, fused);
  (void)hipEventRecord(p->consumed[slot[0]], s->stream);
  (void)hipEventRecord(p->consumed[slot[1]], s->stream);
  p->frames += 2;
  return rc;
}

extern "C" int tsdf_hip_integrate_async(tsdf_handle h, const float *depth, const uint8_t *bgra,
                                        const float cam_from_vol[12]) {
  if (!h || !depth || !cam_from_vol) return TSDF_HIP_E_INVALID;
  if (h->p.integrate_color && !bgra) {
    tsdf_set_error("integrate_color is set but no colour image was given");
    return TSDF_HIP_E_INVALID;
  }
  float *sd = nullptr;
  uint8_t *sc = nullptr;
  const int rc = tsdf_hip_frame_begin(h, &sd, &sc);
  if (rc) return rc;
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  memcpy(sd, depth, npx * 4);
  if (sc) memcpy(sc, bgra, npx * 4);
  return tsdf_hip_frame_commit(h, cam_from_vol);
}

#ifdef TSDF_HIP_TEST_HOOKS  // everything below: libtsdf_hip_test.so only (include/tsdf_hip_test.h)
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
k_calib_rmw(float *__restrict__ D, float *__restrict__ Wt, uint32_t *__restrict__ RGB, uint8_t *__restrict__ K8,
            int64_t first4, int64_t n4, uint32_t xorv /* 0 at run time */) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    // a run-time zero xor-ed into EVERY component keeps the loads and stores alive without changing a bit (an untouched
    // component would let the compiler drop its load and store); non-temporal like the integrate kernel's stream
    u4 *pd = reinterpret_cast<u4 *>(D) + first4 + i;
    __builtin_nontemporal_store(__builtin_nontemporal_load(pd) ^ xorv, pd);
    if (Wt) {
      u4 *pw = reinterpret_cast<u4 *>(Wt) + first4 + i;
      __builtin_nontemporal_store(__builtin_nontemporal_load(pw) ^ xorv, pw);
    }
    if (RGB) {
      u4 *pc = reinterpret_cast<u4 *>(RGB) + first4 + i;
      __builtin_nontemporal_store(__builtin_nontemporal_load(pc) ^ xorv, pc);
    }
    if (K8) {
      uint32_t *pk = reinterpret_cast<uint32_t *>(K8) + first4 + i;
      __builtin_nontemporal_store(__builtin_nontemporal_load(pk) ^ xorv, pk);
    }
  }
}

// Calibration of FETCH_SIZE for NARROW reads (k_calib_rmw only exercises 16 B per lane): a read-only
// sweep of the distance plane taking ONE dword per STRIDE bytes -- STRIDE 4: a wave instruction covers 256 contiguous
// bytes; 64 / 128: every lane touches its own 64 B / 128 B piece, the shape of marching cubes' lane-63 halo words.
// The xor of everything read is stored only if it equals a run-time value it never equals.
template <int STRIDE>
static __global__ void __launch_bounds__(256)
k_calib_read(const uint32_t *__restrict__ D, int64_t n, uint32_t *__restrict__ sink, uint32_t never) {
  uint32_t v = 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    v ^= D[i * (STRIDE / 4)];
  if (v == never) *sink = v;
}

extern "C" int tsdf_hip_selftest_read_sweep(tsdf_handle h, int stride_bytes, uint64_t *span_bytes, uint64_t *dwords_read) {
  if (!h) return TSDF_HIP_E_INVALID;
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_selftest_read_sweep");
  TSDF_ENTER(h);
  const int64_t plane = h->pitch * h->ny;
  const uint32_t *base = reinterpret_cast<const uint32_t *>(h->d) + (int64_t)(h->z_begin - h->z_first) * plane;
  const int64_t span = (int64_t)(h->z_end - h->z_begin) * plane * 4;
  const int64_t n = span / stride_bytes;
  const unsigned grid = 256u * (unsigned)tsdf_tuning().blocks_per_cu;
  uint32_t *sink = reinterpret_cast<uint32_t *>(h->counter + 2047);
  const uint32_t never = 0x7fc5a5a5u;
  switch (stride_bytes) {
  case 4: hipLaunchKernelGGL(k_calib_read<4>, dim3(grid), dim3(256), 0, h->stream, base, n, sink, never); break;
  case 64: hipLaunchKernelGGL(k_calib_read<64>, dim3(grid), dim3(256), 0, h->stream, base, n, sink, never); break;
  case 128: hipLaunchKernelGGL(k_calib_read<128>, dim3(grid), dim3(256), 0, h->stream, base, n, sink, never); break;
  default: tsdf_set_error("tsdf_hip_selftest_read_sweep: stride must be 4, 64 or 128"); return TSDF_HIP_E_INVALID;
  }
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  if (span_bytes) *span_bytes = (uint64_t)n * (uint64_t)stride_bytes;
  if (dwords_read) *dwords_read = (uint64_t)n;
  return TSDF_HIP_OK;
}

// Test hook, host only (no device needed): the index box launch_integrate would restrict a frame's launch to.
// rc 0 and box = {lo x,y,z, hi x,y,z} (inclusive); *state = 0 box valid, 1 nothing observable, 2 no claim.
extern "C" int tsdf_hip_selftest_index_box(const tsdf_params *p, const float cam_from_vol[12], int32_t box[6], int32_t *state) {
  if (!p || !cam_from_vol || !box || !state) return TSDF_HIP_E_INVALID;
  tsdf_hip_volume v;
  v.p = *p;
  for (int a = 0; a < 3; ++a) {
    if (p->res[a] <= 0 || !(p->size[a] > 0.f)) return TSDF_HIP_E_INVALID;
    tsdf_build_centers(p->res[a], tsdf_node_size(*p, a), v.h_ctr[a], &v.levels[a]);
  }
  int lo[3], hi[3];
  bool empty = false;
  if (!observable_index_box(&v, cam_from_vol, lo, hi, &empty)) {
    *state = 2;
    return TSDF_HIP_OK;
  }
  *state = empty ? 1 : 0;
  for (int a = 0; a < 3; ++a) {
    box[a] = lo[a];
    box[3 + a] = hi[a];
  }
  return TSDF_HIP_OK;
}

// Test hook, host only: k_cull's predicate (box_may_be_observed) for every block of the WHOLE grid, blocks of
// bx_vox voxels along x by by_rows rows of one plane; flags[(z * gy + by) * gx + bx] = 1 if the block may hold an
// observable voxel.  gx = ceil(nx / bx_vox), gy = ceil(ny / by_rows).
extern "C" int tsdf_hip_selftest_block_flags(const tsdf_params *p, const float cam_from_vol[12], int bx_vox, int by_rows,
                                             uint8_t *flags) {
  if (!p || !cam_from_vol || !flags || bx_vox < 1 || by_rows < 1) return TSDF_HIP_E_INVALID;
  std::vector<float> ctr[3];
  for (int a = 0; a < 3; ++a) {
    int levels;
    if (p->res[a] <= 0 || !(p->size[a] > 0.f)) return TSDF_HIP_E_INVALID;
    tsdf_build_centers(p->res[a], tsdf_node_size(*p, a), ctr[a], &levels);
  }
  CullArgs c;
  for (int i = 0; i < 12; ++i) c.m[i] = cam_from_vol[i];
  c.fx = p->fx, c.fy = p->fy, c.cx = p->cx, c.cy = p->cy;
  c.zlo = p->min_sensor_dist > 0 ? p->min_sensor_dist : 0;
  c.zmax = p->max_sensor_dist;
  c.W = p->image_width, c.H = p->image_height, c.nx = p->res[0], c.ny = p->res[1], c.z_global0 = 0;
  c.bx_vox = bx_vox, c.by_rows = by_rows;
  c.gx = (p->res[0] + bx_vox - 1) / bx_vox, c.gy = (p->res[1] + by_rows - 1) / by_rows, c.gz = p->res[2];
  for (int bz = 0; bz < c.gz; ++bz)
    for (int by = 0; by < c.gy; ++by)
      for (int bx = 0; bx < c.gx; ++bx) {  // (the same index arithmetic as k_cull)
        const int xa = bx * c.bx_vox, xb = std::min(c.nx, xa + c.bx_vox) - 1;
        const int ya = by * c.by_rows, yb = std::min(c.ny, ya + c.by_rows) - 1;
        flags[((size_t)bz * c.gy + by) * c.gx + bx] =
            box_may_be_observed(c, ctr[0][xa], ctr[0][xb], ctr[1][ya], ctr[1][yb], ctr[2][bz]) ? 1 : 0;
      }
  return TSDF_HIP_OK;
}

// Test hook, host only: k_rows' row intervals (row_interval) for every voxel row of the WHOLE grid,
// words[z * res_y + y] = lo | len << 16 (an empty row: lo = 0xffff); planes = the reference cull's six planes or NULL.
extern "C" int tsdf_hip_selftest_row_intervals(const tsdf_params *p, const float cam_from_vol[12], const float *planes,
                                               uint32_t *words) {
  if (!p || !cam_from_vol || !words) return TSDF_HIP_E_INVALID;
  std::vector<float> ctr[3];
  for (int a = 0; a < 3; ++a) {
    int levels;
    if (p->res[a] <= 0 || !(p->size[a] > 0.f)) return TSDF_HIP_E_INVALID;
    tsdf_build_centers(p->res[a], tsdf_node_size(*p, a), ctr[a], &levels);
  }
  if (p->res[0] > 0xfff0) return TSDF_HIP_E_UNSUPPORTED;
  RowArgs c;
  for (int i = 0; i < 12; ++i) c.m[i] = cam_from_vol[i];
  c.fx = p->fx, c.fy = p->fy, c.cx = p->cx, c.cy = p->cy;
  c.zlo = p->min_sensor_dist > 0 ? p->min_sensor_dist : 0;
  c.zmax = p->max_sensor_dist;
  c.W = p->image_width, c.H = p->image_height, c.nx = p->res[0], c.ny = p->res[1], c.nz = p->res[2], c.z_global0 = 0;
  c.rc = planes ? 1 : 0;
  for (int i = 0; i < 24; ++i) c.cull[i] = planes ? planes[i] : 0.f;
  for (int z = 0; z < c.nz; ++z)
    for (int y = 0; y < c.ny; ++y) words[(size_t)z * c.ny + y] = row_interval(c, ctr[0].data(), ctr[1][y], ctr[2][z]);
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_selftest_sweep(tsdf_handle h, uint64_t *bytes_read, uint64_t *bytes_written) {
  if (!h) return TSDF_HIP_E_INVALID;
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_selftest_sweep");
  TSDF_ENTER(h);
  const int64_t plane = h->pitch * h->ny;
  const int64_t first4 = (int64_t)(h->z_begin - h->z_first) * plane / 4;
  const int64_t n4 = (int64_t)(h->z_end - h->z_begin) * plane / 4;
  const unsigned grid = 256u * (unsigned)tsdf_tuning().blocks_per_cu;
  hipLaunchKernelGGL(k_calib_rmw, dim3(grid), dim3(256), 0, h->stream, h->d, h->w, h->rgb, h->k8, first4, n4, 0u);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  const uint64_t bytes_per_quad = 16u + (h->w ? 16u : 0u) + (h->rgb ? 16u : 0u) + (h->k8 ? 4u : 0u);
  if (bytes_read) *bytes_read = bytes_per_quad * (uint64_t)n4;
  if (bytes_written) *bytes_written = bytes_per_quad * (uint64_t)n4;
  return TSDF_HIP_OK;
}

// Test hook: the PACKED update's divider -- numerator a over an integer count k in [1, 256] through the refined
// table reciprocal and the scale-free ladder, accepted when the RESULT is a normal number, else IEEE division.
static __global__ void k_selftest_div_count(const float *a, const uint32_t *k, float *out, unsigned char *fast, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float b = (float)k[i];
  Rcp32 rs;
  rs.nb = -b;
  rs.y = rcp32_prepare(b).y;  // what s_rcp[k - 1] holds
  const float q = div32_fast(a[i], rs);
  const bool ok = __builtin_amdgcn_classf(q, 0x108);
  fast[i] = ok ? 1 : 0;
  out[i] = ok ? q : a[i] / b;
}

extern "C" int tsdf_hip_selftest_div_count(const float *a, const uint32_t *k, float *out, uint8_t *fast, size_t n) {
  if (!a || !k || !out || !fast || !n) return TSDF_HIP_E_INVALID;
  float *da = nullptr, *dout = nullptr;
  uint32_t *dk = nullptr;
  unsigned char *df = nullptr;
  TSDF_HIP_TRY(hipMalloc(&da, n * 4));
  TSDF_HIP_TRY(hipMalloc(&dk, n * 4));
  TSDF_HIP_TRY(hipMalloc(&dout, n * 4));
  TSDF_HIP_TRY(hipMalloc(&df, n));
  TSDF_HIP_TRY(hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice));
  TSDF_HIP_TRY(hipMemcpy(dk, k, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest_div_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da, dk, dout, df, n);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
  TSDF_HIP_TRY(hipMemcpy(fast, df, n, hipMemcpyDeviceToHost));
  (void)hipFree(da);
  (void)hipFree(dk);
  (void)hipFree(dout);
  (void)hipFree(df);
  return TSDF_HIP_OK;
}

// Test hook: v_cvt_pk_u8_f32 of in[i] into byte 1 of 0xAABBCCDD (how it rounds, that it saturates, that it keeps the
// other bytes).
static __global__ void k_selftest_cvt_pk_u8(const float *in, uint32_t *out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1u, 0xAABBCCDDu);
}

extern "C" int tsdf_hip_selftest_cvt_pk_u8(const float *in, size_t n, uint32_t *out) {
  if (!in || !out || !n) return TSDF_HIP_E_INVALID;
  float *d_in = nullptr;
  uint32_t *d_out = nullptr;
  TSDF_HIP_TRY(hipMalloc(&d_in, n * 4));
  TSDF_HIP_TRY(hipMalloc(&d_out, n * 4));
  TSDF_HIP_TRY(hipMemcpy(d_in, in, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest_cvt_pk_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_in, d_out, n);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return TSDF_HIP_OK;
}

// Test hooks: the device's RGB2LAB (k_lab_image, with the host-built curve) and LAB2RGB on caller-chosen inputs, so
// the tests can sweep all 2^24 pixel colours and millions of L, A, B means against the reference's conversions.
static __global__ void k_selftest_lab2rgb(const float *lab, uint32_t *rgb, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rgb[i] = lab_to_rgb(lab[3 * i], lab[3 * i + 1], lab[3 * i + 2]);
}

extern "C" int tsdf_hip_selftest_rgb2lab(const uint8_t *bgra, size_t n, float *lab4) {
  if (!bgra || !lab4 || !n || n > (size_t)1 << 30) return TSDF_HIP_E_INVALID;
  float lut[256];
  tsdf_lab_curve(lut);
  uint32_t *d_px = nullptr;
  float *d_lut = nullptr;
  float4 *d_lab = nullptr;
  TSDF_HIP_TRY(hipMalloc(&d_px, n * 4));
  TSDF_HIP_TRY(hipMalloc(&d_lut, sizeof lut));
  TSDF_HIP_TRY(hipMalloc(&d_lab, n * sizeof(float4)));
  TSDF_HIP_TRY(hipMemcpy(d_px, bgra, n * 4, hipMemcpyHostToDevice));
  TSDF_HIP_TRY(hipMemcpy(d_lut, lut, sizeof lut, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_lab_image, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_px, d_lut, d_lab, (int)n);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipMemcpy(lab4, d_lab, n * sizeof(float4), hipMemcpyDeviceToHost));
  (void)hipFree(d_px);
  (void)hipFree(d_lut);
  (void)hipFree(d_lab);
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_selftest_lab2rgb(const float *lab3, size_t n, uint32_t *rgb) {
  if (!lab3 || !rgb || !n) return TSDF_HIP_E_INVALID;
  float *d_lab = nullptr;
  uint32_t *d_rgb = nullptr;
  TSDF_HIP_TRY(hipMalloc(&d_lab, n * 12));
  TSDF_HIP_TRY(hipMalloc(&d_rgb, n * 4));
  TSDF_HIP_TRY(hipMemcpy(d_lab, lab3, n * 12, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest_lab2rgb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_lab, d_rgb, n);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipMemcpy(rgb, d_rgb, n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d_lab);
  (void)hipFree(d_rgb);
  return TSDF_HIP_OK;
}

// Test hook: what the hardware returns for structured buffer loads at (u, v) inside and outside a W x H float image
// (row index v, byte offset 4 u): the frame gather relies on out-of-range rows AND columns reading 0.
static __global__ void k_selftest_struct_oob(const float *img, int W, int H, unsigned soff, const int *uv, unsigned *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const i4_rsrc r = make_rsrc_2d(img, (unsigned)W * 4u, (unsigned)H);
  out[i] = tsdf_struct_buffer_load_u32(r, uv[2 * i + 1], uv[2 * i] * 4, (int)soff, 0);
}

extern "C" int tsdf_hip_selftest_struct_oob(const float *img, int W, int H, int planes, int plane, const int32_t *uv, uint32_t *out,
                                            int n) {
  if (!img || !uv || !out || n <= 0 || W <= 0 || H <= 0 || planes < 1 || plane < 0 || plane >= planes) return TSDF_HIP_E_INVALID;
  float *d_img = nullptr;
  int *d_uv = nullptr;
  unsigned *d_out = nullptr;
  const size_t npx = (size_t)W * H;
  TSDF_HIP_TRY(hipMalloc(&d_img, npx * 4 * planes));
  TSDF_HIP_TRY(hipMalloc(&d_uv, (size_t)n * 8));
  TSDF_HIP_TRY(hipMalloc(&d_out, (size_t)n * 4));
  TSDF_HIP_TRY(hipMemcpy(d_img, img, npx * 4 * planes, hipMemcpyHostToDevice));
  TSDF_HIP_TRY(hipMemcpy(d_uv, uv, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest_struct_oob, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_img, W, H,
                     (unsigned)(npx * 4 * plane), d_uv, d_out, n);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipMemcpy(out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d_img);
  (void)hipFree(d_uv);
  (void)hipFree(d_out);
  return TSDF_HIP_OK;
}

// Test hook: the kernel's pixel projection (certified fp32 path and exact fp64 path) on arbitrary
// camera-frame points g (n x 3), with this volume's intrinsics and image size.
static __global__ void k_selftest_project(const IntegrateArgs a, const double *cam, const float *g, size_t n, int *pix_fast,
                                          int *pix_exact, unsigned char *ambiguous) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gx = g[3 * i], gy = g[3 * i + 1], gz = g[3 * i + 2];
  bool amb = false;
  int pf = project_fast(a, gx, gy, gz, amb);
  const int pe = project_exact(a, cam, gx, gy, gz);
  if (amb) pf = pe;  // what integrate_quad does
  pix_fast[i] = pf;
  pix_exact[i] = pe;
  ambiguous[i] = amb ? 1 : 0;
}

extern "C" int tsdf_hip_selftest_project(tsdf_handle h, const float *g, size_t n, int32_t *pix_fast,
                                         int32_t *pix_exact, uint8_t *ambiguous) {
  if (!h || !g || !n || !pix_fast || !pix_exact || !ambiguous) return TSDF_HIP_E_INVALID;
  if (h->multi) h = tsdf_multi_first(h);  // the projection only reads the intrinsics, which every slab holds
  TSDF_ENTER(h);
  const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  const IntegrateHost a = make_args(h, ident);
  if (!fast_projection_ok(a, true, true)) {
    tsdf_set_error("fast projection disabled for this camera");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  float *dg = nullptr;
  int *dpf = nullptr, *dpe = nullptr;
  unsigned char *da = nullptr;
  TSDF_HIP_TRY(hipMalloc(&dg, n * 12));
  TSDF_HIP_TRY(hipMalloc(&dpf, n * 4));
  TSDF_HIP_TRY(hipMalloc(&dpe, n * 4));
  TSDF_HIP_TRY(hipMalloc(&da, n));
  TSDF_HIP_TRY(hipMemcpy(dg, g, n * 12, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest_project, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a.a, h->cam64, dg, n, dpf, dpe, da);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipMemcpy(pix_fast, dpf, n * 4, hipMemcpyDeviceToHost));
  TSDF_HIP_TRY(hipMemcpy(pix_exact, dpe, n * 4, hipMemcpyDeviceToHost));
  TSDF_HIP_TRY(hipMemcpy(ambiguous, da, n, hipMemcpyDeviceToHost));
  (void)hipFree(dg);
  (void)hipFree(dpf);
  (void)hipFree(dpe);
  (void)hipFree(da);
  return TSDF_HIP_OK;
}
#endif  // TSDF_HIP_TEST_HOOKS
