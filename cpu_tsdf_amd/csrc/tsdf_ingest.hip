// libtsdf_hip.so -- frame ingest: the `integrate` program's per-cloud preparation
// (src/prog/integrate.cpp:559-618) as two small kernels, so an unorganised sensor cloud goes from the
// caller's buffer to the integrate kernel's [depth | bgra] frame without a host-side z-buffer loop.
//
//   units      pt.xyz *= cloud_units                                                   (:559-568)
//   zero_nans  (0,0,0) -> NaN                                                          (:570-578)
//   world      pcl::transformPointCloud(cloud, cloud, poses[i].inverse())              (:580-581)
//              [PCL-recall: detail::Transformer<double>::se3, x*c0 + (y*c1 + (z*c2 + c3)) in double,
//              cast to float; points with a non-finite coordinate are left untouched because a cloud
//              read from a PCD file with NaNs is not dense]
//   organise   reprojectPoint (:201-207, ALL-float arithmetic: the program's intrinsics are floats) and
//              a z-buffer: a pixel keeps the point with the smallest z, the earliest one among equals
//              (`pt_old.z > pt.z` replaces, :597)                                      (:596-617)
//
// The serial loop's result does not depend on the visiting order except for that tie rule, so the
// z-buffer is a 64-bit atomicMin over (z bits << 32 | point index): z > 0 makes the float's bit pattern
// order-preserving, the index breaks ties toward the earlier point.  Latency-bound scatter; a frame is
// 0.3 M points, so no roofline claim.  Compiled with -ffp-contract=off like everything else.
#include <math.h>

#include "tsdf_common.h"

struct IngestArgs {
  const float *xyz;        // n points, `xyz_stride` floats apart
  const uint8_t *bgra;     // n colours, `bgra_stride` bytes apart (PCL b,g,r,a); may be null
  size_t xyz_stride, bgra_stride, n;
  float units;
  int zero_nans, has_tf;
  double tf[12];           // rows of poses[i].inverse().matrix()
  float fx, fy, cx, cy;    // focal_length_x_ ... principal_point_y_ (floats in the program)
  int W, H;
};

// One point through :559-581; false if the point is dropped by reprojectPoint.
static __device__ __forceinline__ bool ingest_point(const IngestArgs &a, size_t i, float p[3], int &pix) {
  const float *s = a.xyz + i * a.xyz_stride;
  float x = s[0], y = s[1], z = s[2];
  if (a.units != 1.f) {
    x *= a.units;
    y *= a.units;
    z *= a.units;
  }
  if (a.zero_nans && x == 0 && y == 0 && z == 0) x = y = z = NAN;
  if (a.has_tf && isfinite(x) && isfinite(y) && isfinite(z)) {
    const double dx = x, dy = y, dz = z;
    float o[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = (float)(dx * a.tf[4 * r] + (dy * a.tf[4 * r + 1] + (dz * a.tf[4 * r + 2] + a.tf[4 * r + 3])));
    x = o[0], y = o[1], z = o[2];
  }
  p[0] = x, p[1] = y, p[2] = z;
  // reprojectPoint (:201-207): float expression converted to int (cvttss2si: out of range -> INT_MIN)
  const float fu = (x * a.fx / z) + a.cx, fv = (y * a.fy / z) + a.cy;
  const int u = (fu > -2147483904.f && fu < 2147483648.f) ? (int)fu : INT_MIN;
  const int v = (fv > -2147483904.f && fv < 2147483648.f) ? (int)fv : INT_MIN;
  pix = v * a.W + u;
  return !isnan(z) && z > 0 && u >= 0 && u < a.W && v >= 0 && v < a.H;
}

static __global__ void __launch_bounds__(256)
k_ingest_scatter(const IngestArgs a, unsigned long long *__restrict__ zbuf) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  float p[3];
  int pix;
  if (!ingest_point(a, i, p, pix)) return;
  atomicMin(&zbuf[pix], ((unsigned long long)__float_as_uint(p[2]) << 32) | (unsigned long long)i);
}

static __global__ void __launch_bounds__(256)
k_ingest_resolve(const IngestArgs a, const unsigned long long *__restrict__ zbuf, float *__restrict__ depth,
                 uint32_t *__restrict__ bgra_out, unsigned long long *__restrict__ n_valid) {
  const size_t px = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = px < (size_t)a.W * a.H;
  bool valid = false;
  if (in) {
    const unsigned long long key = zbuf[px];
    valid = key != ~0ull;
    depth[px] = valid ? __uint_as_float((uint32_t)(key >> 32)) : NAN;  // every pixel starts at z = NaN (:594-595)
    if (bgra_out) {
      uint32_t c = 0xff000000u;  // PointXYZRGBA's default constructor: r = g = b = 0, a = 255
      if (valid && a.bgra) {
        const uint8_t *s = a.bgra + (size_t)(uint32_t)key * a.bgra_stride;
        c = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
      }
      bgra_out[px] = c;
    }
  }
  const unsigned long long m = __ballot(valid);
  if (m && (threadIdx.x & 63u) == 0) atomicAdd(n_valid, (unsigned long long)__popcll(m));
}

extern "C" int tsdf_hip_organize(tsdf_handle h, const float *xyz, size_t xyz_stride, const uint8_t *bgra,
                                 size_t bgra_stride, size_t n, float cloud_units, int zero_nans,
                                 const double world_to_cam[12], float *depth_out, uint8_t *bgra_out,
                                 uint64_t *n_valid) {
  if (!h || (n && !xyz) || xyz_stride < 3 || (bgra && bgra_stride < 4) || n >= (1ull << 32)) return TSDF_HIP_E_INVALID;
  if (h->multi)
    return tsdf_multi_organize(h, xyz, xyz_stride, bgra, bgra_stride, n, cloud_units, zero_nans, world_to_cam, depth_out, bgra_out,
                               n_valid);
  TSDF_ENTER(h);
  const tsdf_params &p = h->p;
  const size_t npx = (size_t)p.image_width * p.image_height;
  // scratch: zbuf[npx] u64 | points | colours
  const size_t b_z = npx * 8, b_xyz = ((n * xyz_stride * 4 + 15) / 16) * 16, b_c = bgra ? n * bgra_stride : 0;
  int rc = tsdf_ensure_scratch(h, b_z + b_xyz + b_c + 16);
  if (rc) return rc;
  char *sp = (char *)h->scratch;
  unsigned long long *zbuf = (unsigned long long *)sp;
  IngestArgs a;
  a.xyz = (const float *)(sp + b_z);
  a.bgra = bgra ? (const uint8_t *)(sp + b_z + b_xyz) : nullptr;
  a.xyz_stride = xyz_stride;
  a.bgra_stride = bgra_stride;
  a.n = n;
  a.units = cloud_units;
  a.zero_nans = zero_nans;
  a.has_tf = world_to_cam != nullptr;
  for (int i = 0; i < 12; ++i) a.tf[i] = world_to_cam ? world_to_cam[i] : 0.0;
  a.fx = (float)p.fx, a.fy = (float)p.fy, a.cx = (float)p.cx, a.cy = (float)p.cy;
  a.W = p.image_width, a.H = p.image_height;
  TSDF_HIP_TRY(hipMemsetAsync(zbuf, 0xff, b_z, h->stream));
  TSDF_HIP_TRY(hipMemsetAsync(h->counter, 0, sizeof(unsigned long long), h->stream));
  if (n) {
    if ((rc = tsdf_to_device(h, sp + b_z, xyz, n * xyz_stride * 4))) return rc;
    if (bgra && (rc = tsdf_to_device(h, sp + b_z + b_xyz, bgra, b_c))) return rc;
    hipLaunchKernelGGL(k_ingest_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, a, zbuf);
    TSDF_HIP_TRY(hipGetLastError());
  }
  hipLaunchKernelGGL(k_ingest_resolve, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, h->stream, a, zbuf,
                     h->frame_depth, h->frame_bgra, h->counter);
  TSDF_HIP_TRY(hipGetLastError());
  if (depth_out && (rc = tsdf_to_host(h, depth_out, h->frame_depth, npx * 4))) return rc;
  if (bgra_out && (rc = tsdf_to_host(h, bgra_out, h->frame_bgra, npx * 4))) return rc;
  unsigned long long nv = 0;
  if (n_valid) TSDF_HIP_TRY(hipMemcpyAsync(&nv, h->counter, sizeof nv, hipMemcpyDeviceToHost, h->stream));
  if (depth_out || bgra_out || n_valid) TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  if (n_valid) *n_valid = nv;
  h->frame_staged = 1;
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_integrate_staged(tsdf_handle h, const float cam_from_vol[12], uint64_t *n_observed) {
  if (!h || !cam_from_vol) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_integrate_staged(h, cam_from_vol, n_observed);
  if (!h->frame_staged) {
    tsdf_set_error("no organised frame is staged: call tsdf_hip_organize first");
    return TSDF_HIP_E_INVALID;
  }
  return tsdf_hip_integrate_device(h, h->frame_depth, h->p.integrate_color ? h->frame_bgra : nullptr, cam_from_vol,
                                   n_observed);
}
