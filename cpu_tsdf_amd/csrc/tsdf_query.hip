// libtsdf_hip.so -- renderView (one thread per ray) and getFxn/getGradient/getHessian (one thread per
// query point) on the flat SoA grid.
//
// Replaces TSDFVolumeOctree::renderView (src/lib/tsdf_volume_octree.cpp:278-421),
// getTSDFValue/interpolateTrilinearly (:453-541), getFxn/getGradient/getHessian/getNeighbors (:655-828).
// Every root-to-leaf pointer chase of the reference (Octree::getContainingVoxel, src/lib/octree.cpp:
// 112-133,628-643) becomes an index computation that replays the octree's descent comparisons exactly,
// so a ray visits the same voxels and takes the same steps as on the CPU.  Both kernels are
// gather/latency-bound (dependent loads along the ray); no roofline claim is made for them.
// Compiled with -ffp-contract=off; float/double operation order follows the reference line by line.
#include <limits.h>
#include <math.h>

#include <algorithm>
#include <vector>

#include <string.h>

#include "tsdf_common.h"
#ifdef TSDF_HIP_TEST_HOOKS
#include "tsdf_hip_test.h"
#endif

struct GridView {
  int nx, ny, nz;        // full resolution
  int z_first, nz_alloc; // allocated plane range
  int lv[3];             // octree levels per axis (log2 res) or -1
  float size[3];
  float nsize[3];        // size the axis' node centres descend from (tsdf_node_size: size_x on an octree grid)
  float half[3];         // size/2 in float (root bounds test, octree.cpp:630)
  int64_t pitch;
  const float *d;
  PlaneView pv;          // weights (and colour) through tsdf_load_w: layout-independent
  const float *ctr[3];   // octree node-centre tables
};

static GridView make_view(const tsdf_hip_volume *v) {
  GridView g;
  g.nx = v->nx;
  g.ny = v->ny;
  g.nz = v->nz;
  g.z_first = v->z_first;
  g.nz_alloc = v->nz_alloc;
  for (int a = 0; a < 3; ++a) {
    g.lv[a] = v->levels[a];
    g.size[a] = v->p.size[a];
    g.nsize[a] = tsdf_node_size(v->p, a);
    g.half[a] = v->p.size[a] / 2;
    g.ctr[a] = v->ctr[a];
  }
  g.pitch = v->pitch;
  g.d = v->d;
  g.pv = tsdf_plane_view(v);
  return g;
}

// x86 cvttsd2si semantics (see tsdf_integrate.hip)
static __device__ __forceinline__ int cvtt(double v) {
  return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : INT_MIN;
}

// One axis of OctreeNode::getContainingVoxel's descent (octree.cpp:112-121): child bit = (x - ctr) > 0.
static __device__ __forceinline__ int descend_axis(float x, float size, int L) {
  float c = 0.f, s = size;
  int i = 0;
  for (int l = 0; l < L; ++l) {
    const bool b = (x - c) > 0.f;
    const float off = s / 4;
    c = b ? c + off : c - off;
    s = s / 2;
    i = i * 2 + (b ? 1 : 0);
  }
  return i;
}

static __device__ __forceinline__ int axis_index(const GridView &g, int a, float x) {
  if (g.lv[a] >= 0) return descend_axis(x, g.nsize[a], g.lv[a]);
  const int res = a == 0 ? g.nx : a == 1 ? g.ny : g.nz;
  int i = cvtt(floor(((double)x + (double)g.size[a] / 2.0) / (double)g.size[a] * (double)res));
  return i < 0 ? 0 : (i >= res ? res - 1 : i);
}

// Octree::getContainingVoxel (octree.cpp:628-643) on a fully refined tree; false == NULL.
// `local` reports whether the voxel's plane is held by this handle (always true for a whole grid).
static __device__ __forceinline__ bool containing(const GridView &g, float x, float y, float z, int64_t &vi,
                                                  bool &local, int &k) {
  local = true;
  k = -1;
  if (isnan(z) || fabsf(x) > g.half[0] || fabsf(y) > g.half[1] || fabsf(z) > g.half[2]) return false;
  const int i = axis_index(g, 0, x), j = axis_index(g, 1, y);
  k = axis_index(g, 2, z);
  const int kl = k - g.z_first;
  local = kl >= 0 && kl < g.nz_alloc;
  vi = ((int64_t)(local ? kl : 0) * g.ny + j) * g.pitch + i;
  return true;
}

// getVoxelCenter (tsdf_volume_octree.cpp:553-560), one axis: double formula rounded to float.
static __device__ __forceinline__ float voxel_center(const GridView &g, int a, int i) {
  const int res = a == 0 ? g.nx : a == 1 ? g.ny : g.nz;
  const float off = g.size[a] / 2.0;
  return (float)(((size_t)i + 0.5) * g.size[a] / (double)res - off);
}

// getVoxelIndex (tsdf_volume_octree.cpp:562-574), one axis.
static __device__ __forceinline__ int voxel_index(const GridView &g, int a, float x) {
  const int res = a == 0 ? g.nx : a == 1 ? g.ny : g.nz;
  const double off = (double)g.size[a] / 2.0;
  return cvtt(floor(((double)x + off) / (double)g.size[a] * (double)res));
}

// interpolateTrilinearly (tsdf_volume_octree.cpp:486-541).  *valid is only ever AND-ed, as in the
// reference.  `local` is cleared if a corner plane is not held by this handle.
static __device__ float trilinear(const GridView &g, float x, float y, float z, bool &valid, bool &local) {
  int xi = voxel_index(g, 0, x), yi = voxel_index(g, 1, y), zi = voxel_index(g, 2, z);
  const bool exists = xi >= 0 && yi >= 0 && zi >= 0 && xi < g.nx && yi < g.ny && zi < g.nz;
  if (!exists || xi <= 0 || xi >= g.nx - 1 || yi <= 0 || yi >= g.ny - 1 || zi <= 0 || zi >= g.nz - 1) {
    valid = false;
    return NAN;
  }
  float vx = voxel_center(g, 0, xi), vy = voxel_center(g, 1, yi), vz = voxel_center(g, 2, zi);
  if (x < vx) xi -= 1;
  if (y < vy) yi -= 1;
  if (z < vz) zi -= 1;
  vx = voxel_center(g, 0, xi);
  vy = voxel_center(g, 1, yi);
  vz = voxel_center(g, 2, zi);
  const float a = (x - vx) * g.nx / g.size[0];
  const float b = (y - vy) * g.ny / g.size[1];
  const float c = (z - vz) * g.nz / g.size[2];
  const int kl = zi - g.z_first;
  if (kl < 0 || kl + 1 >= g.nz_alloc) {
    local = false;
    valid = false;
    return NAN;
  }
  const int64_t o = ((int64_t)kl * g.ny + yi) * g.pitch + xi;
  const int64_t sy = g.pitch, sz = (int64_t)g.ny * g.pitch;
  const int64_t ox = o + 1, oy = o + sy, oz = o + sz, oxy = o + sy + 1, oxz = o + sz + 1, oyz = o + sz + sy,
                oxyz = o + sz + sy + 1;
  valid = valid && (tsdf_load_w(g.pv, o) > 0);
  valid = valid && (tsdf_load_w(g.pv, ox) > 0);
  valid = valid && (tsdf_load_w(g.pv, oy) > 0);
  valid = valid && (tsdf_load_w(g.pv, oz) > 0);
  valid = valid && (tsdf_load_w(g.pv, oxy) > 0);
  valid = valid && (tsdf_load_w(g.pv, oxz) > 0);
  valid = valid && (tsdf_load_w(g.pv, oyz) > 0);
  valid = valid && (tsdf_load_w(g.pv, oxyz) > 0);
  return (g.d[o] * (1 - a) * (1 - b) * (1 - c) + g.d[oz] * (1 - a) * (1 - b) * (c) +
          g.d[oy] * (1 - a) * (b) * (1 - c) + g.d[oyz] * (1 - a) * (b) * (c) +
          g.d[ox] * (a) * (1 - b) * (1 - c) + g.d[oxz] * (a) * (1 - b) * (c) + g.d[oxy] * (a) * (b) * (1 - c) +
          g.d[oxyz] * (a) * (b) * (c));
}

// Eigen::Vector3f::normalize() [Eigen-recall 3.3]: z = x*x + (y*y + z*z); if (z > 0) v /= sqrt(z)
static __device__ __forceinline__ void normalize3(float v[3]) {
  const float n2 = v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]);
  if (n2 > 0.f) {
    const float n = sqrtf(n2);
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
  }
}

struct RayArgs {
  float rot[9], org[3];
  double nfx, nfy, ncx, ncy;
  int nw, nh;
  float zmin, zmax, neg;
  float min_step;     // max_dist_neg_ * 3/4.           (:289)
  float refine_step;  // (zsize_/zres_)/2.              (:329)
  float leaf;         // finest leaf size_ (size_x halved L times; getMinSize/getSize, octree.cpp:58-78)
  int to_camera;      // apply the final transformPointCloudWithNormals(trans^-1) (:422) in the kernel
  double inv[12];     // rows of trans.inverse().matrix(), computed by the caller's Eigen
};

// Ray hand-off between Z-slab handles (multi-GPU renderView).  A ray's step sequence depends on the last
// voxel it visited, so slabs cannot march it independently; instead the ray's loop state travels.  One
// record per ray, TSDF_HIP_RAY_RECORD_INTS 32-bit words:
//   [0] status: 0 untouched (only in a delta buffer), 1 suspended, 2 finished
//   [1] need_z: global plane of the voxel the main loop needs next (-1: none yet)
//   [2] niter  [3] hit_voxel  [4] t  [5..7] pt  [8] last_d  [9] last_w  [10] step  [11] pixel index
//   [12] finish flag: the main loop is done, word 4 holds t_star, only the normal is left  [13..15] zero
//   [16..23] the 8 output floats (valid when finished)
// k_raycast<true> resumes every suspended ray this handle is responsible for -- need_z inside its OWNED
// planes [z_begin, z_end), or ray index % world == rank while the ray has not needed a voxel yet -- and
// marches it until it finishes or its main loop reaches a voxel of another slab.  The refinement walk
// (:326-356) and the final trilinear samples read up to `halo` planes beyond the owned range; a read
// outside the allocated planes sets `incomplete` (the caller then asks for a larger halo).
struct RaySlab {
  int rank, world, z_begin, z_end;
  int list_count;  // > 0: state/delta are ONE compact list of that many records, ray id in word 11, updated in place
};
#define RAY_REC TSDF_HIP_RAY_RECORD_INTS

static __device__ __forceinline__ void ray_direction(const RayArgs &a, int64_t i, float du[3]) {
  const size_t x = (size_t)(i % a.nw), y = (size_t)(i / a.nw);
  du[0] = (float)((x - a.ncx) / a.nfx);
  du[1] = (float)((y - a.ncy) / a.nfy);
  du[2] = 1.f;
  normalize3(du);
  // du = R * du: each coefficient p0 + (p1 + p2) [Eigen-recall 3.3 reduction tree]
  const float p = du[0], q = du[1], r = du[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) du[k] = a.rot[3 * k] * p + (a.rot[3 * k + 1] * q + a.rot[3 * k + 2] * r);
}

// The state every ray starts from (:305-313).
static __global__ void __launch_bounds__(256) k_ray_begin(const RayArgs a, int *__restrict__ state) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.nw * a.nh) return;
  float du[3];
  ray_direction(a, i, du);
  int *r = state + RAY_REC * i;
  const float t = a.zmin;
  r[0] = 1;
  r[1] = -1;
  r[2] = 0;
  r[3] = 0;
  r[4] = __float_as_int(t);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float pt = a.org[k];
    pt += t * du[k];
    r[5 + k] = __float_as_int(pt);
  }
  r[8] = r[9] = 0;  // last_d, last_w = 0.f
  r[10] = __float_as_int(a.min_step);
  r[11] = (int)i;  // the ray's pixel index: lets records travel in compact lists (tsdf_hip_raycast_advance_list)
  for (int k = 12; k < RAY_REC; ++k) r[k] = 0;
}

// renderView, tsdf_volume_octree.cpp:290-421, one ray per thread.  The while loop runs until every lane
// of the wavefront has left it (the hardware's exec-mask loop is the ballot); a finished lane idles.
// out: 8 floats per pixel: x,y,z, nx,ny,nz, t (t_star on a hit), iterations; `incomplete` counts rays
// that touched a plane this handle does not hold (only possible for Z-slab handles).
// RESUMABLE: see RaySlab above; `out` is unused, results go to the ray's record in `delta`.
template <bool RESUMABLE>
static __global__ void __launch_bounds__(256)
k_raycast(const GridView g, const RayArgs a, float *__restrict__ out, unsigned *__restrict__ incomplete,
          const int *__restrict__ state, int *__restrict__ delta, const RaySlab rs) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // record (and, unless a list, pixel) index
  int64_t i = slot;
  if (RESUMABLE && rs.list_count > 0) {
    if (slot >= rs.list_count) return;
    i = state[RAY_REC * slot + 11];
  }
  if (i < 0 || i >= (int64_t)a.nw * a.nh) return;
  bool found_crossing = false, all_local = true, finish_only = false;
  float du[3];
  ray_direction(a, i, du);
  float pt[3] = {a.org[0], a.org[1], a.org[2]};
  float dd = 0, ww = 0, last_w = 0, last_d = 0;
  float t = a.zmin;
  float step = a.min_step;
  bool hit_voxel = false;
  int niter = 0;
  if (RESUMABLE) {
    const int *r = state + RAY_REC * slot;
    if (r[0] != 1) return;
    const int need_z = r[1];
    const bool mine = need_z < 0 ? (int)(i % rs.world) == rs.rank : (need_z >= rs.z_begin && need_z < rs.z_end);
    if (!mine) return;
    niter = r[2];
    hit_voxel = r[3] != 0;
    t = __int_as_float(r[4]);
#pragma unroll
    for (int k = 0; k < 3; ++k) pt[k] = __int_as_float(r[5 + k]);
    last_d = __int_as_float(r[8]);
    last_w = __int_as_float(r[9]);
    step = __int_as_float(r[10]);
    finish_only = r[12] != 0;  // the main loop is done; only the normal at t_star (= t) is left, see below
    found_crossing = finish_only;
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) pt[k] += t * du[k];
  }
  int suspend_z = -1;
  while (!finish_only && t < a.zmax) {
    int64_t vi;
    bool local;
    int kz;
    if (containing(g, pt[0], pt[1], pt[2], vi, local, kz)) {
      if (RESUMABLE && (kz < rs.z_begin || kz >= rs.z_end)) {  // another slab's voxel: hand the ray over
        suspend_z = kz;
        break;
      }
      all_local = all_local && local;
      hit_voxel = true;
      dd = local ? g.d[vi] : -1.f;
      ww = local ? tsdf_load_w(g.pv, vi) : 0.f;
      if (((dd < 0 && last_d > 0) || (dd > 0 && last_d < 0)) && last_w && ww) {
        found_crossing = true;
        const float old_t = t - step;
        step = a.refine_step;
        float last_new_d = dd, last_new_w = ww;
        while (t >= old_t) {
          t -= step;
#pragma unroll
          for (int k = 0; k < 3; ++k) pt[k] -= step * du[k];
          if (!containing(g, pt[0], pt[1], pt[2], vi, local, kz)) break;
          all_local = all_local && local;
          const float new_d = local ? g.d[vi] : -1.f, new_w = local ? tsdf_load_w(g.pv, vi) : 0.f;
          if ((last_d > 0 && new_d > 0) || (last_d < 0 && new_d < 0)) {
            last_d = new_d;
            last_w = new_w;
            dd = last_new_d;
            ww = last_new_w;
            t += step;
#pragma unroll
            for (int k = 0; k < 3; ++k) pt[k] += step * du[k];
            break;
          }
          last_new_d = dd;  // sic (:352-353)
          last_new_w = ww;
        }
        break;
      }
      last_d = dd;
      last_w = ww;
      const float s1 = a.leaf / 4.f, s2 = fabsf(dd) * a.neg;  // :360 (the double product rounds to this)
      step = s1 < s2 ? s2 : s1;
    } else if (hit_voxel) {
      break;
    }
    t += step;
#pragma unroll
    for (int k = 0; k < 3; ++k) pt[k] += step * du[k];
    niter++;
  }
  int *rec = RESUMABLE ? delta + RAY_REC * slot : nullptr;
  if (RESUMABLE) {
    rec[1] = suspend_z;
    rec[2] = niter;
    rec[3] = hit_voxel ? 1 : 0;
    rec[4] = __float_as_int(t);
#pragma unroll
    for (int k = 0; k < 3; ++k) rec[5 + k] = __float_as_int(pt[k]);
    rec[8] = __float_as_int(last_d);
    rec[9] = __float_as_int(last_w);
    rec[10] = __float_as_int(step);
    rec[12] = 0;
    rec[0] = suspend_z >= 0 ? 1 : 2;
    if (suspend_z >= 0) return;
  }
  float *o = RESUMABLE ? reinterpret_cast<float *>(rec + 16) : out + 8 * i;
  o[3] = o[4] = o[5] = 0.f;
  o[6] = t;
  o[7] = (float)niter;
  if (!found_crossing) {
    o[0] = o[1] = o[2] = NAN;
  } else {
    float t_star = t;  // finish_only: the record carries t_star in the t word
    if (!finish_only) {
      bool has_data = true;
      const float tcurr = t, tprev = t - step;
      last_d = trilinear(g, a.org[0] + tprev * du[0], a.org[1] + tprev * du[1], a.org[2] + tprev * du[2], has_data,
                         all_local);
      dd = trilinear(g, a.org[0] + tcurr * du[0], a.org[1] + tcurr * du[1], a.org[2] + tcurr * du[2], has_data,
                     all_local);
      // :389  evaluated in double: unqualified fabs(float) is double fabs(double) under <cmath>
      t_star = (float)((double)t + (double)step * (-1 + fabs((double)(last_d / (last_d - dd)))));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = a.org[k] + t_star * du[k];
    o[6] = t_star;
    int64_t vi;
    bool local;
    int kz;
    if (!containing(g, o[0], o[1], o[2], vi, local, kz)) {
      o[3] = o[4] = o[5] = NAN;
    } else if (RESUMABLE && !finish_only &&
               ((kz - 2 < g.z_first && g.z_first > 0) || (kz + 2 >= g.z_first + g.nz_alloc && g.z_first + g.nz_alloc < g.nz))) {
      // t_star extrapolates from two trilinear samples and can land ANY distance ahead when they are nearly equal,
      // so the six samples of the normal (planes kz-2 .. kz+2) may not be held here.  The ray then travels once
      // more: suspended for the owner of the hit point's plane with the finish flag and t_star in the t word; that
      // rank recomputes o = org + t_star * du (the same arithmetic) and the normal.
      rec[0] = 1;
      rec[1] = kz;
      rec[4] = __float_as_int(t_star);
      rec[12] = 1;
      if (!all_local) atomicAdd(incomplete, 1u);
      return;
    } else {
      const float s = a.leaf;
      bool valid = true;
      const float d_xm = trilinear(g, o[0] - s, o[1], o[2], valid, all_local);
      const float d_xp = trilinear(g, o[0] + s, o[1], o[2], valid, all_local);
      const float d_ym = trilinear(g, o[0], o[1] - s, o[2], valid, all_local);
      const float d_yp = trilinear(g, o[0], o[1] + s, o[2], valid, all_local);
      const float d_zm = trilinear(g, o[0], o[1], o[2] - s, valid, all_local);
      const float d_zp = trilinear(g, o[0], o[1], o[2] + s, valid, all_local);
      if (!valid) {
        o[3] = o[4] = o[5] = NAN;
      } else {
        float dF[3];
        dF[0] = (d_xp - d_xm) * a.neg / (2 * s);
        dF[1] = (d_yp - d_ym) * a.neg / (2 * s);
        dF[2] = (d_zp - d_zm) * a.neg / (2 * s);
        normalize3(dF);
        o[3] = dF[0];
        o[4] = dF[1];
        o[5] = dF[2];
      }
    }
  }
  // :422 pcl::transformPointCloudWithNormals(*cloud, *cloud, trans.inverse()) [PCL-recall: Transformer<double>,
  // se3 for the point, so3 for the normal, each x*c0 + (y*c1 + (z*c2 (+ c3))) in double, cast to float; the cloud
  // is not dense, so points with a non-finite coordinate are left untouched]
  if (!RESUMABLE && a.to_camera && isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2])) {
    const double px = o[0], py = o[1], pz = o[2], nx = o[3], ny = o[4], nz = o[5];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      o[r] = (float)(px * a.inv[4 * r] + (py * a.inv[4 * r + 1] + (pz * a.inv[4 * r + 2] + a.inv[4 * r + 3])));
      o[3 + r] = (float)(nx * a.inv[4 * r] + (ny * a.inv[4 * r + 1] + nz * a.inv[4 * r + 2]));
    }
  }
  if (!all_local) atomicAdd(incomplete, 1u);
}

static int make_ray_args(tsdf_handle h, const float rot[9], const float origin[3], int downsample, RayArgs &a) {
  const tsdf_params &p = h->p;
  a.to_camera = 0;
  for (int i = 0; i < 12; ++i) a.inv[i] = 0;
  for (int i = 0; i < 9; ++i) a.rot[i] = rot[i];
  for (int i = 0; i < 3; ++i) a.org[i] = origin[i];
  a.nw = p.image_width / downsample;
  a.nh = p.image_height / downsample;
  a.nfx = p.fx / downsample;
  a.nfy = p.fy / downsample;
  a.ncx = p.cx / downsample;
  a.ncy = p.cy / downsample;
  a.zmin = p.min_sensor_dist;
  a.zmax = p.max_sensor_dist;
  a.neg = p.max_dist_neg;
  a.min_step = p.max_dist_neg * 3 / 4.;
  a.refine_step = (p.size[2] / p.res[2]) / 2.;
  {
    float s = p.size[0];
    if (h->levels[0] >= 0)
      for (int l = 0; l < h->levels[0]; ++l) s = s / 2;
    else
      s = p.size[0] / p.res[0];
    a.leaf = s;
  }
  return (int64_t)a.nw * a.nh > 0 ? TSDF_HIP_OK : TSDF_HIP_E_INVALID;
}

static int raycast_impl(tsdf_handle h, const float rot[9], const float origin[3], int downsample, const double *inv,
                        float *out) {
  if (!h || !rot || !origin || !out || downsample < 1) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_raycast(h, rot, origin, downsample, inv, out);
  TSDF_ENTER(h);
  RayArgs a;
  if (make_ray_args(h, rot, origin, downsample, a)) return TSDF_HIP_E_INVALID;
  if (inv) {
    a.to_camera = 1;
    for (int i = 0; i < 12; ++i) a.inv[i] = inv[i];
  }
  const int64_t n = (int64_t)a.nw * a.nh;
  int rc = tsdf_ensure_scratch(h, (size_t)n * 8 * sizeof(float) + 16);
  if (rc) return rc;
  float *d_out = (float *)h->scratch;
  unsigned *d_inc = (unsigned *)((char *)h->scratch + (size_t)n * 8 * sizeof(float));
  TSDF_HIP_TRY(hipMemsetAsync(d_inc, 0, sizeof(unsigned), h->stream));
  const GridView g = make_view(h);
  hipLaunchKernelGGL(k_raycast<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, g, a, d_out, d_inc,
                     (const int *)nullptr, (int *)nullptr, RaySlab{0, 1, 0, 0, 0});
  TSDF_HIP_TRY(hipGetLastError());
  unsigned inc = 0;
  TSDF_HIP_TRY(hipMemcpyAsync(&inc, d_inc, sizeof inc, hipMemcpyDeviceToHost, h->stream));
  if ((rc = tsdf_to_host(h, out, d_out, (size_t)n * 8 * sizeof(float)))) return rc;  // (synchronises the stream)
  if (inc) {
    tsdf_set_error("raycast touched planes outside this handle's Z-slab (+halo)");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_raycast(tsdf_handle h, const float rot[9], const float origin[3], int downsample,
                                float *out) {
  return raycast_impl(h, rot, origin, downsample, nullptr, out);
}

extern "C" int tsdf_hip_raycast_camera(tsdf_handle h, const float rot[9], const float origin[3], int downsample,
                                       const double cam_from_vol[12], float *out) {
  if (!cam_from_vol) return TSDF_HIP_E_INVALID;
  return raycast_impl(h, rot, origin, downsample, cam_from_vol, out);
}

// Multi-slab renderView: ray records on the DEVICE (see RaySlab).  begin fills the start state of every
// ray; advance zero-fills `d_delta`, resumes the rays this handle is responsible for and writes their new
// records there.  The caller sums the deltas of all ranks (integer all-reduce: every ray is advanced by
// exactly one rank) and overwrites the touched records.
extern "C" int tsdf_hip_raycast_begin(tsdf_handle h, const float rot[9], const float origin[3], int downsample,
                                      int32_t *d_state) {
  if (!h || !rot || !origin || !d_state || downsample < 1) return TSDF_HIP_E_INVALID;
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_raycast_begin");
  TSDF_ENTER(h);
  RayArgs a;
  if (make_ray_args(h, rot, origin, downsample, a)) return TSDF_HIP_E_INVALID;
  const int64_t n = (int64_t)a.nw * a.nh;
  hipLaunchKernelGGL(k_ray_begin, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, a, (int *)d_state);
  TSDF_HIP_TRY(hipGetLastError());
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_raycast_advance(tsdf_handle h, const float rot[9], const float origin[3], int downsample,
                                        int rank, int world, const int32_t *d_state, int32_t *d_delta) {
  if (!h || !rot || !origin || !d_state || !d_delta || downsample < 1 || world < 1 || rank < 0 || rank >= world)
    return TSDF_HIP_E_INVALID;
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_raycast_advance");
  TSDF_ENTER(h);
  RayArgs a;
  if (make_ray_args(h, rot, origin, downsample, a)) return TSDF_HIP_E_INVALID;
  const int64_t n = (int64_t)a.nw * a.nh;
  int rc = tsdf_ensure_scratch(h, 16);
  if (rc) return rc;
  unsigned *d_inc = (unsigned *)h->scratch;
  TSDF_HIP_TRY(hipMemsetAsync(d_inc, 0, sizeof(unsigned), h->stream));
  TSDF_HIP_TRY(hipMemsetAsync(d_delta, 0, (size_t)n * RAY_REC * sizeof(int32_t), h->stream));
  const GridView g = make_view(h);
  hipLaunchKernelGGL(k_raycast<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, g, a,
                     (float *)nullptr, d_inc, (const int *)d_state, (int *)d_delta,
                     RaySlab{rank, world, h->z_begin, h->z_end, 0});
  TSDF_HIP_TRY(hipGetLastError());
  unsigned inc = 0;
  TSDF_HIP_TRY(hipMemcpyAsync(&inc, d_inc, sizeof inc, hipMemcpyDeviceToHost, h->stream));
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  if (inc) {
    tsdf_set_error("ray hand-off: the refinement walk / trilinear samples left this handle's halo planes; "
                   "create the slab with a larger halo (tsdf_hip_render_halo)");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  return TSDF_HIP_OK;
}

// The same on a COMPACT list of `count` records (ray id in word 11), updated in place: the scalable form of the
// hand-off, where a record travels point-to-point to the rank that owns its next voxel instead of every rank
// holding every ray.  Records this handle is not responsible for are left untouched.
extern "C" int tsdf_hip_raycast_advance_list(tsdf_handle h, const float rot[9], const float origin[3], int downsample,
                                             int rank, int world, int32_t *d_records, size_t count) {
  if (!h || !rot || !origin || downsample < 1 || world < 1 || rank < 0 || rank >= world || count >= (1ull << 31) ||
      (count && !d_records))
    return TSDF_HIP_E_INVALID;
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_raycast_advance_list");
  if (!count) return TSDF_HIP_OK;
  TSDF_ENTER(h);
  RayArgs a;
  if (make_ray_args(h, rot, origin, downsample, a)) return TSDF_HIP_E_INVALID;
  int rc = tsdf_ensure_scratch(h, 16);
  if (rc) return rc;
  unsigned *d_inc = (unsigned *)h->scratch;
  TSDF_HIP_TRY(hipMemsetAsync(d_inc, 0, sizeof(unsigned), h->stream));
  hipLaunchKernelGGL(k_raycast<true>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, make_view(h), a,
                     (float *)nullptr, d_inc, (const int *)d_records, (int *)d_records,
                     RaySlab{rank, world, h->z_begin, h->z_end, (int)count});
  TSDF_HIP_TRY(hipGetLastError());
  unsigned inc = 0;
  TSDF_HIP_TRY(hipMemcpyAsync(&inc, d_inc, sizeof inc, hipMemcpyDeviceToHost, h->stream));
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  if (inc) {
    tsdf_set_error("ray hand-off: the refinement walk / trilinear samples left this handle's halo planes; "
                   "create the slab with a larger halo (tsdf_hip_render_halo)");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  return TSDF_HIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Compact ray lists for the one-process multi-GPU renderView (tsdf_multi.hip).  Every slab keeps ONLY the records it is
// responsible for; after a round of k_raycast<true> in list mode a record is either finished -- its pixel index and 8
// output floats go to the image on the first slab -- or suspended for the owner of its next plane, and travels there
// point-to-point.  The kernels below are the device side of that routing: nothing image-sized ever crosses a link.
#define RAY_FIN_INTS TSDF_RAY_FIN_INTS
struct RayRoute {
  int n_slab;
  int z_end[TSDF_MAX_SLABS];  // owner of plane z = first slab whose z_end exceeds it
};

// The start records of the rays with pixel index = rank (mod world), as a compact list (:305-313).
static __global__ void __launch_bounds__(256)
k_ray_begin_list(const RayArgs a, int *__restrict__ list, int rank, int world, unsigned count) {
  const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= count) return;
  const int64_t i = (int64_t)slot * world + rank;
  float du[3];
  ray_direction(a, i, du);
  int *r = list + RAY_REC * (size_t)slot;
  const float t = a.zmin;
  r[0] = 1;
  r[1] = -1;
  r[2] = 0;
  r[3] = 0;
  r[4] = __float_as_int(t);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float pt = a.org[k];
    pt += t * du[k];
    r[5 + k] = __float_as_int(pt);
  }
  r[8] = r[9] = 0;
  r[10] = __float_as_int(a.min_step);
  r[11] = (int)i;
  for (int k = 12; k < RAY_REC; ++k) r[k] = 0;
}

static __device__ __forceinline__ int ray_dest(const RayRoute &rt, const int *r) {
  if (r[0] == 2) return rt.n_slab;  // finished: to the image
  int d = 0;
  while (d < rt.n_slab - 1 && r[1] >= rt.z_end[d]) ++d;
  return d;
}

// counters[0 .. n_slab] = records per destination (n_slab = finished)
static __global__ void __launch_bounds__(256)
k_ray_route_count(const int *__restrict__ list, unsigned count, const RayRoute rt, unsigned *__restrict__ counters) {
  __shared__ unsigned hist[TSDF_MAX_SLABS + 1];
  for (int k = threadIdx.x; k <= rt.n_slab; k += 256) hist[k] = 0;
  __syncthreads();
  const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot < count) atomicAdd(&hist[ray_dest(rt, list + RAY_REC * (size_t)slot)], 1u);
  __syncthreads();
  for (int k = threadIdx.x; k <= rt.n_slab; k += 256)
    if (hist[k]) atomicAdd(&counters[k], hist[k]);
}

// cursors[d] = first outbox slot of destination d (exclusive prefix over the slabs); the finished rays have their own box
static __global__ void k_ray_route_offsets(const unsigned *__restrict__ counters, unsigned *__restrict__ cursors, int n_slab) {
  if (threadIdx.x || blockIdx.x) return;
  unsigned run = 0;
  for (int k = 0; k < n_slab; ++k) {
    cursors[k] = run;
    run += counters[k];
  }
  cursors[n_slab] = 0;
}

// Records sorted by destination into `outbox` (suspended, whole records) and `finbox` (finished, RAY_FIN_INTS words);
// one atomic per wave and destination.
static __global__ void __launch_bounds__(256)
k_ray_route_scatter(const int *__restrict__ list, unsigned count, const RayRoute rt, unsigned *__restrict__ cursors,
                    int *__restrict__ outbox, int *__restrict__ finbox) {
  const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = slot < count;
  const int *r = list + RAY_REC * (size_t)(valid ? slot : 0);
  const int dest = valid ? ray_dest(rt, r) : -1;
  const unsigned lane = threadIdx.x & 63u;
  unsigned pos = 0;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int d = __shfl(dest, leader);
    const unsigned long long same = __ballot(valid && dest == d);
    unsigned base = 0;
    if ((int)lane == leader) base = atomicAdd(&cursors[d], (unsigned)__popcll(same));
    base = __shfl(base, leader);
    if (valid && dest == d) pos = base + (unsigned)__popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  if (!valid) return;
  if (dest == rt.n_slab) {
    int *o = finbox + RAY_FIN_INTS * (size_t)pos;
    o[0] = r[11];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[1 + k] = r[16 + k];
  } else {
    int *o = outbox + RAY_REC * (size_t)pos;
#pragma unroll
    for (int k = 0; k < RAY_REC; ++k) o[k] = r[k];
  }
}

// Finished rays into the image (8 floats per pixel), with renderView's last line (:422) applied as k_raycast does.
struct RayDeliver {
  int to_camera;
  double inv[12];
};
static __global__ void __launch_bounds__(256)
k_ray_deliver(const int *__restrict__ fin, unsigned count, float *__restrict__ out, int64_t n_pix, const RayDeliver dl) {
  const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= count) return;
  const int *f = fin + RAY_FIN_INTS * (size_t)slot;
  const int64_t pix = f[0];
  if (pix < 0 || pix >= n_pix) return;
  float o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = __int_as_float(f[1 + k]);
  if (dl.to_camera && isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2])) {
    const double px = o[0], py = o[1], pz = o[2], nx = o[3], ny = o[4], nz = o[5];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      o[r] = (float)(px * dl.inv[4 * r] + (py * dl.inv[4 * r + 1] + (pz * dl.inv[4 * r + 2] + dl.inv[4 * r + 3])));
      o[3 + r] = (float)(nx * dl.inv[4 * r] + (ny * dl.inv[4 * r + 1] + nz * dl.inv[4 * r + 2]));
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) out[8 * pix + k] = o[k];
}

// Host side of the above, all asynchronous on the slab's stream (the caller is on the slab's device).
unsigned tsdf_ray_list_share(int64_t n_rays, int rank, int world) {  // rays with index = rank (mod world)
  return n_rays > rank ? (unsigned)((n_rays - rank + world - 1) / world) : 0u;
}

int tsdf_ray_list_begin(tsdf_handle s, const float rot[9], const float origin[3], int downsample, int rank, int world,
                        int32_t *d_list, unsigned *count) {
  RayArgs a;
  if (make_ray_args(s, rot, origin, downsample, a)) return TSDF_HIP_E_INVALID;
  *count = tsdf_ray_list_share((int64_t)a.nw * a.nh, rank, world);
  if (!*count) return TSDF_HIP_OK;
  hipLaunchKernelGGL(k_ray_begin_list, dim3((*count + 255u) / 256u), dim3(256), 0, s->stream, a, (int *)d_list, rank, world, *count);
  TSDF_HIP_TRY(hipGetLastError());
  return TSDF_HIP_OK;
}

int tsdf_ray_list_advance(tsdf_handle s, const float rot[9], const float origin[3], int downsample, int rank, int world,
                          int32_t *d_list, unsigned count, unsigned *d_incomplete) {
  if (!count) return TSDF_HIP_OK;
  RayArgs a;
  if (make_ray_args(s, rot, origin, downsample, a)) return TSDF_HIP_E_INVALID;
  hipLaunchKernelGGL(k_raycast<true>, dim3((count + 255u) / 256u), dim3(256), 0, s->stream, make_view(s), a, (float *)nullptr,
                     d_incomplete, (const int *)d_list, (int *)d_list, RaySlab{rank, world, s->z_begin, s->z_end, (int)count});
  TSDF_HIP_TRY(hipGetLastError());
  return TSDF_HIP_OK;
}

// d_counters: 2 * (n_slab + 1) words -- counts per destination, then the scatter cursors.
int tsdf_ray_list_route(tsdf_handle s, const int32_t *d_list, unsigned count, int n_slab, const int *z_end,
                        unsigned *d_counters, int32_t *d_outbox, int32_t *d_finbox) {
  if (n_slab < 1 || n_slab > TSDF_MAX_SLABS) return TSDF_HIP_E_INVALID;
  TSDF_HIP_TRY(hipMemsetAsync(d_counters, 0, 2 * (size_t)(n_slab + 1) * sizeof(unsigned), s->stream));
  if (!count) return TSDF_HIP_OK;
  RayRoute rt;
  rt.n_slab = n_slab;
  for (int k = 0; k < TSDF_MAX_SLABS; ++k) rt.z_end[k] = k < n_slab ? z_end[k] : INT_MAX;
  const dim3 grid((count + 255u) / 256u), block(256);
  hipLaunchKernelGGL(k_ray_route_count, grid, block, 0, s->stream, (const int *)d_list, count, rt, d_counters);
  hipLaunchKernelGGL(k_ray_route_offsets, dim3(1), dim3(64), 0, s->stream, (const unsigned *)d_counters, d_counters + n_slab + 1, n_slab);
  hipLaunchKernelGGL(k_ray_route_scatter, grid, block, 0, s->stream, (const int *)d_list, count, rt, d_counters + n_slab + 1,
                     (int *)d_outbox, (int *)d_finbox);
  TSDF_HIP_TRY(hipGetLastError());
  return TSDF_HIP_OK;
}

int tsdf_ray_deliver(tsdf_handle s, const int32_t *d_fin, unsigned count, float *d_out, int64_t n_pix, const double *inv) {
  if (!count) return TSDF_HIP_OK;
  RayDeliver dl;
  dl.to_camera = inv ? 1 : 0;
  for (int i = 0; i < 12; ++i) dl.inv[i] = inv ? inv[i] : 0.0;
  hipLaunchKernelGGL(k_ray_deliver, dim3((count + 255u) / 256u), dim3(256), 0, s->stream, (const int *)d_fin, count, d_out, n_pix, dl);
  TSDF_HIP_TRY(hipGetLastError());
  return TSDF_HIP_OK;
}

// Planes of halo a Z-slab handle needs on each side for tsdf_hip_raycast_advance: the refinement walk goes
// back at most one main-loop step plus one refinement step, and the trilinear / central-difference samples reach two
// more voxels.  A main-loop step is max(leaf/4, |d| * max_dist_neg) (.cpp:360) and d lies in [-1, max_dist_pos /
// max_dist_neg], so the longest one is max(max_dist_neg, max_dist_pos) -- the hinge value exceeds 1 when the
// truncation is asymmetric (found by tests/evidence/fuzz_product_vs_oracle.py; before, the halo assumed |d| <= 1).
extern "C" int tsdf_hip_render_halo(const tsdf_params *p) {
  if (!p || p->res[2] <= 0 || !(p->size[2] > 0)) return -1;
  const double vs = (double)p->size[2] / p->res[2];
  const double leaf = (double)p->size[0] / p->res[0];
  const double step = std::max(leaf / 4, (double)std::max(p->max_dist_neg, p->max_dist_pos));
  return (int)ceil(step / vs) + 4;
}

// renderColoredView's per-hit lookup (tsdf_volume_octree.cpp:443-448): octree_->getContainingVoxel(v) then
// voxel->getRGB, batched.  found[i] = 0 where the reference gets NULL (or the plane is not held by this handle).
static __global__ void __launch_bounds__(256)
k_lookup_rgb(const GridView g, const int own_lo, const int own_hi, const float *__restrict__ xyz, size_t n,
             unsigned char *__restrict__ rgb, unsigned char *__restrict__ found) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  int64_t vi;
  bool local;
  int k;
  // (a Z-slab handle answers only for voxels in planes it OWNS: halo planes may be stale, and exactly one handle of a
  // partition owns any plane)
  const bool hit = containing(g, xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2], vi, local, k) && local && k >= own_lo && k < own_hi;
  uint32_t c = 0u;
  if (hit && g.pv.rgb) c = tsdf_load_rgb(g.pv, vi);
  rgb[3 * t] = (unsigned char)(c & 255u);
  rgb[3 * t + 1] = (unsigned char)((c >> 8) & 255u);
  rgb[3 * t + 2] = (unsigned char)((c >> 16) & 255u);
  found[t] = hit ? 1 : 0;
}

// the same, but the voxel's element index instead of its cached colour (-1 where not found): TSDF_COLOR_LAB volumes
// finish the colour on the host (tsdf_lab_exact_colors)
static __global__ void __launch_bounds__(256)
k_lookup_index(const GridView g, const int own_lo, const int own_hi, const float *__restrict__ xyz, size_t n,
               int64_t *__restrict__ idx) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  int64_t vi;
  bool local;
  int k;
  const bool hit = containing(g, xyz[3 * t], xyz[3 * t + 1], xyz[3 * t + 2], vi, local, k) && local && k >= own_lo && k < own_hi;
  idx[t] = hit ? vi : -1;
}

extern "C" int tsdf_hip_lookup_rgb(tsdf_handle h, const float *xyz, size_t n, uint8_t *rgb, uint8_t *found) {
  if (!h || !xyz || !n || !rgb || !found) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_lookup_rgb(h, xyz, n, rgb, found);
  TSDF_ENTER(h);
  if (h->lab_img) {  // LABNode::getRGB: exact bytes through the host's pow
    int rc = tsdf_ensure_scratch(h, n * 12);
    if (rc) return rc;
    float *d_xyz = (float *)h->scratch;
    if ((rc = tsdf_to_device(h, d_xyz, xyz, n * 12))) return rc;
    int64_t *d_idx = nullptr;
    TSDF_HIP_TRY(hipMalloc(&d_idx, n * sizeof(int64_t)));
    hipLaunchKernelGGL(k_lookup_index, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, make_view(h), h->z_begin, h->z_end,
                       d_xyz, n, d_idx);
    std::vector<int64_t> idx(n);
    std::vector<uint32_t> words(n);
    rc = tsdf_to_host(h, idx.data(), d_idx, n * sizeof(int64_t));
    if (!rc) rc = tsdf_lab_exact_colors(h, d_idx, n, words.data(), false);
    (void)hipFree(d_idx);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) {
      const bool hit = idx[i] >= 0;
      found[i] = hit ? 1 : 0;
      rgb[3 * i] = hit ? (uint8_t)(words[i] & 255u) : 0;
      rgb[3 * i + 1] = hit ? (uint8_t)((words[i] >> 8) & 255u) : 0;
      rgb[3 * i + 2] = hit ? (uint8_t)((words[i] >> 16) & 255u) : 0;
    }
    return TSDF_HIP_OK;
  }
  int rc = tsdf_ensure_scratch(h, n * 16);
  if (rc) return rc;
  float *d_xyz = (float *)h->scratch;
  unsigned char *d_rgb = (unsigned char *)(d_xyz + 3 * n), *d_found = d_rgb + 3 * n;
  if ((rc = tsdf_to_device(h, d_xyz, xyz, n * 12))) return rc;
  hipLaunchKernelGGL(k_lookup_rgb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, make_view(h), h->z_begin, h->z_end,
                     d_xyz, n, d_rgb, d_found);
  TSDF_HIP_TRY(hipGetLastError());
  if ((rc = tsdf_to_host(h, rgb, d_rgb, n * 3))) return rc;
  return tsdf_to_host(h, found, d_found, n);
}

// Test hook: Octree::getContainingVoxel's voxel index for arbitrary points (idx = i, j, k or -1, -1, -1 for NULL).
// (Measured alternatives to the level-by-level walk -- a boundary-table search and, on dyadic grids, computed
// boundaries -- were slower inside k_raycast: 1.78 and 2.40 ms vs 1.55 ms per 640x480 view at 2048^3; the
// fixed-trip, branch-free walk wins over shorter but divergent searches.)
#ifdef TSDF_HIP_TEST_HOOKS
static __global__ void __launch_bounds__(256)
k_selftest_containing(const GridView g, const float *__restrict__ xyz, size_t n, int *__restrict__ idx) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float x = xyz[3 * t], y = xyz[3 * t + 1], z = xyz[3 * t + 2];
  int64_t vi;
  bool local;
  int k;
  if (containing(g, x, y, z, vi, local, k)) {
    idx[3 * t] = axis_index(g, 0, x);
    idx[3 * t + 1] = axis_index(g, 1, y);
    idx[3 * t + 2] = k;
  } else {
    idx[3 * t] = idx[3 * t + 1] = idx[3 * t + 2] = -1;
  }
}

extern "C" int tsdf_hip_selftest_containing(tsdf_handle h, const float *xyz, size_t n, int32_t *idx) {
  if (!h || !xyz || !n || !idx) return TSDF_HIP_E_INVALID;
  if (h->multi) h = tsdf_multi_first(h);  // the descent only reads the centre tables, which every slab holds
  TSDF_ENTER(h);
  int rc = tsdf_ensure_scratch(h, n * 24);
  if (rc) return rc;
  float *d_xyz = (float *)h->scratch;
  int *d_idx = (int *)(d_xyz + 3 * n);
  if ((rc = tsdf_to_device(h, d_xyz, xyz, n * 12))) return rc;
  hipLaunchKernelGGL(k_selftest_containing, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, make_view(h), d_xyz,
                     n, d_idx);
  TSDF_HIP_TRY(hipGetLastError());
  return tsdf_to_host(h, idx, d_idx, n * 12);
}
#endif  // TSDF_HIP_TEST_HOOKS

// ---------------------------------------------------------------------------------------------
// getNeighbors :796-828, getFxn :655-672, getGradient :681-700, getHessian :703-726.
// Neighbour order: dx outer, dy, dz inner.  getFxn/getGradient read the octree NODE centre
// (vox->getCenter), getHessian reads getVoxelCenter (`centers[i]`).  Unqualified fabs(float) is
// double fabs(double), so every term is a double product accumulated into a float.
static __device__ __forceinline__ int sgn(float x) { return x > 0 ? 1 : -1; }  // :674-678

static __global__ void __launch_bounds__(256)
k_sample(const GridView g, const int own_lo, const int own_hi, const float *__restrict__ xyz, size_t n,
         float *__restrict__ val, float *__restrict__ grad, float *__restrict__ hess, unsigned char *__restrict__ ok) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const float px = xyz[3 * idx], py = xyz[3 * idx + 1], pz = xyz[3 * idx + 2];
  bool good = true;
  int xi = voxel_index(g, 0, px), yi = voxel_index(g, 1, py), zi = voxel_index(g, 2, pz);
  if (!(xi >= 0 && yi >= 0 && zi >= 0 && xi < g.nx && yi < g.ny && zi < g.nz)) good = false;
  if (good) {
    if (px < voxel_center(g, 0, xi)) xi -= 1;
    if (py < voxel_center(g, 1, yi)) yi -= 1;
    if (pz < voxel_center(g, 2, zi)) zi -= 1;
    if (xi < 0 || xi >= g.nx - 1 || yi < 0 || yi >= g.ny - 1 || zi < 0 || zi >= g.nz - 1) good = false;
  }
  const int kl = zi - g.z_first;
  // A Z-slab handle answers only for points whose lower-corner plane it OWNS: halo planes are allocated but only
  // as fresh as the caller's last exchange, and exactly one handle of a partition owns any plane (plane zi + 1
  // may be the first halo plane: the one-plane exchange marching cubes needs as well).
  if (good && (zi < own_lo || zi >= own_hi || kl < 0 || kl + 1 >= g.nz_alloc)) good = false;
  float v = NAN, gr[3] = {NAN, NAN, NAN}, h01 = NAN, h02 = NAN, h12 = NAN;
  if (good) {
    const float c = g.size[0] / g.nx;
    v = 0;
    gr[0] = gr[1] = gr[2] = 0;
    h01 = h02 = h12 = 0;
    for (int dx = 0; dx <= 1; dx++)
      for (int dy = 0; dy <= 1; dy++)
        for (int dz = 0; dz <= 1; dz++) {
          const int i = xi + dx, j = yi + dy, k = zi + dz;
          const float dv = g.d[((int64_t)(k - g.z_first) * g.ny + j) * g.pitch + i];
          const float nc[3] = {g.ctr[0][i], g.ctr[1][j], g.ctr[2][k]};
          const float fc[3] = {voxel_center(g, 0, i), voxel_center(g, 1, j), voxel_center(g, 2, k)};
          v += (c - fabs((double)(px - nc[0]))) * (c - fabs((double)(py - nc[1]))) *
               (c - fabs((double)(pz - nc[2]))) * dv;
          gr[0] += -sgn(px - nc[0]) * (c - fabs((double)(py - nc[1]))) * (c - fabs((double)(pz - nc[2]))) * dv;
          gr[1] += (c - fabs((double)(px - nc[0]))) * -sgn(py - nc[1]) * (c - fabs((double)(pz - nc[2]))) * dv;
          gr[2] += (c - fabs((double)(px - nc[0]))) * (c - fabs((double)(py - nc[1]))) * -sgn(pz - nc[2]) * dv;
          h01 += sgn(px - fc[0]) * sgn(py - fc[1]) * (c - fabs((double)(pz - fc[2]))) * dv;
          h02 += sgn(px - fc[0]) * (c - fabs((double)(py - fc[1]))) * sgn(pz - fc[2]) * dv;
          h12 += (c - fabs((double)(px - fc[0]))) * sgn(py - fc[1]) * sgn(pz - fc[2]) * dv;
        }
    const float c3 = c * c * c;
    v /= c3;
    gr[0] /= c3;
    gr[1] /= c3;
    gr[2] /= c3;
    h01 /= c3;
    h02 /= c3;
    h12 /= c3;
  }
  if (ok) ok[idx] = good ? 1 : 0;
  if (val) val[idx] = v;
  if (grad) {
    grad[3 * idx] = gr[0];
    grad[3 * idx + 1] = gr[1];
    grad[3 * idx + 2] = gr[2];
  }
  if (hess) {
    float *hp = hess + 9 * idx;
    const float z = good ? 0.f : NAN;
    hp[0] = hp[4] = hp[8] = z;
    hp[1] = hp[3] = h01;
    hp[2] = hp[6] = h02;
    hp[5] = hp[7] = h12;
  }
}

extern "C" int tsdf_hip_sample(tsdf_handle h, const float *xyz, size_t n, float *val, float *grad, float *hess,
                               uint8_t *ok) {
  if (!h || !xyz || !n) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_sample(h, xyz, n, val, grad, hess, ok);
  TSDF_ENTER(h);
  // scratch layout: xyz[3n] val[n] grad[3n] hess[9n] floats, ok[n] bytes
  const size_t fl = 16 * n;
  int rc = tsdf_ensure_scratch(h, fl * sizeof(float) + n + 16);
  if (rc) return rc;
  float *d_xyz = (float *)h->scratch, *d_val = d_xyz + 3 * n, *d_grad = d_val + n, *d_hess = d_grad + 3 * n;
  unsigned char *d_ok = (unsigned char *)(d_hess + 9 * n);
  if ((rc = tsdf_to_device(h, d_xyz, xyz, 3 * n * sizeof(float)))) return rc;
  const GridView g = make_view(h);
  hipLaunchKernelGGL(k_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, g, h->z_begin, h->z_end, d_xyz,
                     n, d_val, d_grad, d_hess, d_ok);
  TSDF_HIP_TRY(hipGetLastError());
  if (val && (rc = tsdf_to_host(h, val, d_val, n * sizeof(float)))) return rc;
  if (grad && (rc = tsdf_to_host(h, grad, d_grad, 3 * n * sizeof(float)))) return rc;
  if (hess && (rc = tsdf_to_host(h, hess, d_hess, 9 * n * sizeof(float)))) return rc;
  if (ok && (rc = tsdf_to_host(h, ok, d_ok, n))) return rc;
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  return TSDF_HIP_OK;
}
