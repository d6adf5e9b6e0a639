// libtsdf_hip.so -- marching cubes on the flat SoA grid.
//
// Replaces MarchingCubesTSDFOctree::reconstruct (src/lib/marching_cubes_tsdf_octree.cpp:108-236) and the
// PCL pieces it calls (pcl::MarchingCubes::createSurface / interpolateEdge, Bourke's tables).
//   k_mc_classify  streaming pass, one thread per quad of 4 x-consecutive cells: candidate test (:192-202),
//                  8 corner values (getValidNeighborList1D :145-177 / getGridValue :91-106), case index,
//                  triangle count; active cells are compacted through wave-private LDS lists into
//                  (Morton key, packed cell) pairs
//   rocprim sort   by key: the reference emits triangles in octree pre-order with child index
//                  4*(x>cx) + 2*(y>cy) + (z>cz) (octree.cpp:119,257-264) = Morton order, x the high bit
//   rocprim scan   triangle offsets
//   k_mc_emit      one thread per active cell: edge interpolation + triangle/colour output
// Case tables live in LDS.  Streaming stencil read of d and w: HBM-bound, no MFMA.
#include <string.h>

#include <cstring>

#include <hip/hip_runtime.h>
// only the device-wide sort and scan are needed (rocprim.hpp also drags in texture iterators)
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "tsdf_common.h"
#define TSDF_MC_TABLE_QUALIFIER __device__
#include "mc_tables.h"

struct McArgs {
  int nx, ny, nz;
  int z_first;            // global index of allocated plane 0
  int z_lo, z_hi;         // cells with z in [z_lo, z_hi)
  int64_t pitch;
  const float *d;
  PlaneView pv;           // weights / colour through tsdf_load_w / tsdf_load_rgb (any layout)
  float w_min, neg;
  int color_mode;
  float lower[3], size_voxel[3];
  int flush_at;                  // flush a wave's LDS list once it holds more than this many cells
  int qpr, log2TX, TX, TY, rpb;  // classify launch shape: TX quads along x, TY rows, rpb row groups per block
};

// getGridValue (:91-106): NaN if w < w_min or |d| >= 1, else d * max_dist_neg.
static __device__ __forceinline__ float grid_value(const McArgs &a, int64_t vi) {
  const float d = a.d[vi], w = tsdf_load_w(a.pv, vi);
  if (w < a.w_min || fabsf(d) >= 1.f) return NAN;
  return d * a.neg;
}

static __device__ __forceinline__ uint64_t spread3(uint64_t v) {  // 21 bits -> every third bit
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

// Corner order of pcl::MarchingCubes: (0,0,0)(1,0,0)(1,0,1)(0,0,1)(0,1,0)(1,1,0)(1,1,1)(0,1,1)
static __device__ __forceinline__ bool corner_values(const McArgs &a, int x, int y, int z, float leaf[8]) {
  const int64_t sy = a.pitch, sz = (int64_t)a.ny * a.pitch;
  const int64_t o = ((int64_t)(z - a.z_first) * a.ny + y) * a.pitch + x;
  const int64_t off[8] = {0, 1, 1 + sz, sz, sy, 1 + sy, 1 + sy + sz, sy + sz};
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    leaf[k] = grid_value(a, o + off[k]);
    ok = ok && !isnan(leaf[k]);
  }
  return ok;
}

static __device__ __forceinline__ int cube_index(const float leaf[8]) {
  int c = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (leaf[k] < 0.f) c |= 1 << k;  // iso level 0 (:74-78)
  return c;
}

// Weights of the quad starting at element o (16-byte aligned), per layout:
// WL 0 = F32W (float plane), 1 = PACKED with colour (count in byte 3 of the colour word), 2 = PACKED
// without colour (uint8 count plane).
template <int WL>
static __device__ __forceinline__ void load_w4(const PlaneView &pv, int64_t o, float w[4]) {
  if (WL == 0) {
    const float4 w4 = *reinterpret_cast<const float4 *>(pv.w + o);
    w[0] = w4.x, w[1] = w4.y, w[2] = w4.z, w[3] = w4.w;
  } else if (WL == 1) {
    const uint4 c4 = *reinterpret_cast<const uint4 *>(pv.rgb + o);
    w[0] = tsdf_decode_w(c4.x >> 24, pv.wmax), w[1] = tsdf_decode_w(c4.y >> 24, pv.wmax);
    w[2] = tsdf_decode_w(c4.z >> 24, pv.wmax), w[3] = tsdf_decode_w(c4.w >> 24, pv.wmax);
  } else {
    const uint32_t k4 = *reinterpret_cast<const uint32_t *>(pv.k8 + o);
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = tsdf_decode_w((k4 >> (8 * j)) & 255u, pv.wmax);
  }
}

// Classify: a streaming pass over d and w.  A thread owns a quad of 4 x-consecutive base voxels (one
// 16-byte load per plane) and walks `rpb` rows; a wave touches 1 KiB contiguous per plane, like
// k_integrate.  Quads with no candidate voxel (:192: w >= w_min && |d| < 1) -- almost all of the grid --
// cost one load of d.  A quad with a candidate fetches the distances of the other three rows of its 2x2 row
// bundle (L1/L2 hits: the neighbouring thread / the block one plane up streams them anyway), and its weights
// only if one of its up to 4 cells has corners of both signs; active cells go to a wave-private LDS list, flushed to the
// global (Morton key, packed cell) arrays with ONE atomic per flush and coalesced stores.
// counters[0] = active cells, counters[1] = triangles.
#define MC_WAVE_BUF 512  // entries per wave; one append adds at most 256
template <int WL>
static __global__ void __launch_bounds__(256)
k_mc_classify(const McArgs a, uint64_t *__restrict__ keys, uint64_t *__restrict__ vals, uint64_t capacity,
              unsigned long long *__restrict__ counters) {
  __shared__ unsigned char s_ntri[256];
  __shared__ uint64_t s_buf[4][MC_WAVE_BUF];
  s_ntri[threadIdx.x] = mc_ntri_table[threadIdx.x];
  __syncthreads();
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  volatile uint64_t *buf = s_buf[tid >> 6];
  const int tx = (int)(tid & (unsigned)(a.TX - 1));
  const int ty = (int)(tid >> a.log2TX);
  const int xq = (int)blockIdx.x * a.TX + tx;
  const int x4 = xq * 4;
  const int z = a.z_lo + (int)blockIdx.z;
  const int64_t sz = (int64_t)a.ny * a.pitch;
  const int64_t zbase = (int64_t)(z - a.z_first) * sz;
  const bool tail = x4 + 4 < (int)a.pitch;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  unsigned n_buf = 0;    // wave-uniform: entries waiting in buf
  unsigned tri_sum = 0;  // per lane

  auto flush = [&]() {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&counters[0], (unsigned long long)n_buf);
    base = __shfl(base, 0);
    for (unsigned i = lane; i < n_buf; i += 64u) {
      const uint64_t v = buf[i];
      const unsigned long long slot = base + i;
      if (slot < capacity) {
        const uint64_t x = v & 0xfffffull, y = (v >> 20) & 0xfffffull, zz = (v >> 40) & 0xfffffull;
        keys[slot] = (spread3(x) << 2) | (spread3(y) << 1) | spread3(zz);
        vals[slot] = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // reads done before the next append overwrites
    n_buf = 0;
  };

  const int y0 = 1 + (int)blockIdx.y * a.rpb * a.TY + ty;
  // One row of this thread's quad column; d4 was loaded by the caller (4 rows are in flight at a time: a
  // quad that fails the distance test costs nothing but that load, so the loop is pure memory latency
  // unless several loads overlap).
  auto row = [&](int r, const float4 d4) {
    const int y = y0 + r * a.TY;
    unsigned nt[4] = {0u, 0u, 0u, 0u};
    if (xq < a.qpr && y < a.ny - 1) {
      const int64_t o = zbase + (int64_t)y * a.pitch + x4;
      const float dq[4] = {d4.x, d4.y, d4.z, d4.w};
      // :192 and :199-202 (base voxel strictly inside the grid).  Three filters, cheapest first:
      //  1. |d| < 1 at the base voxel: free space (d at the hinge) and unobserved voxels (d = -1) fail, so most of
      //     the grid costs one load of d;
      //  2. the SIGNS of the eight corner values d * max_dist_neg (:91-106 for valid corners): a cell whose
      //     corners agree in sign has case 0 or 255 and emits nothing whatever its weights -- that is the whole
      //     truncation band except the one or two cells the surface actually crosses -- so the band needs the
      //     distances of its 2x2 row bundle but not the weights;
      //  3. only cells with a mixed case load the bundle's weights: every corner must be valid
      //     (w >= w_min && |d| < 1, :91-106,145-177), which includes the base voxel's own test (:192).
      bool cand[4], any = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        cand[j] = fabsf(dq[j]) < 1.f && x4 + j >= 1 && x4 + j < a.nx - 1;
        any |= cand[j];
      }
      if (any) {
        const int64_t ro[4] = {o, o + a.pitch, o + sz, o + sz + a.pitch};  // rows (y,z) (y+1,z) (y,z+1) (y+1,z+1)
        float dr[4][5];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float4 q = rr == 0 ? d4 : *reinterpret_cast<const float4 *>(a.d + ro[rr]);
          dr[rr][0] = q.x, dr[rr][1] = q.y, dr[rr][2] = q.z, dr[rr][3] = q.w;
          dr[rr][4] = tail ? a.d[ro[rr] + 4] : 1.f;
        }
        unsigned sg[4];  // bit x: value of voxel x4 + x in this row is negative
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          sg[rr] = 0u;
#pragma unroll
          for (int x = 0; x < 5; ++x) sg[rr] |= (dr[rr][x] * a.neg < 0.f ? 1u : 0u) << x;
        }
        // pcl::MarchingCubes corner order (0,0,0)(1,0,0)(1,0,1)(0,0,1)(0,1,0)(1,1,0)(1,1,1)(0,1,1) as (dx,dy,dz)
        auto corners = [](const unsigned m[4], int j) -> unsigned {
          return ((m[0] >> j) & 1u) | (((m[0] >> (j + 1)) & 1u) << 1) | (((m[2] >> (j + 1)) & 1u) << 2) |
                 (((m[2] >> j) & 1u) << 3) | (((m[1] >> j) & 1u) << 4) | (((m[1] >> (j + 1)) & 1u) << 5) |
                 (((m[3] >> (j + 1)) & 1u) << 6) | (((m[3] >> j) & 1u) << 7);
        };
        unsigned ci[4];
        any = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ci[j] = corners(sg, j);
          cand[j] = cand[j] && ci[j] != 0u && ci[j] != 255u;
          any |= cand[j];
        }
        if (any) {
          unsigned vm[4];  // bit x: voxel x4 + x of this row is a valid grid value
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            float w[5];
            load_w4<WL>(a.pv, ro[rr], w);
            w[4] = tail ? tsdf_load_w(a.pv, ro[rr] + 4) : 0.f;
            vm[rr] = 0u;
#pragma unroll
            for (int x = 0; x < 5; ++x) vm[rr] |= (w[x] >= a.w_min && fabsf(dr[rr][x]) < 1.f ? 1u : 0u) << x;
            if (!tail) vm[rr] &= 15u;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (cand[j] && corners(vm, j) == 255u) nt[j] = s_ntri[ci[j]];
        }
      }
    }
    const unsigned cnt = (nt[0] > 0) + (nt[1] > 0) + (nt[2] > 0) + (nt[3] > 0);
    const unsigned long long b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u);
    if ((b0 | b1 | b2) == 0) return;
    unsigned pos = n_buf + (unsigned)__popcll(b0 & lanes_below) + 2u * (unsigned)__popcll(b1 & lanes_below) +
                   4u * (unsigned)__popcll(b2 & lanes_below);
    n_buf += (unsigned)__popcll(b0) + 2u * (unsigned)__popcll(b1) + 4u * (unsigned)__popcll(b2);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (nt[j]) {
        buf[pos++] = (uint64_t)(x4 + j) | ((uint64_t)y << 20) | ((uint64_t)z << 40) | ((uint64_t)nt[j] << 60);
        tri_sum += nt[j];
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if ((int)n_buf > a.flush_at) flush();
  };
  const bool col_ok = xq < a.qpr;
  for (int rg = 0; rg < a.rpb; rg += 4) {
    if (y0 - ty + rg * a.TY >= a.ny - 1) break;  // whole block past the last cell row (uniform)
    float4 d4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int y = y0 + (rg + u) * a.TY;
      d4[u] = make_float4(1.f, 1.f, 1.f, 1.f);  // |d| >= 1: no candidate
      if (col_ok && rg + u < a.rpb && y < a.ny - 1)
        d4[u] = *reinterpret_cast<const float4 *>(a.d + zbase + (int64_t)y * a.pitch + x4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (rg + u < a.rpb) row(rg + u, d4[u]);
  }
  if (n_buf) flush();
  if (__ballot(tri_sum > 0)) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) tri_sum += __shfl_xor(tri_sum, s);
    if (lane == 0) atomicAdd(&counters[1], (unsigned long long)tri_sum);
  }
}

static __global__ void __launch_bounds__(256)
k_mc_counts(const uint64_t *__restrict__ vals, uint32_t *__restrict__ counts, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) counts[i] = (uint32_t)(vals[i] >> 60);
}

static __global__ void __launch_bounds__(256)
k_mc_emit(const McArgs a, const uint64_t *__restrict__ vals, const uint32_t *__restrict__ offsets, uint64_t n_cells,
          float *__restrict__ verts, unsigned char *__restrict__ rgb_out, uint64_t *__restrict__ cell_out) {
  __shared__ signed char s_tri[256 * 16];
  __shared__ unsigned short s_edge[256];
  for (int i = threadIdx.x; i < 256 * 16; i += blockDim.x) s_tri[i] = mc_tri_table[i >> 4][i & 15];
  s_edge[threadIdx.x] = mc_edge_table[threadIdx.x];
  __syncthreads();
  const uint64_t ci = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= n_cells) return;
  const uint64_t v = vals[ci];
  const int x = (int)(v & 0xfffff), y = (int)((v >> 20) & 0xfffff), z = (int)((v >> 40) & 0xfffff);
  float leaf[8];
  corner_values(a, x, y, z, leaf);
  const int cubeindex = cube_index(leaf);
  // createSurface [PCL-recall]: centre = lower_boundary_ + size_voxel_ * index; corner k adds size_voxel_
  // in y if k&4, in z if k&2, in x if (k&1)^((k>>1)&1)
  const int idx[3] = {x, y, z};
  float center[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) center[k] = a.lower[k] + a.size_voxel[k] * (float)idx[k];
  float pc[8][3];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    pc[k][0] = ((k & 1) ^ ((k >> 1) & 1)) ? center[0] + a.size_voxel[0] : center[0];
    pc[k][1] = (k & 4) ? center[1] + a.size_voxel[1] : center[1];
    pc[k][2] = (k & 2) ? center[2] + a.size_voxel[2] : center[2];
  }
  const int ea[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, eb[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
  float vl[12][3];
  const unsigned edges = s_edge[cubeindex];
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    // interpolateEdge: mu = (iso - v1) / (v2 - v1); out = p1 + mu * (p2 - p1)
    const float mu = (0.f - leaf[ea[e]]) / (leaf[eb[e]] - leaf[ea[e]]);
#pragma unroll
    for (int k = 0; k < 3; ++k) vl[e][k] = pc[ea[e]][k] + mu * (pc[eb[e]][k] - pc[ea[e]][k]);
  }
  (void)edges;  // unused edges produce garbage that no triangle references
  unsigned char col[3] = {0, 0, 0};
  const int64_t vi = ((int64_t)(z - a.z_first) * a.ny + y) * a.pitch + x;
  if (a.color_mode == 2) {  // :217-224 colour by confidence, evaluated in double like the reference
    const float std_dev = (float)((100. - (double)tsdf_load_w(a.pv, vi)) / 100.);
    const double r = (double)(1 - std_dev) * 255., b = (double)std_dev * 255.;
    const double rmin = (255. < r) ? 255. : r, bmin = (255. < b) ? 255. : b;  // std::min(x, 255.)
    col[0] = (unsigned char)((0. < rmin) ? rmin : 0.);                        // std::max(0., x)
    col[2] = (unsigned char)((0. < bmin) ? bmin : 0.);
  } else if (a.color_mode == 1 && a.pv.rgb) {  // :226-231
    const uint32_t c = tsdf_load_rgb(a.pv, vi);
    col[0] = (unsigned char)(c & 255u);
    col[1] = (unsigned char)((c >> 8) & 255u);
    col[2] = (unsigned char)((c >> 16) & 255u);
  }
  uint64_t t = offsets[ci];
  const signed char *tri = s_tri + cubeindex * 16;
  for (int i = 0; tri[i] != -1; i += 3, ++t) {
#pragma unroll
    for (int vtx = 0; vtx < 3; ++vtx) {
      const int e = tri[i + vtx];
      float *o = verts + 9 * t + 3 * vtx;
      // select the edge vertex without dynamic register indexing
      float vx = 0, vy = 0, vz = 0;
#pragma unroll
      for (int q = 0; q < 12; ++q)
        if (q == e) {
          vx = vl[q][0];
          vy = vl[q][1];
          vz = vl[q][2];
        }
      o[0] = vx;
      o[1] = vy;
      o[2] = vz;
      if (rgb_out) {
        unsigned char *c = rgb_out + 9 * t + 3 * vtx;
        c[0] = col[0];
        c[1] = col[1];
        c[2] = col[2];
      }
    }
    if (cell_out) cell_out[t] = ((uint64_t)x << 42) | ((uint64_t)y << 21) | (uint64_t)z;
  }
}

static float host_voxel_center(const tsdf_params &p, int a, int i) {  // tsdf_volume_octree.cpp:553-560
  const float off = p.size[a] / 2.0;
  return (float)(((size_t)i + 0.5) * p.size[a] / (double)p.res[a] - off);
}

template <typename T>
static int ensure_buf(T **buf, size_t *cap_elems, size_t need, hipStream_t s) {
  if (need <= *cap_elems && *buf) return TSDF_HIP_OK;
  if (*buf) {
    TSDF_HIP_TRY(hipStreamSynchronize(s));
    TSDF_HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *cap_elems = 0;
  }
  TSDF_HIP_TRY(hipMalloc(buf, need * sizeof(T)));
  *cap_elems = need;
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_march(tsdf_handle h, float w_min, int color_mode, uint64_t *n_tri) {
  if (!h || color_mode < 0 || color_mode > 2) return TSDF_HIP_E_INVALID;
  TSDF_ON_DEVICE(h->device);
  const tsdf_params &p = h->p;
  if (p.res[0] >= (1 << 20) || p.res[1] >= (1 << 20) || p.res[2] >= (1 << 20)) return TSDF_HIP_E_UNSUPPORTED;
  McArgs a;
  a.nx = h->nx;
  a.ny = h->ny;
  a.nz = h->nz;
  a.z_first = h->z_first;
  a.pitch = h->pitch;
  a.d = h->d;
  a.pv = tsdf_plane_view(h);
  a.w_min = w_min;
  a.neg = p.max_dist_neg;
  a.color_mode = color_mode;
  // setInputTSDF (:44-83): the two +- terms at :64-66 cancel, so the bounding box is [centre(voxel 0),
  // centre(voxel res)] and size_voxel_ = (upper - lower) * (1 / res) in float
  for (int k = 0; k < 3; ++k) {
    a.lower[k] = host_voxel_center(p, k, 0);
    const float upper = host_voxel_center(p, k, p.res[k]);
    a.size_voxel[k] = (upper - a.lower[k]) * (1.0f / (float)p.res[k]);
  }
  // cells owned by this handle: base voxel z in the slab, strictly inside the grid (:199-202), and plane
  // z+1 must be allocated (own slab or halo)
  a.z_lo = std::max(1, h->z_begin);
  a.z_hi = std::min(h->nz - 1, h->z_end);
  if (a.z_hi > a.z_lo && a.z_hi + 1 > h->z_first + h->nz_alloc) {
    tsdf_set_error("marching cubes needs plane z_end as a halo (create the handle with halo >= 1)");
    return TSDF_HIP_E_INVALID;
  }
  h->mc_ntri = 0;
  if (n_tri) *n_tri = 0;
  if (a.z_hi <= a.z_lo || a.nx < 3 || a.ny < 3) return TSDF_HIP_OK;

  a.qpr = (a.nx + 3) / 4;
  a.log2TX = 0;
  while ((1 << a.log2TX) < a.qpr && a.log2TX < 8) ++a.log2TX;
  a.TX = 1 << a.log2TX;
  a.TY = 256 / a.TX;
  a.rpb = std::max(1, tsdf_tuning().rows_per_block / a.TY);
  a.flush_at = std::min(MC_WAVE_BUF - 256, std::max(0, tsdf_tuning().mc_flush_at));
  const int cell_rows = a.ny - 2;
  const dim3 block(256), grid((unsigned)((a.qpr + a.TX - 1) / a.TX),
                              (unsigned)((cell_rows + a.rpb * a.TY - 1) / (a.rpb * a.TY)), (unsigned)(a.z_hi - a.z_lo));
  if (grid.y > 65535u || grid.z > 65535u) return TSDF_HIP_E_UNSUPPORTED;
  unsigned long long counts[2] = {0, 0};
  // pass 1 with the capacity we already have; if the surface turned out larger, grow and repeat
  for (int attempt = 0; attempt < 2; ++attempt) {
    const size_t cap = h->mc_cells_cap;
    TSDF_HIP_TRY(hipMemsetAsync(h->counter, 0, 2 * sizeof(unsigned long long), h->stream));
    if (!h->packed)
      hipLaunchKernelGGL(k_mc_classify<0>, grid, block, 0, h->stream, a, h->mc_keys, h->mc_vals, (uint64_t)cap, h->counter);
    else if (h->rgb)
      hipLaunchKernelGGL(k_mc_classify<1>, grid, block, 0, h->stream, a, h->mc_keys, h->mc_vals, (uint64_t)cap, h->counter);
    else
      hipLaunchKernelGGL(k_mc_classify<2>, grid, block, 0, h->stream, a, h->mc_keys, h->mc_vals, (uint64_t)cap, h->counter);
    TSDF_HIP_TRY(hipGetLastError());
    TSDF_HIP_TRY(hipMemcpyAsync(counts, h->counter, sizeof counts, hipMemcpyDeviceToHost, h->stream));
    TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
    if (counts[0] <= cap) break;
    const size_t need = (size_t)counts[0] + (size_t)counts[0] / 8 + 1024;
    size_t c1 = h->mc_cells_cap, c2 = h->mc_cells_cap;
    int rc = ensure_buf(&h->mc_keys, &c1, need, h->stream);
    if (rc) return rc;
    rc = ensure_buf(&h->mc_vals, &c2, need, h->stream);
    if (rc) return rc;
    h->mc_cells_cap = need;
  }
  const uint64_t n_cells = counts[0], ntri = counts[1];
  if (n_cells == 0) return TSDF_HIP_OK;
  if (n_cells > 0xffffffffull || ntri > 0xffffffffull) {
    tsdf_set_error("mesh too large (more than 2^32 cells or triangles)");
    return TSDF_HIP_E_UNSUPPORTED;
  }

  // sort (key, val) by Morton key -> reference triangle order; only the bits a coordinate can set take part
  int coord_bits = 1;
  while ((1 << coord_bits) < std::max(a.nx, std::max(a.ny, a.nz))) ++coord_bits;
  const unsigned key_bits = 3u * (unsigned)coord_bits;
  uint64_t *keys_out = nullptr, *vals_out = nullptr;
  uint32_t *cnt = nullptr, *off = nullptr;
  size_t tmp_bytes_sort = 0, tmp_bytes_scan = 0;
  TSDF_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes_sort, h->mc_keys, keys_out, h->mc_vals, vals_out,
                                         (size_t)n_cells, 0, key_bits, h->stream));
  TSDF_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes_scan, cnt, off, 0u, (size_t)n_cells,
                                       rocprim::plus<uint32_t>(), h->stream));
  const size_t al = 256;
  auto up = [&](size_t v) { return (v + al - 1) / al * al; };
  const size_t b_keys = up(n_cells * 8), b_cnt = up(n_cells * 4);
  const size_t total = 2 * b_keys + 2 * b_cnt + up(std::max(tmp_bytes_sort, tmp_bytes_scan));
  int rc = tsdf_ensure_scratch(h, total);
  if (rc) return rc;
  char *sp = (char *)h->scratch;
  keys_out = (uint64_t *)sp;
  vals_out = (uint64_t *)(sp + b_keys);
  cnt = (uint32_t *)(sp + 2 * b_keys);
  off = (uint32_t *)(sp + 2 * b_keys + b_cnt);
  void *tmp = sp + 2 * b_keys + 2 * b_cnt;
  TSDF_HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes_sort, h->mc_keys, keys_out, h->mc_vals, vals_out,
                                         (size_t)n_cells, 0, key_bits, h->stream));
  const unsigned cell_blocks = (unsigned)((n_cells + 255) / 256);
  hipLaunchKernelGGL(k_mc_counts, dim3(cell_blocks), dim3(256), 0, h->stream, vals_out, cnt, n_cells);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(rocprim::exclusive_scan(tmp, tmp_bytes_scan, cnt, off, 0u, (size_t)n_cells,
                                       rocprim::plus<uint32_t>(), h->stream));

  // output buffers
  if (ntri > h->mc_cap) {
    TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->mc_verts) (void)hipFree(h->mc_verts);
    if (h->mc_rgb) (void)hipFree(h->mc_rgb);
    if (h->mc_cell) (void)hipFree(h->mc_cell);
    h->mc_verts = nullptr;
    h->mc_rgb = nullptr;
    h->mc_cell = nullptr;
    h->mc_cap = 0;
    const size_t cap = (size_t)ntri + (size_t)ntri / 8 + 1024;
    TSDF_HIP_TRY(hipMalloc(&h->mc_verts, cap * 9 * sizeof(float)));
    TSDF_HIP_TRY(hipMalloc(&h->mc_rgb, cap * 9));
    TSDF_HIP_TRY(hipMalloc(&h->mc_cell, cap * sizeof(uint64_t)));
    h->mc_cap = cap;
  }
  hipLaunchKernelGGL(k_mc_emit, dim3(cell_blocks), dim3(256), 0, h->stream, a, vals_out, off, n_cells, h->mc_verts,
                     color_mode ? h->mc_rgb : nullptr, h->mc_cell);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  h->mc_ntri = ntri;
  h->mc_has_rgb = color_mode != 0;
  if (n_tri) *n_tri = ntri;
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_march_fetch(tsdf_handle h, float *verts, uint8_t *rgb, uint64_t *cell) {
  if (!h) return TSDF_HIP_E_INVALID;
  TSDF_ON_DEVICE(h->device);
  const size_t n = (size_t)h->mc_ntri;
  if (!n) return TSDF_HIP_OK;
  if (rgb && !h->mc_has_rgb) {
    tsdf_set_error("the last tsdf_hip_march ran without a colour mode");
    return TSDF_HIP_E_INVALID;
  }
  int rc = TSDF_HIP_OK;
  if (verts && (rc = tsdf_to_host(h, verts, h->mc_verts, n * 9 * sizeof(float)))) return rc;
  if (rgb && (rc = tsdf_to_host(h, rgb, h->mc_rgb, n * 9))) return rc;
  if (cell && (rc = tsdf_to_host(h, cell, h->mc_cell, n * sizeof(uint64_t)))) return rc;
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  return TSDF_HIP_OK;
}

// The same mesh into DEVICE buffers owned by the caller (asynchronous on the handle's stream): what a rank of
// a Z-slab job hands to RCCL when the per-slab meshes are merged on the GPU.
extern "C" int tsdf_hip_march_fetch_device(tsdf_handle h, float *d_verts, uint8_t *d_rgb, uint64_t *d_cell) {
  if (!h) return TSDF_HIP_E_INVALID;
  TSDF_ON_DEVICE(h->device);
  const size_t n = (size_t)h->mc_ntri;
  if (!n) return TSDF_HIP_OK;
  if (d_verts)
    TSDF_HIP_TRY(hipMemcpyAsync(d_verts, h->mc_verts, n * 9 * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  if (d_rgb) {
    if (!h->mc_has_rgb) {
      tsdf_set_error("the last tsdf_hip_march ran without a colour mode");
      return TSDF_HIP_E_INVALID;
    }
    TSDF_HIP_TRY(hipMemcpyAsync(d_rgb, h->mc_rgb, n * 9, hipMemcpyDeviceToDevice, h->stream));
  }
  if (d_cell) TSDF_HIP_TRY(hipMemcpyAsync(d_cell, h->mc_cell, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, h->stream));
  return TSDF_HIP_OK;
}
