// libtsdf_hip.so -- marching cubes on the flat SoA grid.
//
// Replaces MarchingCubesTSDFOctree::reconstruct (src/lib/marching_cubes_tsdf_octree.cpp:108-236) and the
// PCL pieces it calls (pcl::MarchingCubes::createSurface / interpolateEdge, Bourke's tables).
//   k_mc_classify  ONE streaming pass over the distance plane, marching along z: candidate test (:192-202), the
//                  signs and |d| < 1 of the 8 corner values (getValidNeighborList1D :145-177 / getGridValue
//                  :91-106) as bit masks, case index, triangle count; candidate cells collect in wave-private LDS
//                  lists, their corner weights are tested when a list is flushed (all lanes busy), survivors become
//                  one word each: (Morton key << 4) | triangle count
//   rocprim sort   of those words by key: the reference emits triangles in octree pre-order with child index
//                  4*(x>cx) + 2*(y>cy) + (z>cz) (octree.cpp:119,257-264) = Morton order, x the high bit
//   rocprim scan   triangle offsets
//   k_mc_emit      256 active cells per block, output loop over the block's vertices: edge interpolation + triangle/colour output
// Case tables live in LDS.  Streaming stencil read of d (w only at the surface): HBM-bound, no MFMA.
#include <string.h>

#include <cstring>

#include <hip/hip_runtime.h>
// only the device-wide sort and scan are needed (rocprim.hpp also drags in texture iterators)
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "tsdf_common.h"
#ifdef TSDF_HIP_TEST_HOOKS
#include "tsdf_hip_test.h"
#endif
#include "tsdf_buffer.h"
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
  int qpr;                       // quads per row = ceil(nx / 4)
  int check_w;                   // the weight test can fail (0: PACKED layout and w_min <= 0, every count passes)
  int zb;                        // cell planes a classify block marches (<= MC_ZB; zb + 1 planes must span < 4 GB)
  // What k_mc_need derived from the "band seen" flags (NULL: read everything).  need[(z - z_lo) * need_rows + wave row]
  // [x-chunk] = bits 0-3: the wave's four 16-lane groups (64 voxels each) must load their quads of plane z, bit 4: the
  // one column beyond the wave's last quad; need_blk[(blockIdx.z * grid.y + block row) * grid.x + x-chunk] != 0: the
  // block has anything to do at all.
  const uint8_t *need, *need_blk;
  int need_gx, need_rows, need_by;
};

static __device__ __forceinline__ uint64_t spread3(uint64_t v) {  // 21 bits -> every third bit
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

static __device__ __forceinline__ uint32_t compact3(uint64_t v) {  // every third bit -> 21 bits (spread3's inverse)
  v &= 0x1249249249249249ull;
  v = (v | v >> 2) & 0x10c30c30c30c30c3ull;
  v = (v | v >> 4) & 0x100f00f00f00f00full;
  v = (v | v >> 8) & 0x1f0000ff0000ffull;
  v = (v | v >> 16) & 0x1f00000000ffffull;
  v = (v | v >> 32) & 0x1fffffull;
  return (uint32_t)v;
}
// An active cell is ONE 64-bit word since round 6: (Morton key << MC_KEY_SHIFT) | triangle count.  The key IS the cell's
// coordinates (x the high bit of every triple: octree.cpp:119,257-264), so the (key, packed cell) PAIRS of rounds 1-5 carried them
// twice; sorting words instead of pairs halves what the radix sort moves (the count's bits take no part in it).  mc_unpack
// gives the packed form the rest of the file reads: x | y << 20 | z << 40 | count << 60.
#define MC_KEY_SHIFT 4
#define MC_STAT_COUNTS_PASS (1ull << 63)  // in tsdf_hip_volume::mc_d_bytes, see tsdf_hip_march
static __device__ __forceinline__ uint64_t mc_unpack(uint64_t word) {
  const uint64_t key = word >> MC_KEY_SHIFT;
  return (uint64_t)compact3(key >> 2) | ((uint64_t)compact3(key >> 1) << 20) | ((uint64_t)compact3(key) << 40) | ((word & 15ull) << 60);
}

static __device__ __forceinline__ int cube_index(const float leaf[8]) {
  int c = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (leaf[k] < 0.f) c |= 1 << k;  // iso level 0 (:74-78)
  return c;
}

// Classify: ONE streaming pass over the distance plane d, nothing else.
//
// A cell's case index depends only on the SIGNS of its eight corner values d * max_dist_neg (:91-106 for valid
// corners; createSurface compares them with iso level 0), and seven of the eight validity tests are |d| < 1.  So a
// thread reduces every quad of 4 x-consecutive distances it loads to two 4-bit masks (negative / inside the
// truncation band) and the whole cell logic runs on bits.  A wave owns a tile of 64 quads (1 KiB of a row) by
// MC_R rows and MARCHES along z: the masks of plane z stay in registers while plane z + 1 streams in, so every
// distance is loaded once (plus one halo row per block of 4 * MC_R rows and one plane per MC_ZB).  Neighbours along x
// come from the next lane (one shuffle of the packed bit-0 column; lane 63 loads its five halo words itself),
// neighbours along z are the thread's own registers, and the row below a wave's last cells is the next wave's first
// row: its 10 mask bits per lane cross through LDS, one barrier per plane.  (A first version had every wave load
// that row itself and rely on the cache: PMC showed 44 GB fetched for a 34 GB plane, waves of a block drift apart.)
// No dependent global load: the only thing a cell that might emit triangles (mixed signs, all eight corners inside
// the band, base voxel strictly inside the grid :199-202) costs is an append to the wave-private LDS list.
//
// The eighth test, w >= w_min at the eight corners (:91-106,145-177, which includes the base voxel's :192), is
// deferred to the flush of that list, where every lane holds one listed cell: the weight gathers run with all 64
// lanes busy instead of stalling a whole wave for the one lane whose quad touches a surface (a box face
// perpendicular to x puts exactly one such lane into every wave of its x-chunk; that was what bound the previous
// version: 11.8 ms at 2048^3).  Cells that pass go to the global (Morton key, packed cell) arrays, one atomic per
// 64 cells.  counters[0] = active cells, counters[1] = triangles.
#ifndef MC_WAVE_BUF
#define MC_WAVE_BUF 1024  // 32-bit entries per wave; HALF a plane step of a wave (two cell rows) adds at most 512
#endif
#define MC_R 4            // cell rows per wave
#ifndef MC_ZB
#define MC_ZB 32          // cell planes per block
#endif
#define MC_COL0 0x0108421u  // bit 0 of each of the five 5-bit row groups

// Which parts of which planes can a classify wave skip?  Marching cubes emits a cell only if one of its eight corners
// is negative (cube index != 0), and a distance only turns negative through an observation inside the truncation
// band, which the integrate kernels record per cell of 64 x 4 x 1 voxels (tsdf_hip_volume::band).  A voxel is a
// corner of a cell that may emit only if a flagged voxel lies within one step of it along every axis, so a wave needs
// the quads of a 16-lane group (64 voxels x its MC_R + 1 rows, plane z) only if a flag is set among the flag cells
// that touch that box grown by one voxel.  Planes outside the handle's own slab (halo planes, filled by copies) have
// no flags and count as set.  One thread per (x-chunk, wave row, plane); all 32-bit-safe sizes.
#define MC_NEED_SLOTS 64
struct NeedArgs {
  const uint8_t *band;
  int fx, fy;                 // flag cells along x / y
  int z_first, nz_alloc;      // allocated planes
  int z_begin, z_end;         // owned planes: only these have flags
  int z_lo, n_planes;         // planes z_lo .. z_lo + n_planes - 1 are classified (cells z_lo .. z_hi - 1 read one more)
  int nx, ny;
  int gx, rows, by;           // x-chunks, wave rows (4 per block row), block rows
  int zb;
};

static __global__ void __launch_bounds__(256)
k_mc_need(const NeedArgs n, uint8_t *__restrict__ need, uint8_t *__restrict__ need_blk,
          unsigned long long *__restrict__ d_bytes) {
  const int64_t t_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n.gx * n.rows * n.n_planes;
  const bool live = t_raw < total;  // (the last wave's spare lanes stay for the wave-wide sum below)
  const int64_t t = live ? t_raw : total - 1;
  const int bx = (int)(t % n.gx), wrow = (int)((t / n.gx) % n.rows), zi = (int)(t / ((int64_t)n.gx * n.rows));
  const int z = n.z_lo + zi;
  const int yw = 1 + wrow * MC_R;                      // the wave's rows yw .. yw + MC_R (the last one is its halo row)
  const int fy0 = max(0, (yw - 1) >> 2), fy1 = min(n.fy - 1, (yw + MC_R + 1) >> 2);
  // One pass over the <= 6 x 2 x 3 flag cells that can matter: m6 bit c = "a flag is set in x cell 4 bx - 1 + c" (over
  // the row groups and the three planes).  A 16-lane group g (x cells 4 bx + g) grown by one voxel touches cells
  // g - 1 .. g + 1 -> m6 bits g, g + 1, g + 2; the halo column (voxel 256 bx + 256) touches cells 4 bx + 3, 4 bx + 4.
  unsigned m6 = 0u;
  bool halo_plane = false;
  if (yw < n.ny && fy0 <= fy1) {
    for (int zz = z - 1; zz <= z + 1; ++zz) {
      if (zz < n.z_first || zz >= n.z_first + n.nz_alloc) continue;
      if (zz < n.z_begin || zz >= n.z_end) {  // a halo plane: unknown, so everything near it is needed
        halo_plane = true;
        continue;
      }
      const uint8_t *pl = n.band + (int64_t)(zz - n.z_first) * n.fy * n.fx;
      for (int fy = fy0; fy <= fy1; ++fy)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const int fx = 4 * bx - 1 + c;
          if (fx >= 0 && fx < n.fx && pl[fy * n.fx + fx]) m6 |= 1u << c;
        }
    }
  }
  unsigned bits = 0u;
  if (yw < n.ny) {
    if (halo_plane) m6 = 63u;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (bx * 256 + g * 64 < n.nx && ((m6 >> g) & 7u)) bits |= 1u << g;
    if (bx * 256 + 256 < n.nx && ((m6 >> 4) & 3u)) bits |= 16u;
  }
  if (live) need[t] = (uint8_t)bits;
  {  // bytes of the distance plane classify will request for this (wave, plane): 16 lanes x 16 B per row and group,
     // one 4-byte word per row for the halo column; the plane a block ends on is the plane the next block starts on
    const unsigned rows = (unsigned)max(0, min((wrow & 3) == 3 ? MC_R + 1 : MC_R, n.ny - yw));
    const unsigned times = (zi > 0 && zi < n.n_planes - 1 && zi % n.zb == 0) ? 2u : 1u;
    unsigned long long b = live ? (unsigned long long)(__popc(bits & 15u) * 256u + ((bits >> 4) & 1u) * 4u) * rows * times : 0ull;
    for (int o = 32; o; o >>= 1) b += __shfl_xor(b, o);
    // (one global atomic per block, spread over MC_NEED_SLOTS addresses: 130 k wave sums on ONE address cost 1 ms)
    __shared__ unsigned long long s_sum;
    if (threadIdx.x == 0u) s_sum = 0ull;
    __syncthreads();
    if ((threadIdx.x & 63u) == 0u && b) atomicAdd(&s_sum, b);
    __syncthreads();
    if (threadIdx.x == 0u && s_sum) atomicAdd(d_bytes + (blockIdx.x % MC_NEED_SLOTS), s_sum);
  }
  if (live && bits) {  // (every writer stores the same 1); a cell plane z of block zb-index k reads planes up to z + 1
    const int brow = wrow >> 2;
    const int k_hi = min((n.n_planes - 2) / n.zb, zi / n.zb), k_lo = max(0, (zi - 1) / n.zb);
    for (int k = k_lo; k <= k_hi; ++k) need_blk[((int64_t)k * n.by + brow) * n.gx + bx] = 1;
  }
}

// The same verdicts, one thread per (wave row, plane) for ALL the x-chunks of the row (round 6): the <= 2 x 3 flag rows that
// matter are read once, 16 bytes at a time, and folded into one bit per flag cell; every x-chunk's six cells are then a
// shift of that mask.  (k_mc_need above reads 36 single bytes per x-chunk: 0.2 ms at 2048^3 for a 33 MB array.)  For
// grids whose flag rows are whole 16-byte groups and fit 64 bits (nx a multiple of 1024, at most 4096); the others keep
// the kernel above.  Same outputs, compared by tests/test_query_gpu.py::test_marching_cubes_skips_*.
static __global__ void __launch_bounds__(256)
k_mc_need_rows(const NeedArgs n, uint8_t *__restrict__ need, uint8_t *__restrict__ need_blk,
               unsigned long long *__restrict__ d_bytes) {
  const int64_t t_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n.rows * n.n_planes;
  const bool live = t_raw < total;
  const int64_t t = live ? t_raw : total - 1;
  const int wrow = (int)(t % n.rows), zi = (int)(t / n.rows);
  const int z = n.z_lo + zi;
  const int yw = 1 + wrow * MC_R;
  const int fy0 = max(0, (yw - 1) >> 2), fy1 = min(n.fy - 1, (yw + MC_R + 1) >> 2);
  unsigned long long mask = 0ull;  // bit c: a flag is set in x cell c (over the row groups and the three planes)
  bool halo_plane = false;
  if (yw < n.ny && fy0 <= fy1) {
    for (int zz = z - 1; zz <= z + 1; ++zz) {
      if (zz < n.z_first || zz >= n.z_first + n.nz_alloc) continue;
      if (zz < n.z_begin || zz >= n.z_end) {
        halo_plane = true;
        continue;
      }
      const uint8_t *pl = n.band + (int64_t)(zz - n.z_first) * n.fy * n.fx;
      for (int fy = fy0; fy <= fy1; ++fy)
        for (int q = 0; q < n.fx; q += 16) {
          const u4 v = *reinterpret_cast<const u4 *>(pl + fy * n.fx + q);
          const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b)
              if ((w[k] >> (8 * b)) & 255u) mask |= 1ull << (q + 4 * k + b);
        }
    }
  }
  unsigned long long bytes = 0ull;
  const unsigned rows = (unsigned)max(0, min((wrow & 3) == 3 ? MC_R + 1 : MC_R, n.ny - yw));
  const unsigned times = (zi > 0 && zi < n.n_planes - 1 && zi % n.zb == 0) ? 2u : 1u;
  const int brow = wrow >> 2;
  const int k_hi = min((n.n_planes - 2) / n.zb, zi / n.zb), k_lo = max(0, (zi - 1) / n.zb);
  for (int bx = 0; bx < n.gx; ++bx) {
    unsigned m6 = halo_plane ? 63u : (unsigned)((bx ? mask >> (4 * bx - 1) : mask << 1) & 63ull);
    unsigned bits = 0u;
    if (yw < n.ny) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (bx * 256 + g * 64 < n.nx && ((m6 >> g) & 7u)) bits |= 1u << g;
      if (bx * 256 + 256 < n.nx && ((m6 >> 4) & 3u)) bits |= 16u;
    }
    if (live) {
      need[t * n.gx + bx] = (uint8_t)bits;
      bytes += (unsigned long long)(__popc(bits & 15u) * 256u + ((bits >> 4) & 1u) * 4u) * rows * times;
      if (bits)
        for (int k = k_lo; k <= k_hi; ++k) need_blk[((int64_t)k * n.by + brow) * n.gx + bx] = 1;
    }
  }
  for (int o = 32; o; o >>= 1) bytes += __shfl_xor(bytes, o);
  __shared__ unsigned long long s_sum;
  if (threadIdx.x == 0u) s_sum = 0ull;
  __syncthreads();
  if ((threadIdx.x & 63u) == 0u && bytes) atomicAdd(&s_sum, bytes);
  __syncthreads();
  if (threadIdx.x == 0u && s_sum) atomicAdd(d_bytes + (blockIdx.x % MC_NEED_SLOTS), s_sum);
}

template <int WL>  // 0 = F32W (float plane), 1 = PACKED with colour (count in byte 3), 2 = PACKED count plane
static __device__ __forceinline__ float mc_load_w(const PlaneView &pv, int64_t i) {
  if (WL == 0) return pv.w[i];
  return tsdf_decode_w(WL == 1 ? (pv.rgb[i] >> 24) : (unsigned)pv.k8[i], pv.wmax);
}

template <int WL>
static __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_mc_classify(const McArgs a, uint64_t *__restrict__ keys, uint64_t capacity,
              unsigned long long *__restrict__ counters) {
  __shared__ unsigned char s_ntri[256];
  __shared__ uint32_t s_buf[4][MC_WAVE_BUF];
  __shared__ uint32_t s_halo[2][256];
  // Morton-key parts of this block's x (low 8 bits; the rest is block-uniform) / y / z values
  __shared__ uint32_t s_xkey[256];
  __shared__ uint64_t s_ykey[4 * MC_R], s_zkey[MC_ZB];
  s_ntri[threadIdx.x] = mc_ntri_table[threadIdx.x];
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // in an SGPR: row bases stay scalar
  volatile uint32_t *buf = s_buf[wave];
  // Blocks go to the 8 XCDs round-robin by linear id.  With x fastest and 8 x-chunks per row (2048 voxels) every XCD
  // would own ONE x-chunk, and the two chunks that hold the scene's x-facing walls would keep their XCDs busy long
  // after the others finished (measured: half the chip idle on average).  So the linear id is re-read as
  // (row-group low 3 bits, x-chunk, row-group high bits): an XCD sees every x-chunk and every eighth row group.
  const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
  const unsigned bx = (lin >> 3) % gridDim.x, by = ((lin >> 3) / gridDim.x) * 8u + (lin & 7u);
  const int xq = (int)bx * 64 + (int)lane;
  const int x4 = xq * 4;
  const int yw = 1 + ((int)by * 4 + (int)wave) * MC_R;  // first cell row of this wave
  if (1 + (int)by * 4 * MC_R >= a.ny - 1) return;  // the whole BLOCK lies past the last cell row (block-uniform)
  // nothing near this block was ever observed inside the truncation band: no negative distance, no triangle
  if (a.need_blk && !a.need_blk[((int64_t)blockIdx.z * a.need_by + by) * a.need_gx + bx]) return;
  const int zs = a.z_lo + (int)blockIdx.z * a.zb;
  const int ze = min(zs + a.zb, a.z_hi);                       // cell planes [zs, ze); plane ze is read
  s_xkey[tid] = (uint32_t)(spread3((uint64_t)tid) << 2);
  const uint64_t xkey_hi = spread3((uint64_t)bx * 256u) << 2;
  if (tid < 4u * MC_R) s_ykey[tid] = spread3((uint64_t)(1u + by * 4u * MC_R + tid)) << 1;
  if (tid < (unsigned)MC_ZB) s_zkey[tid] = spread3((uint64_t)(zs + (int)tid));
  __syncthreads();
  const int64_t sz = (int64_t)a.ny * a.pitch;
  // k_mc_need's verdicts on this wave's part of planes zs .. ze, one per lane, fetched once: a load per plane step
  // would put a second dependent memory latency into every step of the march
  int nd_all = 31;
  if (a.need) {
    nd_all = 0;
    if ((int)lane <= ze - zs)
      nd_all = (int)a.need[((int64_t)(zs + (int)lane - a.z_lo) * a.need_rows + (int)(by * 4u + wave)) * a.need_gx + (int)bx];
  }
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  unsigned n_buf = 0;    // wave-uniform: entries waiting in buf
  unsigned tri_sum = 0;  // per lane
  // Cells this thread may emit, as a mask over the packed layout below (row r of the wave at bits 5r .. 5r+3):
  // base voxel strictly inside the grid along x and y (:199-202)
  unsigned cell_mask = 0u;
#pragma unroll
  for (int r = 0; r < MC_R; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      cell_mask |= (x4 + j >= 1 && x4 + j < a.nx - 1 && yw + r < a.ny - 1 ? 1u : 0u) << (5 * r + j);
  const bool edge_lane = lane == 63u && x4 + 4 < (int)a.pitch;  // the only lane whose x + 1 neighbour is not in the wave

  const unsigned wbytes = WL == 2 ? 1u : 4u;
  const void *wbase = WL == 0 ? (const void *)(a.pv.w + (int64_t)(zs - a.z_first) * sz)
                    : WL == 1 ? (const void *)(a.pv.rgb + (int64_t)(zs - a.z_first) * sz)
                              : (const void *)(a.pv.k8 + (int64_t)(zs - a.z_first) * sz);
  const rsrc_t rsW = make_rsrc(wbase, (unsigned)(ze + 1 - zs) * (unsigned)sz * wbytes);
  // Rows this wave LOADS: its own MC_R; the row below them (y + 1 of its last cells) is the next wave's first row and
  // arrives through LDS (s_halo) -- only the block's last wave has no such neighbour and loads it itself.  Rows past
  // the grid read 0 through the descriptor's bounds check (a wave wholly past the grid still takes part in the
  // barriers; cell_mask keeps it from emitting anything).
  const int n_load = wave == 3u ? MC_R + 1 : MC_R;
  const unsigned rows_here = (unsigned)max(0, min(n_load, a.ny - yw));
  const unsigned voff = (unsigned)x4 * 4u;
  const unsigned row_bytes = (unsigned)a.pitch * 4u;

  // The deferred weight test + the copy-out.  Every lane takes one listed cell:
  // entry = x - 256 blockIdx.x | row << 8 | (z - zs) << 10 | triangles << 15.
  auto flush = [&]() {
#if defined(MC_PROBE) && MC_PROBE == 4  // probe: the append loop alone -- lists are built and dropped
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    n_buf = 0;
    return;
#endif
    // pass 1 (only if a weight can fail): test the 8 corner weights of every listed cell, 64 cells at a time with every
    // lane busy; a cell that fails loses its triangle count (a listed cell always has one), which marks it dropped
    unsigned total = n_buf;
    if (a.check_w) {
      total = 0u;
      for (unsigned i0 = 0; i0 < n_buf; i0 += 64u) {
        const unsigned i = i0 + lane;
        const bool have = i < n_buf;
        const uint32_t e = have ? buf[i] : 0u;
        // 32-bit offsets into the block's planes [zs, ze] of the weight plane (the host keeps that span < 4 GB)
        const unsigned o = (((e >> 10) & 31u) * (unsigned)a.ny + (unsigned)(yw + (int)((e >> 8) & 3u))) * (unsigned)a.pitch +
                           bx * 256u + (e & 255u);
        const unsigned p1 = (unsigned)a.pitch, s1 = (unsigned)sz;
        const unsigned off[8] = {0u, 1u, 1u + s1, s1, p1, 1u + p1, 1u + p1 + s1, p1 + s1};
        float w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // (a lane without a cell reads cell 0 of the block: in range, ignored)
          if (WL == 0)
            w[k] = __uint_as_float(bload32(rsW, (o + off[k]) * 4u, 0u));
          else if (WL == 1)
            w[k] = tsdf_decode_w(bload32(rsW, (o + off[k]) * 4u, 0u) >> 24, a.pv.wmax);
          else
            w[k] = tsdf_decode_w(bload8(rsW, o + off[k], 0u), a.pv.wmax);
        }
        bool ok = have;
#pragma unroll
        for (int k = 0; k < 8; ++k) ok = ok && !(w[k] < a.w_min);  // getGridValue :98: NaN iff w < w_min (or |d| >= 1)
        if (have && !ok) buf[i] = e & 0x7fffu;
        total += (unsigned)__popcll(__ballot(ok));
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // ONE atomic per flush reserves the output range (all flushes of the launch queue on this one address: it
    // was the kernel's bottleneck when every 64 cells paid for one)
    unsigned long long base = 0;
    if (total) {
      if (lane == 0) base = atomicAdd(&counters[0], (unsigned long long)total);
      base = __shfl(base, 0);
      for (unsigned i0 = 0; i0 < n_buf; i0 += 64u) {
        const unsigned i = i0 + lane;
        const uint32_t e = i < n_buf ? buf[i] : 0u;
        const bool ok = (e >> 15) != 0u;
        const unsigned long long m = __ballot(ok);
        const unsigned long long slot = base + (unsigned long long)__popcll(m & lanes_below);
        base += (unsigned long long)__popcll(m);
        if (ok) {
          tri_sum += e >> 15;
#if defined(MC_PROBE) && MC_PROBE == 5  // probe: everything but the key assembly and its store
          if (slot == ~0ull) {
#else
          if (slot < capacity) {
#endif
            const unsigned xr = e & 255u, row = wave * MC_R + ((e >> 8) & 3u), zr = (e >> 10) & 31u;
            keys[slot] = ((xkey_hi | (uint64_t)s_xkey[xr] | s_ykey[row] | s_zkey[zr]) << MC_KEY_SHIFT) | (uint64_t)(e >> 15);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // reads done before the next append overwrites
    n_buf = 0;
  };

  // Masks of one plane's MC_R + 1 rows, packed: row r occupies bits 5r .. 5r+4; bit 5r + j (j = 0..3) = voxel
  // x4 + j, bit 5r + 4 = voxel x4 + 4 (the next quad's first).  ng: d * max_dist_neg < 0; bd: |d| < 1.
  auto load_plane = [&](int z, unsigned &ng, unsigned &bd) {
    // one descriptor per plane over exactly the rows of the tile that exist: a row past the grid reads 0, which
    // only ever feeds cells that cell_mask excludes (so do lanes past the row's last quad, which read on into
    // the next row)
    const rsrc_t rsD = make_rsrc(a.d + ((int64_t)(z - a.z_first) * a.ny + yw) * a.pitch, rows_here * row_bytes);
    u4 q[MC_R + 1];
    float e[MC_R + 1];
    const u4 outside = {0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};  // 1.f: outside the band, not negative
    // the band flags' verdict on this wave's part of plane z (k_mc_need): quads no flagged voxel is near cannot be a
    // corner of an emitting cell, and read as "outside the band" without being loaded
    const unsigned nd = a.need ? (unsigned)__builtin_amdgcn_readlane(nd_all, z - zs) : 31u;
    const bool ld = ((nd >> (lane >> 4)) & 1u) != 0u;
#pragma unroll
    for (int r = 0; r <= MC_R; ++r) q[r] = outside;
    if (ld) {
#pragma unroll
      for (int r = 0; r < MC_R; ++r) q[r] = bload128(rsD, voff, (unsigned)r * row_bytes);
      if (n_load > MC_R) q[MC_R] = bload128(rsD, voff, (unsigned)MC_R * row_bytes);
    }
#pragma unroll
    for (int r = 0; r <= MC_R; ++r) e[r] = 1.f;
    if (edge_lane && (nd & 16u)) {
#pragma unroll
      for (int r = 0; r < MC_R; ++r) e[r] = __uint_as_float(bload32(rsD, voff + 16u, (unsigned)r * row_bytes));
      if (n_load > MC_R) e[MC_R] = __uint_as_float(bload32(rsD, voff + 16u, (unsigned)MC_R * row_bytes));
    }
    // Most of the grid is free space or unobserved: d sits exactly at +-1 there and no cell that touches such a voxel
    // can emit.  One min over |d| per lane and one ballot decide whether this wave's part of the plane holds ANY
    // voxel inside the band; if not, both masks are zero (a cell needs all eight corners inside the band) and the
    // bit assembly below -- the bulk of this kernel's instructions -- is skipped.
    // (on the bit patterns: |d| < 1 <=> (bits & 0x7fffffff) < bits(1.f); a NaN compares as large, as it must)
    unsigned mn = 0x7f800000u;
#pragma unroll
    for (int r = 0; r <= MC_R; ++r) {
      const unsigned m = 0x7fffffffu;
      mn = min(min(mn, min(q[r].x & m, q[r].y & m)), min(min(q[r].z & m, q[r].w & m), __float_as_uint(e[r]) & m));
    }
    ng = bd = 0u;
    volatile uint32_t *hb = s_halo[z & 1];
#ifndef MC_PROBE
    const bool quiet = __ballot(mn < 0x3f800000u) == 0ull;
#elif MC_PROBE < 3  // bandwidth probes (tools/build_variant.py): 1 = every wave takes the quiet path, 2 = and no barrier
    const bool quiet = __ballot(mn < 0x00000001u) == 0ull;
#else
    const bool quiet = __ballot(mn < 0x3f800000u) == 0ull;
#endif
    if (quiet) {
#if !defined(MC_PROBE) || MC_PROBE < 2
      hb[wave * 64u + lane] = 0u;
      __syncthreads();
#endif
      return;
    }
#pragma unroll
    for (int r = 0; r <= MC_R; ++r) {
      const float dq[4] = {__uint_as_float(q[r].x), __uint_as_float(q[r].y), __uint_as_float(q[r].z), __uint_as_float(q[r].w)};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ng |= dq[j] * a.neg < 0.f ? 1u << (5 * r + j) : 0u;  // createSurface: leaf < iso (0), leaf = d * max_dist_neg
        bd |= fabsf(dq[j]) < 1.f ? 1u << (5 * r + j) : 0u;   // :98
      }
    }
    // the next lane's first column (both masks in one shuffle)
    unsigned nb = (unsigned)__shfl_down((int)((ng & MC_COL0) | ((bd & MC_COL0) << 1)), 1);
    if (lane == 63u) {
      nb = 0u;
#pragma unroll
      for (int r = 0; r <= MC_R; ++r)
        nb |= (e[r] * a.neg < 0.f ? 1u << (5 * r) : 0u) | (fabsf(e[r]) < 1.f ? 2u << (5 * r) : 0u);
    }
    ng |= (nb & MC_COL0) << 4;
    bd |= ((nb >> 1) & MC_COL0) << 4;
    // row MC_R of waves 0..2 = row 0 of the next wave, through LDS: one barrier per plane (two buffers, so a wave
    // that is already writing the next plane's row cannot overwrite what a slower wave still has to read)
    hb[wave * 64u + lane] = (ng & 31u) | ((bd & 31u) << 5);
    __syncthreads();
    if (wave < 3u) {
      const uint32_t h = hb[(wave + 1u) * 64u + lane];
      ng = (ng & ~(31u << (5 * MC_R))) | ((h & 31u) << (5 * MC_R));
      bd = (bd & ~(31u << (5 * MC_R))) | (((h >> 5) & 31u) << (5 * MC_R));
    }
  };

  unsigned n0, b0, n1, b1;
  load_plane(zs, n0, b0);
  for (int z = zs; z < ze; ++z) {
    load_plane(z + 1, n1, b1);
    // all 4 * MC_R cells of the thread at once: rows r and r + 1 are 5 bits apart, columns j and j + 1 one bit
    const unsigned A = n0 & (n0 >> 5) & n1 & (n1 >> 5), O = n0 | (n0 >> 5) | n1 | (n1 >> 5);
    const unsigned V = b0 & (b0 >> 5) & b1 & (b1 >> 5);
    // mixed signs (neither all eight negative nor none) and all eight inside the band
    unsigned cand = ~(A & (A >> 1)) & (O | (O >> 1)) & (V & (V >> 1)) & cell_mask;
#if defined(MC_PROBE) && MC_PROBE == 3  // probe: mask assembly as usual, but nothing is ever listed
    cand &= (unsigned)(z < 0);
#endif
    if (__ballot(cand != 0u)) {
      // appended in two halves (cell rows 0-1, then 2-3): a half adds at most 512 entries, which is the list's size
      for (int half = 0; half < 2; ++half) {
        unsigned c2 = half ? cand >> (5 * (MC_R / 2)) : cand & ((1u << (5 * (MC_R / 2))) - 1u);
        if (!__ballot(c2 != 0u)) continue;
        const unsigned cnt = (unsigned)__popc(c2);  // 0 .. 8
        unsigned pos = 0u, add = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned long long bk = __ballot((cnt >> k) & 1u);
          pos += (unsigned)__popcll(bk & lanes_below) << k;
          add += (unsigned)__popcll(bk) << k;
        }
        if (n_buf && ((int)n_buf > a.flush_at || n_buf + add > MC_WAVE_BUF)) flush();
        pos += n_buf;
        n_buf += add;
        const unsigned shift = half ? 5u * (MC_R / 2) : 0u;
        while (c2) {
          const unsigned bit = (unsigned)__builtin_ctz(c2) + shift;
          c2 &= c2 - 1u;
          const unsigned r = (bit * 13u) >> 6, j = bit - 5u * r;  // bit / 5 for bit < 25
          const unsigned p = n0 >> bit, q = n1 >> bit;
          // pcl::MarchingCubes corner order (0,0,0)(1,0,0)(1,0,1)(0,0,1)(0,1,0)(1,1,0)(1,1,1)(0,1,1) as (dx,dy,dz)
          const unsigned ci = (p & 3u) | ((q & 2u) << 1) | ((q & 1u) << 3) | (((p >> 5) & 1u) << 4) | (((p >> 6) & 1u) << 5) |
                              (((q >> 6) & 1u) << 6) | (((q >> 5) & 1u) << 7);
          buf[pos++] = (lane * 4u + j) | (r << 8) | ((unsigned)(z - zs) << 10) | ((unsigned)s_ntri[ci] << 15);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    n0 = n1;
    b0 = b1;
  }
  if (n_buf) flush();
  if (__ballot(tri_sum > 0)) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) tri_sum += __shfl_xor(tri_sum, s);
    if (lane == 0) atomicAdd(&counters[1], (unsigned long long)tri_sum);
  }
}

// element index of every active cell's base voxel (TSDF_COLOR_LAB: the voxels whose exact colour the mesh shows)
static __global__ void __launch_bounds__(256)
k_mc_cell_index(const McArgs a, const uint64_t *__restrict__ vals, uint64_t n, int64_t *__restrict__ idx) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t v = mc_unpack(vals[i]);
  const int x = (int)(v & 0xfffff), y = (int)((v >> 20) & 0xfffff), z = (int)((v >> 40) & 0xfffff);
  idx[i] = ((int64_t)(z - a.z_first) * a.ny + y) * a.pitch + x;
}

// The triangle count rides in the low bits of a cell word: the offsets are a scan straight over the sorted words.
struct McCountOf {
  __host__ __device__ uint32_t operator()(uint64_t word) const { return (uint32_t)(word & 15ull); }
};
using McCountIt = rocprim::transform_iterator<const uint64_t *, McCountOf, uint32_t>;

// Emit: a block takes 256 active cells (they are all valid: classify tested the eight corner weights) and its OUTPUT loop
// runs over the block's vertices, not over each thread's own triangles.  Phase 1 (one thread per cell): the eight corner
// values go to LDS (the triangle table indexes them dynamically; in registers that costs a select chain per access), with
// the case index, the colour and the cell key, and the thread notes for each of its 1-5 triangles which cell they belong
// to (s_map).  Phase 2 walks the block's 3 * (triangles of the block) vertices with consecutive threads on consecutive
// vertices -- every vertex interpolated on its own edge (interpolateEdge), a cell uses 3 to 15 of the twelve -- so every
// store instruction of a wave covers 768 CONTIGUOUS bytes of `verts`; the cell keys leave as contiguous 8-byte words,
// and the per-vertex colour bytes are composed right here: whole dwords over the block's byte range, byte stores for
// the at most three bytes on either end that a neighbouring block shares.
// (Rounds 1-5 ran the output loop per cell: 36-byte pieces at a stride of the neighbours' triangle counts in up to five
// divergent rounds, a per-triangle colour array and a second kernel to expand it -- 1.6 ms for 40 M triangles where this
// one takes 1.0, profiles/r06_mc_emit_ab_call28.txt.  Staging a wave's output run in LDS and copying it out was
// measured SLOWER than that in round 2, 3.1 vs 2.4 ms: the LDS round trip of the OUTPUT is what cost, not the inputs.)
static __global__ void __launch_bounds__(256)
k_mc_emit(const McArgs a, const uint64_t *__restrict__ vals, const uint32_t *__restrict__ offsets, uint64_t n_cells,
            float *__restrict__ verts, uint8_t *__restrict__ rgb_out, uint64_t *__restrict__ cell_out) {
  __shared__ signed char s_tri[256 * 16];
  __shared__ float s_leaf[256 * 9];   // 8 corner values per cell, stride 9
  __shared__ uint64_t s_key[256];     // x << 42 | y << 21 | z of the cell's base voxel
  __shared__ uint32_t s_col[256];     // r | g << 8 | b << 16
  __shared__ uint32_t s_first[256];   // the cell's first triangle, relative to the block's
  __shared__ uint8_t s_cube[256];
  __shared__ uint8_t s_map[256 * 5];  // triangle of the block -> cell of the block
  __shared__ uint32_t s_ntri;
  for (int i = threadIdx.x; i < 256 * 16; i += blockDim.x) s_tri[i] = mc_tri_table[i >> 4][i & 15];
  const uint64_t c0 = (uint64_t)blockIdx.x * blockDim.x, ci = c0 + threadIdx.x;
  const uint32_t T0 = offsets[c0];  // (block-uniform; c0 < n_cells by the grid's size)
  const int64_t sy = a.pitch, sz = (int64_t)a.ny * a.pitch;
  if (ci < n_cells) {
    const uint64_t word = vals[ci];
    const uint64_t v = mc_unpack(word);
    const int x = (int)(v & 0xfffff), y = (int)((v >> 20) & 0xfffff), z = (int)((v >> 40) & 0xfffff);
    // getGridValue (:91-106) of a valid corner: d * max_dist_neg, corners in pcl::MarchingCubes order
    const int64_t vi = ((int64_t)(z - a.z_first) * a.ny + y) * a.pitch + x;
    const int64_t off[8] = {0, 1, 1 + sz, sz, sy, 1 + sy, 1 + sy + sz, sy + sz};
    float leaf[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) leaf[k] = a.d[vi + off[k]] * a.neg;
    float *mine = s_leaf + threadIdx.x * 9;
#pragma unroll
    for (int k = 0; k < 8; ++k) mine[k] = leaf[k];
    s_cube[threadIdx.x] = (uint8_t)cube_index(leaf);
    uint32_t col = 0u;
    if (a.color_mode == 2) {  // :217-224 colour by confidence, evaluated in double like the reference
      const float std_dev = (float)((100. - (double)tsdf_load_w(a.pv, vi)) / 100.);
      const double r = (double)(1 - std_dev) * 255., b = (double)std_dev * 255.;
      const double rmin = (255. < r) ? 255. : r, bmin = (255. < b) ? 255. : b;  // std::min(x, 255.)
      col = (uint32_t)(unsigned char)((0. < rmin) ? rmin : 0.) |                 // std::max(0., x)
            ((uint32_t)(unsigned char)((0. < bmin) ? bmin : 0.) << 16);
    } else if (a.color_mode == 1 && a.pv.rgb) {  // :226-231
      col = tsdf_load_rgb(a.pv, vi);
    }
    s_col[threadIdx.x] = col;
    s_key[threadIdx.x] = ((uint64_t)x << 42) | ((uint64_t)y << 21) | (uint64_t)z;
    const uint32_t first = offsets[ci] - T0, cnt = (uint32_t)(word & 15ull);  // 1 .. 5 triangles
    s_first[threadIdx.x] = first;
    for (uint32_t k = 0; k < cnt; ++k) s_map[first + k] = (uint8_t)threadIdx.x;
    if (ci + 1 == n_cells || threadIdx.x == blockDim.x - 1) s_ntri = first + cnt;
  }
  __syncthreads();
  const uint32_t n_t = s_ntri, n_v = 3u * n_t;
  // edge e joins corners ea[e], eb[e]: {0,1,2,3,4,5,6,7,0,1,2,3} / {1,2,3,0,5,6,7,4,4,5,6,7}, packed 4 bits each
  const uint64_t EA = 0x321076543210ull, EB = 0x765447650321ull;
  struct __attribute__((packed, aligned(4))) F3 {
    float x, y, z;
  };
  F3 *vout = reinterpret_cast<F3 *>(verts) + 3ull * T0;
  for (uint32_t j = threadIdx.x; j < n_v; j += blockDim.x) {
    const uint32_t t = j / 3u, vtx = j - 3u * t;
    const uint32_t c = s_map[t];
    const uint64_t key = s_key[c];
    const int e = (int)s_tri[(uint32_t)s_cube[c] * 16u + 3u * (t - s_first[c]) + vtx];
    const int ka = (int)((EA >> (4 * e)) & 15u), kb = (int)((EB >> (4 * e)) & 15u);
    const float va = s_leaf[c * 9u + (uint32_t)ka], vb = s_leaf[c * 9u + (uint32_t)kb];
    // createSurface [PCL-recall]: centre = lower_boundary_ + size_voxel_ * index; corner k adds size_voxel_
    // in y if k&4, in z if k&2, in x if (k&1)^((k>>1)&1)
    const int idx[3] = {(int)(key >> 42), (int)((key >> 21) & 0x1fffff), (int)(key & 0x1fffff)};
    float center[3], far_[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      center[k] = a.lower[k] + a.size_voxel[k] * (float)idx[k];
      far_[k] = center[k] + a.size_voxel[k];
    }
    // interpolateEdge: mu = (iso - v1) / (v2 - v1); out = p1 + mu * (p2 - p1)
    const float mu = (0.f - va) / (vb - va);
    F3 p;
    {
      const float pa = ((ka & 1) ^ ((ka >> 1) & 1)) ? far_[0] : center[0], pb = ((kb & 1) ^ ((kb >> 1) & 1)) ? far_[0] : center[0];
      p.x = pa + mu * (pb - pa);
    }
    {
      const float pa = (ka & 4) ? far_[1] : center[1], pb = (kb & 4) ? far_[1] : center[1];
      p.y = pa + mu * (pb - pa);
    }
    {
      const float pa = (ka & 2) ? far_[2] : center[2], pb = (kb & 2) ? far_[2] : center[2];
      p.z = pa + mu * (pb - pa);
    }
    vout[j] = p;
  }
  if (cell_out)
    for (uint32_t t = threadIdx.x; t < n_t; t += blockDim.x) cell_out[(uint64_t)T0 + t] = s_key[s_map[t]];
  if (rgb_out) {  // bytes [9 T0, 9 (T0 + n_t)): byte b = channel (b % 9) % 3 of triangle b / 9 (:208-233)
    const uint64_t b0 = 9ull * T0, b1 = b0 + 9ull * n_t;
    auto byte_at = [&](uint64_t b) -> uint32_t {
      const uint32_t r = (uint32_t)(b - b0), t = r / 9u, ch = (r - 9u * t) % 3u;
      return (s_col[s_map[t]] >> (8u * ch)) & 255u;
    };
    const uint64_t up = (b0 + 3ull) & ~3ull, a0 = up < b1 ? up : b1, dn = b1 & ~3ull, a1 = dn > a0 ? dn : a0;  // whole dwords: [a0, a1)
    for (uint64_t w = a0 + 4ull * threadIdx.x; w < a1; w += 4ull * blockDim.x)
      *reinterpret_cast<uint32_t *>(rgb_out + w) = byte_at(w) | (byte_at(w + 1) << 8) | (byte_at(w + 2) << 16) | (byte_at(w + 3) << 24);
    if (threadIdx.x < 3u && b0 + threadIdx.x < a0) rgb_out[b0 + threadIdx.x] = (uint8_t)byte_at(b0 + threadIdx.x);
    if (threadIdx.x >= 4u && threadIdx.x < 7u && a1 + (threadIdx.x - 4u) < b1) rgb_out[a1 + (threadIdx.x - 4u)] = (uint8_t)byte_at(a1 + (threadIdx.x - 4u));
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
  if (h->multi) return tsdf_multi_march(h, w_min, color_mode, n_tri);
  TSDF_ENTER(h);
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
  a.check_w = !(h->packed && !(w_min > 0.f));  // a PACKED weight is min(k, max_weight) >= 0: never below w_min <= 0
  // Nor can it be below w_min <= min(1, max_weight) at a corner of a listed cell, while every owned plane holds only what
  // this handle's own integrate launches wrote since the reset (band_exact) and no halo plane is among the corners: a
  // listed cell has all eight |d| < 1 (:98), a distance leaves the reset value -1 only with an observation, and every
  // observation counts (octree.cpp:157-159: w + 1) -- so all eight counts are >= 1.  The 8 gathers per listed cell go.
  const bool counts_pass = h->packed && h->band_exact && tsdf_tuning().mc_skip && a.z_hi < h->z_end &&
                           w_min <= 1.f && w_min <= a.pv.wmax;
  if (counts_pass) a.check_w = 0;
  a.flush_at = std::min(MC_WAVE_BUF, std::max(0, tsdf_tuning().mc_flush_at));
  const int cell_rows = a.ny - 2;
  // a block = 4 waves = 4 * MC_R consecutive cell rows of one 256-voxel x-chunk, marching zb planes; the weight test
  // addresses the block's zb + 1 planes with 32-bit byte offsets
  {
    const uint64_t plane_bytes = (uint64_t)a.ny * (uint64_t)a.pitch * 4u;
    const int64_t fit = (int64_t)(0xffffffffull / plane_bytes) - 1;
    if (fit < 1) return TSDF_HIP_E_UNSUPPORTED;
    a.zb = (int)std::min<int64_t>(MC_ZB, fit);
  }
  // (row groups rounded up to a multiple of 8 for the kernel's XCD-aware re-reading of the block id; the padding
  // blocks lie past the grid and leave at once)
  const dim3 block(256), grid((unsigned)((a.qpr + 63) / 64), (unsigned)(((cell_rows + 4 * MC_R - 1) / (4 * MC_R) + 7) / 8 * 8),
                              (unsigned)((a.z_hi - a.z_lo + a.zb - 1) / a.zb));
  if (grid.y > 65535u || grid.z > 65535u) return TSDF_HIP_E_UNSUPPORTED;
  // What the "band seen" flags allow classify to skip (only while they describe the planes: tsdf_hip_volume::band_exact)
  NeedArgs need_args;
  size_t need_elems = 0, need_blocks = 0;
  a.need = a.need_blk = nullptr;
  a.need_gx = (int)grid.x;
  a.need_by = (int)grid.y;
  a.need_rows = 4 * (int)grid.y;
  if (h->band_exact && tsdf_tuning().mc_skip) {
    NeedArgs n;
    n.band = h->band;
    n.fx = h->band_fx, n.fy = h->band_fy;
    n.z_first = h->z_first, n.nz_alloc = h->nz_alloc;
    n.z_begin = h->z_begin, n.z_end = h->z_end;
    n.z_lo = a.z_lo, n.n_planes = a.z_hi - a.z_lo + 1;
    n.nx = a.nx, n.ny = a.ny;
    n.gx = a.need_gx, n.rows = a.need_rows, n.by = a.need_by;
    n.zb = a.zb;
    const size_t n_need = (size_t)n.gx * n.rows * n.n_planes, n_blk = (size_t)grid.x * grid.y * grid.z;
    if (n_need + n_blk > h->mc_need_cap) {
      TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
      if (h->mc_need) (void)hipFree(h->mc_need);
      h->mc_need = nullptr, h->mc_need_cap = 0;
      TSDF_HIP_TRY(hipMalloc(&h->mc_need, n_need + n_blk));
      h->mc_need_cap = n_need + n_blk;
    }
    a.need = h->mc_need;
    a.need_blk = h->mc_need + n_need;
    need_args = n;
    need_elems = n_need;
    need_blocks = n_blk;
  }
  unsigned long long counts[2 + MC_NEED_SLOTS] = {0};
  for (int i = 0; i < 4; ++i)
    if (!h->mc_ev[i]) TSDF_HIP_TRY(hipEventCreate(&h->mc_ev[i]));
  h->mc_ms[0] = h->mc_ms[1] = h->mc_ms[2] = 0.f;
  h->mc_ncells = 0;
  // pass 1 with the capacity we already have; if the surface turned out larger, grow and repeat
  for (int attempt = 0; attempt < 2; ++attempt) {
    const size_t cap = h->mc_cells_cap;
    TSDF_HIP_TRY(hipMemsetAsync(h->counter, 0, (attempt == 0 ? 2 + MC_NEED_SLOTS : 2) * sizeof(unsigned long long), h->stream));
    TSDF_HIP_TRY(hipEventRecord(h->mc_ev[0], h->stream));
    if (a.need && attempt == 0) {  // (inside the classify phase's timing)
      TSDF_HIP_TRY(hipMemsetAsync(const_cast<uint8_t *>(a.need_blk), 0, need_blocks, h->stream));
      if ((need_args.fx & 15) == 0 && need_args.fx <= 64) {
        const size_t row_planes = need_elems / (size_t)need_args.gx;
        hipLaunchKernelGGL(k_mc_need_rows, dim3((unsigned)((row_planes + 255) / 256)), dim3(256), 0, h->stream, need_args,
                           const_cast<uint8_t *>(a.need), const_cast<uint8_t *>(a.need_blk), h->counter + 2);
      } else
      hipLaunchKernelGGL(k_mc_need, dim3((unsigned)((need_elems + 255) / 256)), dim3(256), 0, h->stream, need_args,
                         const_cast<uint8_t *>(a.need), const_cast<uint8_t *>(a.need_blk), h->counter + 2);
      TSDF_HIP_TRY(hipGetLastError());
    }
    if (!h->packed)
      hipLaunchKernelGGL(k_mc_classify<0>, grid, block, 0, h->stream, a, h->mc_keys, (uint64_t)cap, h->counter);
    else if (h->rgb)
      hipLaunchKernelGGL(k_mc_classify<1>, grid, block, 0, h->stream, a, h->mc_keys, (uint64_t)cap, h->counter);
    else
      hipLaunchKernelGGL(k_mc_classify<2>, grid, block, 0, h->stream, a, h->mc_keys, (uint64_t)cap, h->counter);
    TSDF_HIP_TRY(hipGetLastError());
    TSDF_HIP_TRY(hipEventRecord(h->mc_ev[1], h->stream));
    TSDF_HIP_TRY(hipMemcpyAsync(counts, h->counter, sizeof counts, hipMemcpyDeviceToHost, h->stream));
    TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
    if (counts[0] <= cap) break;
    const size_t need = (size_t)counts[0] + (size_t)counts[0] / 8 + 1024;
    size_t c1 = h->mc_cells_cap;
    const int rc = ensure_buf(&h->mc_keys, &c1, need, h->stream);
    if (rc) return rc;
    h->mc_cells_cap = need;
  }
  const uint64_t n_cells = counts[0], ntri = counts[1];
  (void)hipEventElapsedTime(&h->mc_ms[0], h->mc_ev[0], h->mc_ev[1]);  // the last (successful) classify pass
  h->mc_ncells = n_cells;
  // distance bytes the classify pass requested: what k_mc_need allowed, or every plane of every block
  unsigned long long need_bytes = 0;
  for (int i = 0; i < MC_NEED_SLOTS; ++i) need_bytes += counts[2 + i];
  if (!a.need) {  // every wave reads its rows of every plane of its block: the same accounting as k_mc_need's
    uint64_t per_plane = 0;
    for (int wrow = 0; wrow < 4 * (int)grid.y; ++wrow) {
      const int yw = 1 + wrow * MC_R;
      if (1 + (wrow >> 2) * 4 * MC_R >= a.ny - 1) break;  // the block lies past the last cell row
      const uint64_t rows = (uint64_t)std::max(0, std::min((wrow & 3) == 3 ? MC_R + 1 : MC_R, a.ny - yw));
      for (int bx = 0; bx < (int)grid.x; ++bx) {
        for (int g = 0; g < 4; ++g)
          if (bx * 256 + g * 64 < a.nx) per_plane += 256u * rows;
        if (bx * 256 + 256 < a.nx) per_plane += 4u * rows;
      }
    }
    need_bytes = per_plane * (uint64_t)((a.z_hi - a.z_lo) + (int)grid.z);
  }
  // (bit 63 carries "the weight test could not fail and was not evaluated" to tsdf_hip_march_stats: the handle struct lives
  //  in tsdf_common.h, one of the sources whose hash stamps the committed k_integrate profiles -- bench.py kernel_sha16 --
  //  and a report-only flag is not worth invalidating them)
  h->mc_d_bytes = need_bytes | (counts_pass ? MC_STAT_COUNTS_PASS : 0ull);
  h->mc_skipped = a.need != nullptr;
  if (n_cells == 0) return TSDF_HIP_OK;
  if (n_cells > 0xffffffffull || ntri > 0xffffffffull) {
    tsdf_set_error("mesh too large (more than 2^32 cells or triangles)");
    return TSDF_HIP_E_UNSUPPORTED;
  }

  // sort the cell words by their Morton key -> reference triangle order; only the bits a coordinate can set take part (the
  // triangle count below MC_KEY_SHIFT rides along)
  int coord_bits = 1;
  while ((1 << coord_bits) < std::max(a.nx, std::max(a.ny, a.nz))) ++coord_bits;
  const unsigned key_bits = 3u * (unsigned)coord_bits;
  uint64_t *vals_out = nullptr;
  uint32_t *off = nullptr;
  size_t tmp_bytes_sort = 0, tmp_bytes_scan = 0;
  TSDF_HIP_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes_sort, h->mc_keys, vals_out, (size_t)n_cells, MC_KEY_SHIFT,
                                        MC_KEY_SHIFT + key_bits, h->stream));
  TSDF_HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes_scan, McCountIt(vals_out, McCountOf()), off, 0u, (size_t)n_cells,
                                       rocprim::plus<uint32_t>(), h->stream));
  const size_t al = 256;
  auto up = [&](size_t v) { return (v + al - 1) / al * al; };
  const size_t b_keys = up(n_cells * 8), b_cnt = up(n_cells * 4);
  const size_t total = b_keys + b_cnt + up(std::max(tmp_bytes_sort, tmp_bytes_scan));
  int rc = tsdf_ensure_scratch(h, total);
  if (rc) return rc;
  char *sp = (char *)h->scratch;
  vals_out = (uint64_t *)sp;  // the sorted cell words
  off = (uint32_t *)(sp + b_keys);
  void *tmp = sp + b_keys + b_cnt;
  TSDF_HIP_TRY(rocprim::radix_sort_keys(tmp, tmp_bytes_sort, h->mc_keys, vals_out, (size_t)n_cells, MC_KEY_SHIFT,
                                        MC_KEY_SHIFT + key_bits, h->stream));
  const unsigned cell_blocks = (unsigned)((n_cells + 255) / 256);
  TSDF_HIP_TRY(rocprim::exclusive_scan(tmp, tmp_bytes_scan, McCountIt(vals_out, McCountOf()), off, 0u, (size_t)n_cells,
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
  if (color_mode == 1 && h->lab_img) {
    // setColorByRGB on LABNode voxels (:226-231 -> LABNode::getRGB): the emit kernel reads the rgb plane; put the EXACT
    // bytes there for the cells it is about to read (host pow, tsdf_lab_exact_colors)
    int64_t *d_idx = nullptr;
    TSDF_HIP_TRY(hipMalloc(&d_idx, (size_t)n_cells * sizeof(int64_t)));
    hipLaunchKernelGGL(k_mc_cell_index, dim3(cell_blocks), dim3(256), 0, h->stream, a, vals_out, n_cells, d_idx);
    const int rcx = tsdf_lab_exact_colors(h, d_idx, (size_t)n_cells, nullptr, true);
    (void)hipFree(d_idx);
    if (rcx) return rcx;
  }
  TSDF_HIP_TRY(hipEventRecord(h->mc_ev[2], h->stream));
  hipLaunchKernelGGL(k_mc_emit, dim3(cell_blocks), dim3(256), 0, h->stream, a, vals_out, off, n_cells, h->mc_verts,
                     color_mode ? h->mc_rgb : nullptr, h->mc_cell);
  TSDF_HIP_TRY(hipGetLastError());
  TSDF_HIP_TRY(hipEventRecord(h->mc_ev[3], h->stream));
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  (void)hipEventElapsedTime(&h->mc_ms[1], h->mc_ev[1], h->mc_ev[2]);  // count read-back, sort, scan (+ buffer growth)
  (void)hipEventElapsedTime(&h->mc_ms[2], h->mc_ev[2], h->mc_ev[3]);
  h->mc_ntri = ntri;
  h->mc_has_rgb = color_mode != 0;
  if (n_tri) *n_tri = ntri;
  return TSDF_HIP_OK;
}

#ifdef TSDF_HIP_TEST_HOOKS
// Test / tuning hook: blocks of 256 threads the runtime admits per CU for the marching-cubes kernels
// (hipOccupancyMaxActiveBlocksPerMultiprocessor): out[0] k_mc_classify<1>, out[1] k_mc_emit.
extern "C" int tsdf_hip_selftest_occupancy_mc(int out[2]) {
  if (!out) return TSDF_HIP_E_INVALID;
  TSDF_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[0], k_mc_classify<1>, 256, 0));
  TSDF_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[1], k_mc_emit, 256, 0));
  return TSDF_HIP_OK;
}
#endif  // TSDF_HIP_TEST_HOOKS

// Report-only: device time of the last tsdf_hip_march by phase (HIP events on the handle's stream).
extern "C" int tsdf_hip_march_timing(tsdf_handle h, float ms[3], uint64_t *n_cells) {
  if (!h || !ms) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_march_timing(h, ms, n_cells);
  for (int i = 0; i < 3; ++i) ms[i] = h->mc_ms[i];
  if (n_cells) *n_cells = h->mc_ncells;
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_march_stats(tsdf_handle h, uint64_t out[4]) {
  if (!h || !out) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_march_stats(h, out);
  out[0] = h->mc_ncells;
  out[1] = h->mc_ntri;
  out[2] = h->mc_d_bytes & ~MC_STAT_COUNTS_PASS;
  out[3] = (h->mc_skipped ? 1u : 0u) | ((h->mc_d_bytes & MC_STAT_COUNTS_PASS) ? 2u : 0u);
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_march_fetch(tsdf_handle h, float *verts, uint8_t *rgb, uint64_t *cell) {
  if (!h) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_march_fetch(h, verts, rgb, cell);
  TSDF_ENTER(h);
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
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_march_fetch_device (the merged mesh of a multi-GPU set lives on the host)");
  TSDF_ENTER(h);
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
