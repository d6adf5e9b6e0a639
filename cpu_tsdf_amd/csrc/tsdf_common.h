// Internal definitions shared by the HIP translation units of libtsdf_hip.so.
// Not part of the public boundary (that is include/tsdf_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <string>
#include <vector>

#include "tsdf_hip.h"

struct tsdf_hip_pipeline;  // tsdf_integrate.hip: pinned staging ring of tsdf_hip_integrate_async
struct tsdf_hip_multi;     // tsdf_multi.hip: the Z-slab handles of a multi-GPU volume

struct tsdf_hip_volume {
  // non-null: this handle is a SET of Z-slab handles on several GPUs (tsdf_hip_create_multi) and owns no voxel plane
  // itself; every entry point forwards to tsdf_multi_* (tsdf_multi.hip)
  tsdf_hip_multi *multi = nullptr;
  tsdf_params p;
  int device = 0;
  int nx = 0, ny = 0, nz = 0;  // full grid resolution
  int z_begin = 0, z_end = 0;  // owned slab (global plane indices)
  int z_first = 0;             // global index of allocated plane 0 (= max(0, z_begin - halo))
  int nz_alloc = 0;            // allocated planes (slab + halos clipped to the grid)
  int64_t pitch = 0;           // floats per x row
  int levels[3] = {0, 0, 0};   // octree depth per axis (log2 res) or -1 if res is not a power of two
  // Voxel planes.  F32W layout: d, w (float) and, with colour, rgb (r | g<<8 | b<<16).
  // PACKED layout: w is not stored; the observation count k (w == min(k, max_weight), saturating at
  // kmax = ceil(max_weight)) lives in byte 3 of the colour word (rgb plane) or, without colour, in k8.
  float *d = nullptr, *w = nullptr;
  uint32_t *rgb = nullptr;
  uint8_t *k8 = nullptr;
  // TSDF_COLOR_RGB_NORMALIZED: running means r/i, g/i, b/i, i (RGBNormalized, octree.cpp:380-402); the rgb
  // plane then caches getRGB() of that state for every reader (queries, marching cubes, downloads)
  // TSDF_COLOR_LAB: cn[0..2] are the L, A, B means (LABNode, octree.cpp:531-551); lab_lut is the sRGB curve of
  // RGB2LAB tabulated by the host's libm, lab_img the frame's pixels converted once per frame (k_lab_image)
  float *cn[4] = {nullptr, nullptr, nullptr, nullptr};
  float *lab_lut = nullptr;
  float4 *lab_img = nullptr;
  // weight_by_variance_ (hpp:203-204): OctreeNode::M_ and nsample_ per voxel (octree.cpp:160-161), allocated when the
  // flag arrives (tsdf_hip_set_weighting; only a loaded .vol can carry it) on F32W / TSDF_COLOR_RGB volumes
  float *vm = nullptr;
  int32_t *vn = nullptr;
  int expf_fused_r = 0;  // which expf the host's libm runs (tsdf_integrate.hip tsdf_expf_glibc)
  // placement selection at create (tsdf_core.hip): probe sweep of each candidate allocation, which one was kept
  float alloc_probe_ms[8] = {-1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f, -1.f};
  int alloc_tried = 0, alloc_chosen = 0;
  int packed = 0;
  unsigned kmax = 0;
  // hpp:200-204: weightings only a loaded .vol can switch on (tsdf_hip_set_weighting).  weight_by_depth integrates
  // through the plain kernel (float weights); weight_by_variance makes integrate refuse (it needs M_ / nsample_).
  int weight_by_depth = 0, weight_by_variance = 0;
  float *ctr[3] = {nullptr, nullptr, nullptr};  // device centre tables (full axis length)
  std::vector<float> h_ctr[3];
  float *frame_depth = nullptr;  // staging for the host-pointer entry points
  uint32_t *frame_bgra = nullptr;  // = frame_depth + W*H (same allocation)
  // pinned two-slot bounce buffer every host<->device transfer of caller memory goes through (tsdf_to_host /
  // tsdf_to_device in tsdf_core.hip)
  char *bounce = nullptr;
  hipEvent_t bounce_ev[2] = {nullptr, nullptr};
  bool bounce_busy[2] = {false, false};
  unsigned bounce_turn = 0;  // slot of the next chunk: consecutive small transfers alternate instead of queueing on one slot
  double *cam64 = nullptr;         // fx, fy, cx, cy on the device
  tsdf_hip_pipeline *pipe = nullptr;
  int frame_staged = 0;            // tsdf_hip_organize left a frame in [frame_depth | frame_bgra]
  // "Band seen" flags, one byte per cell of 64 x 4 x 1 voxels (x, y, z) of the allocated planes: set by the integrate
  // kernels when a voxel of the cell is observed INSIDE the truncation band -- the only way a distance becomes negative,
  // and marching cubes only emits where one corner is negative, so k_mc_classify skips what no set flag is near.
  // band_exact: the flags describe the planes (true from reset while only the flag-keeping kernels have written them;
  // an upload, a plane copy or handing out raw pointers clears it, and classify then reads everything).
  uint8_t *band = nullptr;
  int band_fx = 0, band_fy = 0;
  bool band_exact = false;
  // Implied distances (k_integrate's s_bin): in a cell whose flag is 0 every observed voxel sits at the hinge value p of ALL
  // launches since the reset, and every other voxel at the reset value.  rest_state: 0 no flag-keeping launch yet, 1 all of
  // them so far were PACKED launches with hinge_fixed, kmax >= 1 and p == rest_bits, 2 one was not (until the next reset).
  int rest_state = 0;
  uint32_t rest_bits = 0;
  unsigned long long last_implied = 0, last_read_bytes = 0;  // tsdf_hip_last_read_detail
  bool last_implied_on = false;
  uint8_t *live = nullptr;         // brick-cull flags, one per k_integrate block
  size_t live_cap = 0;
  uint32_t *row_iv = nullptr;      // row intervals of a LIVE launch (k_rows, tsdf_integrate.hip), one word per voxel row
  size_t row_iv_cap = 0;
  bool ctr_increasing[3] = {false, false, false};  // the axis' centre table strictly increases (checked at create)
  unsigned long long *counter = nullptr;  // device scratch (n_observed etc.): 2048 slots
  unsigned long long last_observed = 0, last_changed_bytes = 0;  // tsdf_hip_last_count_detail
  bool ref_cull = false;   // tsdf_hip_set_reference_cull: replicate getFrustumCulledVoxels with these planes
  float cull_planes[24] = {0};
  int last_launch[4] = {0, 0, 0, 0};  // tsdf_hip_last_launch_info: ALLIN instance, fast projection, brick flags, blocks
  bool pair_pending = false;  // frame pairing: a committed frame sits uploaded in its ring slot, its launch waiting for a partner
  bool pair_fused = false;  // the last tsdf_integrate_launch2 went through k_integrate2 (else two launches)
  unsigned long long pair_first_observed = 0, pair_first_changed = 0, pair_first_implied = 0, pair_first_read = 0;  // ... of its first launch when it did not
  int count_slots = 0;     // counter slots the last counting launch filled (0 = none pending), tsdf_integrate_collect
  bool count_ran = false;  // that launch really ran (finite pose, something observable)
  hipStream_t stream = nullptr;
  // marching-cubes result buffers (owned, reused between calls)
  float *mc_verts = nullptr;
  uint8_t *mc_rgb = nullptr;
  uint64_t *mc_cell = nullptr;
  uint64_t mc_ntri = 0;
  size_t mc_cap = 0;       // triangles the output buffers hold
  bool mc_has_rgb = false;
  uint64_t *mc_keys = nullptr, *mc_vals = nullptr;  // active-cell list (Morton key, packed cell)
  size_t mc_cells_cap = 0;
  uint8_t *mc_need = nullptr;  // k_mc_need's per-wave-plane and per-block verdicts (tsdf_march.hip)
  size_t mc_need_cap = 0;
  hipEvent_t mc_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // tsdf_hip_march_timing
  float mc_ms[3] = {0.f, 0.f, 0.f};                            // classify, sort + scan, emit of the last call
  uint64_t mc_ncells = 0;
  uint64_t mc_d_bytes = 0;   // distance bytes the last classify pass requested (tsdf_hip_march_stats)
  bool mc_skipped = false;   // ... with the band flags deciding what to read
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
};

// Multi-GPU forwarding targets (tsdf_multi.hip); `h` is a handle with h->multi != nullptr.
void tsdf_multi_free(tsdf_hip_volume *v);
int tsdf_multi_reset(tsdf_handle h);
int tsdf_multi_synchronize(tsdf_handle h);
int tsdf_multi_set_weighting(tsdf_handle h, int by_depth, int by_variance);
int tsdf_multi_set_reference_cull(tsdf_handle h, const float planes[24]);
int tsdf_multi_variance_block(tsdf_handle h, bool down, int x0, int y0, int z0, int nx, int ny, int nz, float *M, int32_t *nsample);
int tsdf_multi_integrate(tsdf_handle h, const float *depth, const uint8_t *bgra, const float T[12], uint64_t *n_observed,
                         bool asynchronous);
int tsdf_multi_integrate_device(tsdf_handle h, const float *d_depth, const uint32_t *d_bgra, const float T[12],
                                uint64_t *n_observed);
int tsdf_multi_organize(tsdf_handle h, const float *xyz, size_t xyz_stride, const uint8_t *bgra, size_t bgra_stride, size_t n,
                        float cloud_units, int zero_nans, const double world_to_cam[12], float *depth_out, uint8_t *bgra_out,
                        uint64_t *n_valid);
int tsdf_multi_integrate_staged(tsdf_handle h, const float T[12], uint64_t *n_observed);
int tsdf_multi_last_count_detail(tsdf_handle h, uint64_t out[2]);
int tsdf_multi_last_read_detail(tsdf_handle h, uint64_t out[3]);
int tsdf_multi_block(tsdf_handle h, bool down, int x0, int y0, int z0, int nx, int ny, int nz, float *d, float *w, uint8_t *rgb);
int tsdf_multi_sample(tsdf_handle h, const float *xyz, size_t n, float *val, float *grad, float *hess, uint8_t *ok);
int tsdf_multi_lookup_rgb(tsdf_handle h, const float *xyz, size_t n, uint8_t *rgb, uint8_t *found);
int tsdf_multi_raycast(tsdf_handle h, const float rot[9], const float origin[3], int downsample, const double *inv, float *out);
int tsdf_multi_march(tsdf_handle h, float w_min, int color_mode, uint64_t *n_tri);
int tsdf_multi_march_fetch(tsdf_handle h, float *verts, uint8_t *rgb, uint64_t *cell);
int tsdf_multi_march_timing(tsdf_handle h, float ms[3], uint64_t *n_cells);
int tsdf_multi_march_stats(tsdf_handle h, uint64_t out[4]);
tsdf_handle tsdf_multi_first(tsdf_handle h);
// frame pairing on a set (round 6): the slabs pair the frames of their own rings; tsdf_multi_flush lets them launch what they hold
int tsdf_multi_flush(tsdf_handle h);
int tsdf_multi_set_frame_pairing(tsdf_handle h, int on);
int tsdf_multi_integrate_device2(tsdf_handle h, const float *da, const uint32_t *ca, const float TA[12], const float *planes_a, const float *db,
                                 const uint32_t *cb, const float TB[12], const float *planes_b, uint64_t *n_observed, int32_t *fused);
#define TSDF_NOT_ON_MULTI(h, what)                                                                          \
  do {                                                                                                      \
    if ((h) && (h)->multi) {                                                                                \
      tsdf_set_error(what " works on ONE slab handle; this handle is a multi-GPU set (tsdf_hip_create_multi)"); \
      return TSDF_HIP_E_UNSUPPORTED;                                                                        \
    }                                                                                                       \
  } while (0)

// integrateCloud in two halves (tsdf_integrate.hip): queue the launch; read the counting instance's counters later.
int tsdf_integrate_launch(tsdf_handle h, const float *d_depth, const uint32_t *d_bgra, const float T[12], bool count);
int tsdf_integrate_collect(tsdf_handle h, uint64_t *n_observed);
int tsdf_integrate_launch2(tsdf_handle h, const float *dA, const uint32_t *cA, const float TA[12], const float *planesA,
                           const float *dB, const uint32_t *cB, const float TB[12], const float *planesB, bool count, bool *fused);
int tsdf_integrate_collect2(tsdf_handle h, uint64_t n_observed[2]);

// Compact ray lists of the one-process multi-GPU renderView (tsdf_query.hip; all asynchronous on the slab's stream).
#define TSDF_MAX_SLABS 64
#define TSDF_RAY_FIN_INTS 9  // a finished ray on the wire: pixel index + 8 output floats
unsigned tsdf_ray_list_share(int64_t n_rays, int rank, int world);
int tsdf_ray_list_begin(tsdf_handle s, const float rot[9], const float origin[3], int downsample, int rank, int world,
                        int32_t *d_list, unsigned *count);
int tsdf_ray_list_advance(tsdf_handle s, const float rot[9], const float origin[3], int downsample, int rank, int world,
                          int32_t *d_list, unsigned count, unsigned *d_incomplete);
int tsdf_ray_list_route(tsdf_handle s, const int32_t *d_list, unsigned count, int n_slab, const int *z_end,
                        unsigned *d_counters, int32_t *d_outbox, int32_t *d_finbox);
int tsdf_ray_deliver(tsdf_handle s, const int32_t *d_fin, unsigned count, float *d_out, int64_t n_pix, const double *inv);

// Error plumbing -------------------------------------------------------------------------------
void tsdf_set_error(const std::string &msg);
int tsdf_hip_fail(hipError_t e, const char *what, const char *file, int line);

#define TSDF_HIP_TRY(expr)                                              \
  do {                                                                  \
    hipError_t _e = (expr);                                             \
    if (_e != hipSuccess) return tsdf_hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

// Every entry point works on its handle's device and leaves the caller's current device as it found it
// (a process may hold handles on several GPUs, or share the thread with another HIP user such as torch).
struct TsdfDeviceScope {
  int prev = -1;
  hipError_t err;
  explicit TsdfDeviceScope(int dev) {
    err = hipGetDevice(&prev);
    if (err != hipSuccess || prev == dev)
      prev = -1;  // nothing to restore
    else
      err = hipSetDevice(dev);
  }
  ~TsdfDeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  TsdfDeviceScope(const TsdfDeviceScope &) = delete;
  TsdfDeviceScope &operator=(const TsdfDeviceScope &) = delete;
};
#define TSDF_ON_DEVICE(dev)          \
  TsdfDeviceScope _device_scope(dev); \
  TSDF_HIP_TRY(_device_scope.err)
// Entry points that read or write the volume: on the handle's device, and after a frame that frame pairing is still
// holding back (tsdf_hip_set_frame_pairing: its kernel launch waits for a partner frame) has been launched on its own.
int tsdf_pipeline_flush(tsdf_hip_volume *v);
#define TSDF_ENTER(h)                                  \
  TSDF_ON_DEVICE((h)->device);                         \
  if ((h)->pair_pending) {                             \
    const int _rc_flush = tsdf_pipeline_flush(h);      \
    if (_rc_flush) return _rc_flush;                   \
  }

// Per-axis voxel-centre table (tsdf_core.hip): the octree's node-centre recurrence, or the closed form.
void tsdf_build_centers(int res, float size, std::vector<float> &out, int *levels);
// The size an axis' octree node centres are built from: OctreeNode keeps ONE size_, initialised from size_x
// (include/cpu_tsdf/octree.h:63-66), and split() offsets all three centre coordinates by size_ / 4
// (src/lib/octree.cpp:244-266) -- under a non-cubic setGridSize the reference's octree is still a cube of edge size_x,
// and integrateCloud updates THOSE leaves (pinned against the compiled reference, tests/test_oracle_golden.py).
// Grids without an octree equivalent (not a power of two on every axis, or not cubic in resolution) use the axis' own
// size with the closed-form centres.
// RGB2LAB's per-channel prefix (octree.cpp:441-458): float(v)/255., the sRGB curve through std::pow, times 100.
// A function of one byte, so the host's own libm tabulates it -- the same libm the reference would call here.
static inline void tsdf_lab_curve(float lut[256]) {
  for (int v = 0; v < 256; ++v) {
    float f = ((float)v / 255.);
    if (f > 0.0405)
      f = std::pow(((f + 0.055) / 1.055), 2.4);
    else
      f /= 12.92;
    f *= 100;
    lut[v] = f;
  }
}

// LAB2RGB (src/lib/octree.cpp:483-527) on the HOST: the cubes and the 1/2.4 powers go through std::pow there, so the bytes
// LABNode::getRGB shows are the host libm's -- the libm the reference would call on this machine -- by construction.
// Every result a caller can see (downloads, mesh colours, renderColoredView) is finished here from the voxels' float
// L, A, B state (tsdf_lab_exact_colors); the device's own lab_to_rgb only fills a cache nothing user-visible reads.
// static_cast<uint8_t>(float) is cvttss2si and the low byte on x86-64.
static inline uint8_t tsdf_u8_of_float(float v) {
  const int i = (v > -2147483904.f && v < 2147483648.f) ? (int)v : (int)0x80000000;
  return (uint8_t)(i & 255);
}
static inline uint32_t tsdf_lab2rgb_host(float L, float A, float B) {
  float Y = (L + 16) / 116.;
  float X = A / 500. + Y;
  float Z = Y - (B / 200.);
  if (std::pow((double)X, 3.0) > 0.008856) X = std::pow((double)X, 3.0); else X = (X - 16 / 116.) / 7.787;
  if (std::pow((double)Y, 3.0) > 0.008856) Y = std::pow((double)Y, 3.0); else Y = (Y - 16 / 116.) / 7.787;
  if (std::pow((double)Z, 3.0) > 0.008856) Z = std::pow((double)Z, 3.0); else Z = (Z - 16 / 116.) / 7.787;
  X *= 95.047;
  Y *= 100.;
  Z *= 108.883;
  X /= 100;
  Y /= 100;
  Z /= 100;
  float rf = X * +3.2406 + Y * -1.5372 + Z * -0.4986;
  float gf = X * -0.9689 + Y * +1.8758 + Z * +0.0415;
  float bf = X * +0.0557 + Y * -0.2040 + Z * +1.0570;
  if (rf > 0.0031308) rf = 1.055 * std::pow(static_cast<double>(rf), 1. / 2.4) - 0.055; else rf *= 12.92;
  if (gf > 0.0031308) gf = 1.055 * std::pow(static_cast<double>(gf), 1. / 2.4) - 0.055; else gf *= 12.92;
  if (bf > 0.0031308) bf = 1.055 * std::pow(static_cast<double>(bf), 1. / 2.4) - 0.055; else bf *= 12.92;
  return (uint32_t)tsdf_u8_of_float(rf * 255) | ((uint32_t)tsdf_u8_of_float(gf * 255) << 8) | ((uint32_t)tsdf_u8_of_float(bf * 255) << 16);
}
// n LAB triples (planar: L[n], A[n], B[n]) -> r | g << 8 | b << 16, on all host threads (tsdf_core.hip)
void tsdf_lab2rgb_host_many(const float *L, const float *A, const float *B, size_t n, uint32_t *out);
// TSDF_COLOR_LAB volumes: the exact getRGB() of n voxels given by their element indices (a DEVICE array; a negative index
// is skipped and yields 0).  host_rgb (n words, optional) receives them; write_plane also stores them into the rgb plane
// (what marching cubes' emit kernel reads).  Synchronises the stream.
int tsdf_lab_exact_colors(tsdf_hip_volume *v, const int64_t *d_idx, size_t n, uint32_t *host_rgb, bool write_plane);

static inline float tsdf_node_size(const tsdf_params &p, int axis) {
  const int r = p.res[0];
  const bool cubic_pow2 = r > 0 && (r & (r - 1)) == 0 && p.res[1] == r && p.res[2] == r;
  return cubic_pow2 ? p.size[0] : p.size[axis];
}
int tsdf_ensure_scratch(tsdf_hip_volume *v, size_t bytes);
// Copies between DEVICE memory and the CALLER's host memory, through the handle's pinned bounce buffer in chunks
// (two slots, the host memcpy of one chunk overlapping the DMA of the next).  tsdf_to_host returns with the
// data in `dst` (everything queued on the stream before it has completed); tsdf_to_device returns as soon as
// `src` has been consumed, the device copy being ordered on the handle's stream like any other work.
int tsdf_to_host(tsdf_hip_volume *v, void *dst, const void *dev_src, size_t bytes);
int tsdf_to_device(tsdf_hip_volume *v, void *dev_dst, const void *src, size_t bytes);
void tsdf_pipeline_destroy(tsdf_hip_volume *v);

// Launch-shape knobs, overridable from the environment for A/B runs (TSDF_HIP_ROWS_PER_BLOCK,
// TSDF_HIP_BLOCKS_PER_CU, TSDF_HIP_FAST_PROJECTION, TSDF_HIP_MC_FLUSH_AT, TSDF_HIP_CULL, TSDF_HIP_VOL_CHUNK, TSDF_HIP_ALLIN); read once, changeable
// through tsdf_hip_set_tuning.
struct TsdfTuning {
  int rows_per_block;  // voxel rows (of up to 1024 voxels) each integrate block walks
  int blocks_per_cu;   // grid-stride helper kernels: grid = 256 CUs x this
  int fast_projection; // certified fp32 pixel projection with exact fp64 fallback: 0 off, anything else on
  int mc_flush_at;     // marching-cubes classify: wave-private list flush threshold (tests lower it)
  int mc_skip;         // marching-cubes classify: skip what the band flags rule out (1)
  int cull;            // brick-level frustum cull in integrate: 1 when useful (default), 0 never, 2 always
  int vol_chunk;       // edge of the voxel blocks save / load stream through host memory
  int plain_kernel;    // F32W volumes integrate through the plain per-voxel kernel (the weight_by_depth one, w_new = 1)
  int alloc_tries;     // tsdf_hip_create: placements of a large volume's planes to probe before keeping the fastest
  int allin;           // integrate: use the ALLIN kernel instance when the whole slab is provably in range and in the image (1)
  int refcull_plain;   // reference-cull replication through the plain per-voxel kernel instead of the row intervals (tests: 0)
  int live_log2tx;     // LIVE launches of a partly visible slab: log2 of the quads per block row (5: 128 voxels x 8 rows per block pass; Scene B at 2048^3: 0.37 ms against 0.60 at 6)
  int zfast;           // integrate launches hand out blocks planes-fastest: 1 always (default: 16.26 against 16.60 ms at 2048^3 + colour, 15.5 against 16.9 ms on a 4096 x 4096 x 512 slab with 1280x960 frames), 0 never, -1 only when the frame outgrows an XCD's L2
  int fuse2;           // tsdf_hip_integrate_device2 / frame pairing: 1 = one sweep per pair where that is the faster way (with colour; without, where k_integrate_p does not apply), 2 = wherever both poses qualify, 0 = never
  int implied_d;       // PACKED integrate launches do not read distance words the "band seen" flags and the counts determine (1)
  int pipe;            // ALLIN PACKED launches run the software-pipelined row loop: bit 0 without colour (k_integrate_p: on), bit 1 with (k_integrate_pc: measured no faster than k_integrate's own loop, off); 1
};
const TsdfTuning &tsdf_tuning();
// Edge of the voxel blocks save / load stream through host memory: TSDF_HIP_VOL_CHUNK, read at EVERY call (an I/O path: a
// getenv costs nothing there), so that a caller -- the tests, through the C++ drop-in as well -- can change it at run time
// without an entry point for it; else the tuning value.
int tsdf_vol_chunk();

// Read-only view of the voxel planes for the gather kernels (raycast, sample, marching cubes, transfers).
struct PlaneView {
  const float *d, *w;
  const uint32_t *rgb;
  const uint8_t *k8;
  float wmax;
  unsigned kmax;
  int packed;
};

static inline PlaneView tsdf_plane_view(const tsdf_hip_volume *v) {
  return PlaneView{v->d, v->w, v->rgb, v->k8, v->p.max_weight, v->kmax, v->packed};
}

// w after k observations: each addObservation does w += 1; if (w > max_weight) w = max_weight
// (octree.cpp:156-158 with w_new = 1), so w == min(k, max_weight) for every k while 0 <= max_weight.
static __device__ __forceinline__ float tsdf_decode_w(unsigned k, float wmax) {
  const float kf = (float)k;
  return kf > wmax ? wmax : kf;
}

// Inverse of tsdf_decode_w; false if no count represents w (then the volume needs the F32W layout).
static __device__ __forceinline__ bool tsdf_encode_w(float w, float wmax, unsigned kmax, unsigned &k) {
  if (w == wmax) {
    k = kmax;
    return true;
  }
  if (w >= 0.f && w < (float)kmax && w == floorf(w)) {
    k = (unsigned)w;
    return true;
  }
  k = 0;
  return false;
}

static __device__ __forceinline__ float tsdf_load_w(const PlaneView &v, int64_t i) {
  if (!v.packed) return v.w[i];
  const unsigned k = v.rgb ? (v.rgb[i] >> 24) : (unsigned)v.k8[i];
  return tsdf_decode_w(k, v.wmax);
}

static __device__ __forceinline__ uint32_t tsdf_load_rgb(const PlaneView &v, int64_t i) {
  return v.rgb[i] & 0xffffffu;
}

// Volume element index of (x, y, z_global); the plane must be allocated.
static inline __host__ __device__ int64_t tsdf_index(int64_t pitch, int ny, int z_first, int x, int y,
                                                      int zg) {
  return ((int64_t)(zg - z_first) * ny + y) * pitch + x;
}
