// libtsdf_hip.so -- volume lifetime, voxel-centre tables, raw block transfer.
// gfx950 only.  Boundary: include/tsdf_hip.h.
#include <math.h>

#include <string>
#include <stdlib.h>
#include <stdio.h>

#include <algorithm>
#include <thread>
#include <string.h>

#include <mutex>

#include "tsdf_common.h"
#ifdef TSDF_HIP_TEST_HOOKS
#include "tsdf_hip_test.h"
#endif

// ---------------------------------------------------------------------------------------------
// errors
static thread_local std::string g_last_error;

void tsdf_set_error(const std::string &msg) { g_last_error = msg; }

int tsdf_hip_fail(hipError_t e, const char *what, const char *file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
  g_last_error = buf;
  (void)hipGetLastError();
  return e == hipErrorOutOfMemory ? TSDF_HIP_E_NOMEM : TSDF_HIP_E_HIP;
}

extern "C" const char *tsdf_hip_last_error(void) { return g_last_error.c_str(); }

extern "C" const char *tsdf_hip_error_string(int code) {
  switch (code) {
    case TSDF_HIP_OK: return "ok";
    case TSDF_HIP_E_INVALID: return "invalid argument";
    case TSDF_HIP_E_NOMEM: return "out of device memory";
    case TSDF_HIP_E_HIP: return "HIP runtime error";
    case TSDF_HIP_E_NODEVICE: return "no HIP device";
    case TSDF_HIP_E_UNSUPPORTED: return "unsupported";
    case TSDF_HIP_E_IO: return "file error";
  }
  return "unknown";
}

static int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

static TsdfTuning &tuning_storage() {
  static TsdfTuning t = {std::max(1, env_int("TSDF_HIP_ROWS_PER_BLOCK", 64)),
                         std::max(1, env_int("TSDF_HIP_BLOCKS_PER_CU", 8)),
                         env_int("TSDF_HIP_FAST_PROJECTION", -1), env_int("TSDF_HIP_MC_FLUSH_AT", 512), env_int("TSDF_HIP_MC_SKIP", 1),
                         env_int("TSDF_HIP_CULL", 1), std::max(1, env_int("TSDF_HIP_VOL_CHUNK", 256)),
                         env_int("TSDF_HIP_PLAIN_KERNEL", 0), env_int("TSDF_HIP_ALLOC_TRIES", 3), env_int("TSDF_HIP_ALLIN", 1),
                         env_int("TSDF_HIP_REFCULL_PLAIN", 0), env_int("TSDF_HIP_LIVE_LOG2TX", 5), env_int("TSDF_HIP_ZFAST", 1), env_int("TSDF_HIP_FUSE2", 1), env_int("TSDF_HIP_IMPLIED_D", 1),
                         env_int("TSDF_HIP_PIPE", 1)};
  return t;
}

const TsdfTuning &tsdf_tuning() { return tuning_storage(); }

int tsdf_vol_chunk() {
  const int live = env_int("TSDF_HIP_VOL_CHUNK", 0);
  return live > 0 ? live : tsdf_tuning().vol_chunk;
}

#ifdef TSDF_HIP_TEST_HOOKS
// Test / A-B hook: change a launch-shape knob at run time (same names as the TSDF_HIP_* variables, lower
// case, without the prefix).  None of them may change results; the tests use this to prove it.
extern "C" int tsdf_hip_set_tuning(const char *name, int value) {
  if (!name) return TSDF_HIP_E_INVALID;
  TsdfTuning &t = tuning_storage();
  const std::string n(name);
  if (n == "rows_per_block")
    t.rows_per_block = std::max(1, value);
  else if (n == "blocks_per_cu")
    t.blocks_per_cu = std::max(1, value);
  else if (n == "fast_projection")
    t.fast_projection = value;
  else if (n == "mc_flush_at")
    t.mc_flush_at = value;
  else if (n == "mc_skip")
    t.mc_skip = value;
  else if (n == "cull")
    t.cull = value;
  else if (n == "vol_chunk")
    t.vol_chunk = std::max(1, value);
  else if (n == "plain_kernel")
    t.plain_kernel = value;
  else if (n == "alloc_tries")
    t.alloc_tries = std::max(1, value);
  else if (n == "allin")
    t.allin = value;
  else if (n == "refcull_plain")
    t.refcull_plain = value;
  else if (n == "fuse2")
    t.fuse2 = value;
  else if (n == "implied_d")
    t.implied_d = value;
  else if (n == "pipe")
    t.pipe = value;
  else if (n == "live_log2tx")
    t.live_log2tx = value;
  else if (n == "zfast")
    t.zfast = value;
  else
    return TSDF_HIP_E_INVALID;
  return TSDF_HIP_OK;
}
#endif  // TSDF_HIP_TEST_HOOKS

extern "C" int tsdf_hip_abi_version(void) { return TSDF_HIP_ABI_VERSION; }

// Pinned host memory for callers that have no HIP of their own (the C++ shell, ctypes): buffers handed to the
// transfer entry points (raycast, sample, march_fetch, download, integrate, ...) from such memory are DMA targets /
// sources directly.
extern "C" int tsdf_hip_host_alloc(size_t bytes, void **out) {
  if (!out || !bytes) return TSDF_HIP_E_INVALID;
  *out = nullptr;
  TSDF_HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocPortable));
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_host_free(void *p) {
  if (!p) return TSDF_HIP_OK;
  TSDF_HIP_TRY(hipHostFree(p));
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// ---------------------------------------------------------------------------------------------
// reference constructor defaults -- src/lib/tsdf_volume_octree.cpp:54-85
extern "C" void tsdf_hip_default_params(tsdf_params *p) {
  memset(p, 0, sizeof *p);
  p->res[0] = p->res[1] = p->res[2] = 512;
  p->size[0] = p->size[1] = p->size[2] = 3.0f;
  p->max_dist_pos = 0.03f;
  p->max_dist_neg = 0.03f;
  p->max_weight = 100.f;
  p->min_sensor_dist = 0.3f;
  p->max_sensor_dist = 3.0f;
  p->fx = p->fy = 525.;
  p->cx = 320;
  p->cy = 240;
  p->image_width = 640;
  p->image_height = 480;
  p->integrate_color = 0;
  p->xform_order = TSDF_XFORM_PCL_SSE;
  p->z_begin = p->z_end = 0;
  p->halo = 0;
  p->device = -1;
  p->layout = TSDF_LAYOUT_AUTO;
  p->color_mode = TSDF_COLOR_RGB;
}

// ---------------------------------------------------------------------------------------------
// Voxel centres.
//
// updateVoxel reads the *octree node* centre (hpp:143-144), which the reference builds by repeated
// halving: child centre = parent centre -/+ size/4, child size = size/2, all in float
// (src/lib/octree.cpp:244-266), starting from the root at 0 with size_ = size_x
// (octree.cpp:589-590, octree.h:63-66).  For a power-of-two resolution we replay exactly that
// arithmetic per axis, so the table is bit-identical to the leaf centres for any grid size, dyadic
// or not.  For other resolutions there is no octree equivalent (the reference CLI forces a power of
// two, src/prog/integrate.cpp:486-494) and we fall back to getVoxelCenter's closed form
// (tsdf_volume_octree.cpp:553-560).
static int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

void tsdf_build_centers(int res, float size, std::vector<float> &out, int *levels) {
  out.resize(res);
  const int L = ilog2_exact(res);
  *levels = L;
  if (L >= 0) {
    for (int i = 0; i < res; ++i) {
      float c = 0.f;
      float s = size;
      for (int l = L - 1; l >= 0; --l) {
        const float off = s / 4;
        c = ((i >> l) & 1) ? c + off : c - off;
        s = s / 2;
      }
      out[i] = c;
    }
  } else {
    const float off = size / 2.0;
    for (int i = 0; i < res; ++i) out[i] = (float)((i + 0.5) * size / (double)res - off);
  }
}

// ---------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
k_fill_u32(uint32_t *__restrict__ p, uint32_t value, int64_t n4 /* number of uint4 */) {
  const uint4 v4 = make_uint4(value, value, value, value);
  uint4 *q = reinterpret_cast<uint4 *>(p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x)
    q[i] = v4;
}

static int fill_u32(tsdf_hip_volume *v, void *p, uint32_t value, int64_t n) {
  // n is a multiple of 4 by construction (pitch % 4 == 0)
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)blocks), dim3(256), 0, v->stream, (uint32_t *)p, value, n4);
  TSDF_HIP_TRY(hipGetLastError());
  return TSDF_HIP_OK;
}

int tsdf_ensure_scratch(tsdf_hip_volume *v, size_t bytes) {
  if (bytes <= v->scratch_bytes) return TSDF_HIP_OK;
  if (v->scratch) {
    TSDF_HIP_TRY(hipStreamSynchronize(v->stream));
    TSDF_HIP_TRY(hipFree(v->scratch));
    v->scratch = nullptr;
    v->scratch_bytes = 0;
  }
  TSDF_HIP_TRY(hipMalloc(&v->scratch, bytes));
  v->scratch_bytes = bytes;
  return TSDF_HIP_OK;
}

// ---- exact LAB colours (host finish) ------------------------------------------------------------------------------------
void tsdf_lab2rgb_host_many(const float *L, const float *A, const float *B, size_t n, uint32_t *out) {
  const unsigned hw = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
  const unsigned nt = n < 4096 ? 1u : hw;
  auto work = [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) out[i] = tsdf_lab2rgb_host(L[i], A[i], B[i]);
  };
  if (nt == 1) return work(0, n);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
  for (auto &t : th) t.join();
}

static __global__ void __launch_bounds__(256)
k_lab_gather(const int64_t *__restrict__ idx, size_t n, const float *__restrict__ L, const float *__restrict__ A,
             const float *__restrict__ B, float *__restrict__ out) {  // out: planar L[n] A[n] B[n]
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = idx[i];
  out[i] = v >= 0 ? L[v] : 0.f;
  out[n + i] = v >= 0 ? A[v] : 0.f;
  out[2 * n + i] = v >= 0 ? B[v] : 0.f;
}

static __global__ void __launch_bounds__(256)
k_rgb_scatter(const int64_t *__restrict__ idx, size_t n, const uint32_t *__restrict__ words, uint32_t *__restrict__ plane) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = idx[i];
  if (v >= 0) plane[v] = (plane[v] & 0xff000000u) | (words[i] & 0xffffffu);  // several entries may name one voxel: same value
}

int tsdf_lab_exact_colors(tsdf_hip_volume *v, const int64_t *d_idx, size_t n, uint32_t *host_rgb, bool write_plane) {
  if (!n) return TSDF_HIP_OK;
  if (!v->lab_img || !v->cn[0] || !v->cn[1] || !v->cn[2]) return TSDF_HIP_E_INVALID;
  const size_t chunk = (size_t)16 << 20;  // voxels per round trip (192 MB of floats)
  float *d_lab = nullptr;
  uint32_t *d_words = nullptr;
  const size_t m = std::min(n, chunk);
  TSDF_HIP_TRY(hipMalloc(&d_lab, m * 3 * sizeof(float)));
  if (write_plane && hipMalloc(&d_words, m * sizeof(uint32_t)) != hipSuccess) {
    (void)hipFree(d_lab);
    return tsdf_hip_fail(hipErrorOutOfMemory, "hipMalloc", __FILE__, __LINE__);
  }
  std::vector<float> lab(m * 3);
  std::vector<uint32_t> words(m);
  int rc = TSDF_HIP_OK;
  for (size_t off = 0; off < n && !rc; off += chunk) {
    const size_t c = std::min(chunk, n - off);
    hipLaunchKernelGGL(k_lab_gather, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, v->stream, d_idx + off, c, v->cn[0], v->cn[1],
                       v->cn[2], d_lab);
    if ((rc = tsdf_to_host(v, lab.data(), d_lab, c * 3 * sizeof(float)))) break;
    tsdf_lab2rgb_host_many(lab.data(), lab.data() + c, lab.data() + 2 * c, c, words.data());
    if (host_rgb) memcpy(host_rgb + off, words.data(), c * sizeof(uint32_t));
    if (write_plane) {
      if ((rc = tsdf_to_device(v, d_words, words.data(), c * sizeof(uint32_t)))) break;
      hipLaunchKernelGGL(k_rgb_scatter, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, v->stream, d_idx + off, c, d_words, v->rgb);
      if (hipStreamSynchronize(v->stream) != hipSuccess) rc = TSDF_HIP_E_HIP;
    }
  }
  (void)hipFree(d_lab);
  if (d_words) (void)hipFree(d_words);
  return rc;
}

// Host <-> device transfers of caller memory.  A hipMemcpy straight from / into pageable memory makes the
// runtime pin and unpin the caller's pages around every call; with a caller that hands in a new buffer each
// time (numpy does) that bookkeeping surfaces as multi-millisecond stalls in LATER calls on the stream
// (measured: every other integrateCloud taking 22 ms instead of 2.2 ms after a renderView into a fresh array).
// Going through pinned memory costs one host memcpy (~10 GB/s, overlapped with the DMA chunk by chunk) and is
// the same every time.
static const size_t kBounceChunk = 2u << 20;

static int bounce_ready(tsdf_hip_volume *v) {
  if (!v->bounce) {
    TSDF_HIP_TRY(hipHostMalloc((void **)&v->bounce, 2 * kBounceChunk, hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) TSDF_HIP_TRY(hipEventCreateWithFlags(&v->bounce_ev[i], hipEventDisableTiming));
  }
  return TSDF_HIP_OK;
}

static int bounce_wait(tsdf_hip_volume *v, int slot) {
  if (v->bounce_busy[slot]) {
    TSDF_HIP_TRY(hipEventSynchronize(v->bounce_ev[slot]));
    v->bounce_busy[slot] = false;
  }
  return TSDF_HIP_OK;
}

// Is `p` host memory the runtime already knows as pinned (hipHostMalloc / hipHostRegister / tsdf_hip_host_alloc)?
// Then the DMA engine can write it directly and the bounce buffer -- one host memcpy per transfer -- is skipped.
static bool is_pinned_host(const void *p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // an ordinary pageable pointer: not an error
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

int tsdf_to_host(tsdf_hip_volume *v, void *dst, const void *dev_src, size_t bytes) {
  if (!bytes) return TSDF_HIP_OK;
  if (bytes >= (64u << 10) && is_pinned_host(dst)) {
    TSDF_HIP_TRY(hipMemcpyAsync(dst, dev_src, bytes, hipMemcpyDeviceToHost, v->stream));
    TSDF_HIP_TRY(hipStreamSynchronize(v->stream));
    return TSDF_HIP_OK;
  }
  int rc = bounce_ready(v);
  if (rc) return rc;
  const size_t chunks = (bytes + kBounceChunk - 1) / kBounceChunk;
  const unsigned turn = v->bounce_turn;
  v->bounce_turn += (unsigned)chunks;
  for (size_t k = 0; k <= chunks; ++k) {
    if (k < chunks) {  // queue chunk k into its slot (free: chunk k-2 left it in the previous iteration)
      const int slot = (int)((turn + k) & 1);
      if ((rc = bounce_wait(v, slot))) return rc;
      const size_t off = k * kBounceChunk, len = std::min(kBounceChunk, bytes - off);
      TSDF_HIP_TRY(hipMemcpyAsync(v->bounce + slot * kBounceChunk, (const char *)dev_src + off, len, hipMemcpyDeviceToHost,
                                  v->stream));
      TSDF_HIP_TRY(hipEventRecord(v->bounce_ev[slot], v->stream));
      v->bounce_busy[slot] = true;
    }
    if (k > 0) {  // hand chunk k-1 to the caller while chunk k is in flight
      const int slot = (int)((turn + k - 1) & 1);
      if ((rc = bounce_wait(v, slot))) return rc;
      const size_t off = (k - 1) * kBounceChunk, len = std::min(kBounceChunk, bytes - off);
      memcpy((char *)dst + off, v->bounce + slot * kBounceChunk, len);
    }
  }
  return TSDF_HIP_OK;
}

int tsdf_to_device(tsdf_hip_volume *v, void *dev_dst, const void *src, size_t bytes) {
  if (!bytes) return TSDF_HIP_OK;
  if (bytes >= (64u << 10) && is_pinned_host(src)) {
    // (the caller may reuse `src` when the call returns, as with the bounce path: wait for the copy)
    TSDF_HIP_TRY(hipMemcpyAsync(dev_dst, src, bytes, hipMemcpyHostToDevice, v->stream));
    TSDF_HIP_TRY(hipStreamSynchronize(v->stream));
    return TSDF_HIP_OK;
  }
  int rc = bounce_ready(v);
  if (rc) return rc;
  const size_t chunks = (bytes + kBounceChunk - 1) / kBounceChunk;
  const unsigned turn = v->bounce_turn;
  v->bounce_turn += (unsigned)chunks;
  for (size_t k = 0; k < chunks; ++k) {
    const int slot = (int)((turn + k) & 1);
    if ((rc = bounce_wait(v, slot))) return rc;  // the copy that last read this slot has finished
    const size_t off = k * kBounceChunk, len = std::min(kBounceChunk, bytes - off);
    memcpy(v->bounce + slot * kBounceChunk, (const char *)src + off, len);
    TSDF_HIP_TRY(hipMemcpyAsync((char *)dev_dst + off, v->bounce + slot * kBounceChunk, len, hipMemcpyHostToDevice, v->stream));
    TSDF_HIP_TRY(hipEventRecord(v->bounce_ev[slot], v->stream));
    v->bounce_busy[slot] = true;
  }
  return TSDF_HIP_OK;
}

static void free_volume(tsdf_hip_volume *v) {
  if (!v) return;
  if (v->multi) tsdf_multi_free(v);
  TsdfDeviceScope scope(v->device);
  tsdf_pipeline_destroy(v);
  if (v->d) (void)hipFree(v->d);
  if (v->w) (void)hipFree(v->w);
  if (v->rgb) (void)hipFree(v->rgb);
  if (v->k8) (void)hipFree(v->k8);
  for (int c = 0; c < 4; ++c)
    if (v->cn[c]) (void)hipFree(v->cn[c]);
  if (v->vm) (void)hipFree(v->vm);
  if (v->vn) (void)hipFree(v->vn);
  if (v->lab_lut) (void)hipFree(v->lab_lut);
  if (v->lab_img) (void)hipFree(v->lab_img);
  for (int a = 0; a < 3; ++a)
    if (v->ctr[a]) (void)hipFree(v->ctr[a]);
  if (v->frame_depth) (void)hipFree(v->frame_depth);
  if (v->bounce) (void)hipHostFree(v->bounce);
  for (int i = 0; i < 2; ++i)
    if (v->bounce_ev[i]) (void)hipEventDestroy(v->bounce_ev[i]);
  if (v->cam64) (void)hipFree(v->cam64);
  if (v->live) (void)hipFree(v->live);
  if (v->row_iv) (void)hipFree(v->row_iv);
  if (v->band) (void)hipFree(v->band);
  if (v->counter) (void)hipFree(v->counter);
  if (v->mc_verts) (void)hipFree(v->mc_verts);
  if (v->mc_rgb) (void)hipFree(v->mc_rgb);
  if (v->mc_cell) (void)hipFree(v->mc_cell);
  if (v->mc_keys) (void)hipFree(v->mc_keys);
  if (v->mc_vals) (void)hipFree(v->mc_vals);
  if (v->mc_need) (void)hipFree(v->mc_need);
  for (int i = 0; i < 4; ++i)
    if (v->mc_ev[i]) (void)hipEventDestroy(v->mc_ev[i]);
  if (v->scratch) (void)hipFree(v->scratch);
  delete v;
}

// ---- placement of the voxel planes -----------------------------------------------------------------------------------
// On MI355X the physical pages behind an allocation decide how well a streaming read-modify-write of it runs: the same
// 2048^3 volume integrates in 17.9 ms or in 18.4-19.0 ms depending on the ALLOCATION (not on the process: destroying and
// re-creating the volume inside one process changes it; tools/bimodal_probe.py), and a plain sweep of the planes shows
// the same split (26.1-26.5 ms against 27.3-27.7 ms).  So tsdf_hip_create allocates a large volume's planes up to
// `alloc_tries` times (tuning knob, default 3; two candidates are alive at once, so only while twice the volume fits),
// sweeps each candidate once with the probe below and keeps the fastest placement.  Costs a few tenths of a second at
// create; every later integrateCloud runs in the fast mode with probability 1 - 0.4^3 instead of 0.6.
struct PlaneSet {
  float *d = nullptr, *w = nullptr;
  uint32_t *rgb = nullptr;
  uint8_t *k8 = nullptr;
};

static __global__ void __launch_bounds__(256)
k_probe_rmw(uint4 *__restrict__ a, uint4 *__restrict__ b, uint4 *__restrict__ c, int64_t n4, unsigned zero) {
  // reads every word of the planes and writes it back unchanged (`zero` is 0, but only at run time), non-temporal on
  // both sides like k_integrate's voxel stream (tsdf_buffer.h)
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u *pa = reinterpret_cast<v4u *>(a), *pb = reinterpret_cast<v4u *>(b), *pc = reinterpret_cast<v4u *>(c);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    v4u x = __builtin_nontemporal_load(pa + i);
    x ^= zero;  // (every component: an untouched one would let the compiler drop its load and store)
    __builtin_nontemporal_store(x, pa + i);
    if (b) {
      v4u y = __builtin_nontemporal_load(pb + i);
      y ^= zero;
      __builtin_nontemporal_store(y, pb + i);
    }
    if (c) {
      v4u z = __builtin_nontemporal_load(pc + i);
      z ^= zero;
      __builtin_nontemporal_store(z, pc + i);
    }
  }
}

static void release_planes(PlaneSet &s) {
  if (s.d) (void)hipFree(s.d);
  if (s.w) (void)hipFree(s.w);
  if (s.rgb) (void)hipFree(s.rgb);
  if (s.k8) (void)hipFree(s.k8);
  s = PlaneSet();
}

// milliseconds of one probe sweep over the candidate (second of two sweeps), or a negative number on any error
static float probe_planes(const PlaneSet &s, int64_t n, hipStream_t stream) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
  float ms = -1.f;
  const int64_t n4 = n / 4;
  const unsigned grid = 256u * (unsigned)tsdf_tuning().blocks_per_cu;
  bool ok = true;
  for (int pass = 0; pass < 2 && ok; ++pass) {
    ok &= hipEventRecord(e0, stream) == hipSuccess;
    hipLaunchKernelGGL(k_probe_rmw, dim3(grid), dim3(256), 0, stream, (uint4 *)s.d, (uint4 *)s.w, (uint4 *)s.rgb, n4, 0u);
    ok &= hipGetLastError() == hipSuccess;
    ok &= hipEventRecord(e1, stream) == hipSuccess;
    ok &= hipEventSynchronize(e1) == hipSuccess;
    if (ok) ok &= hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ok ? ms : -1.f;
}

// integrateCloud in the reference first drops octree cells with pcl::FrustumCulling (getFrustumCulledVoxels,
// tsdf_volume_octree.cpp:619-652): a pyramid of 1.1 x the angle 2 atan(W/2 / fx) (and likewise vertically) around the
// optical AXIS, between min_ and max_sensor_dist_ -- the principal point does not enter.  For an ordinary camera that
// pyramid contains every ray of the image, the cull removes only voxels updateVoxel would reject anyway, and the dense
// grid (which has no cull that could change results) agrees with the reference voxel for voxel.  It does NOT contain
// the image when the principal point sits more than ~10 % of the half-width off centre, or when max_sensor_dist is not
// a finite, moderate number (the frustum's corners become inf/NaN and the reference integrates nothing): then the
// reference skips voxels that project into the image and this library integrates them.  This function says which
// regime a parameter set is in (1 = the reference's cull is a no-op, results are identical).
extern "C" int tsdf_hip_reference_cull_is_noop(const tsdf_params *p) {
  if (!p || !(p->fx > 0.0) || !(p->fy > 0.0) || p->image_width <= 0 || p->image_height <= 0) return 0;
  if (!(p->max_sensor_dist > 0.f) || !(p->max_sensor_dist < 1e15f)) return 0;
  const double W = p->image_width, H = p->image_height;
  const double th = tan(1.1 * atan(0.5 * W / p->fx)) * (1.0 - 1e-5), tv = tan(1.1 * atan(0.5 * H / p->fy)) * (1.0 - 1e-5);
  if (!(th > 0.0) || !(tv > 0.0)) return 0;  // 1.1 x the half angle reaches 90 degrees
  // (int)(x * fx / z + cx) in [0, W) accepts x * fx / z + cx in (-1, W)
  return (p->cx + 1.0) / p->fx <= th && (W - p->cx) / p->fx <= th && (p->cy + 1.0) / p->fy <= tv && (H - p->cy) / p->fy <= tv;
}

// The six planes pcl::FrustumCulling::applyFilter builds for getFrustumCulledVoxels (tsdf_volume_octree.cpp:633-646) from
// the forward pose [PCL-recall: filters/impl/frustum_culling.hpp], for callers without Eigen / PCL of their own (the
// Python binding; cpu_tsdf::TSDFVolumeOctree builds them with the CALLER's Eigen instead): camera pose =
// trans.cast<float>() * cam2robot, i.e. view / up / right / T = the pose's z, -y, x columns and translation; the field of
// view 1.1 x the image's, through float setters; frustum corners, edge vectors, cross products and the plane offsets in
// float, one rounding per operation, left to right (this file is built with -ffp-contract=off); `tan` in double.
extern "C" int tsdf_hip_reference_cull_planes(const tsdf_params *p, const double trans[16], float planes[24]) {
  if (!p || !trans || !planes) return TSDF_HIP_E_INVALID;
  struct V3 { float v[3]; };
  auto axis = [&](int col, float sign) { V3 o; for (int r = 0; r < 3; ++r) o.v[r] = sign * (float)trans[4 * r + col]; return o; };
  const V3 view = axis(2, 1.f), up = axis(1, -1.f), right = axis(0, 1.f), T = axis(3, 1.f);
  const float hfov = (float)(1.1 * 2 * fabs(atan((double)(0.5 * p->image_width / p->fx)) * 180 / M_PI));   // :641, setHorizontalFOV(float)
  const float vfov = (float)(1.1 * 2 * fabs(atan((double)(0.5 * p->image_height / p->fy)) * 180 / M_PI));  // :642
  const float vrad = (float)(vfov * M_PI / 180), hrad = (float)(hfov * M_PI / 180);
  const float dist[2] = {p->max_sensor_dist, p->min_sensor_dist};  // far, near (:643-644)
  V3 centre[2], corner[2][4];  // corner: tl, tr, bl, br
  for (int f = 0; f < 2; ++f) {
    // (`tan` of a float divided by an int, evaluated in double: `tan (vfov_rad / 2)` with the double overload, as a C
    // compiler reads PCL's expression and as the stand-in the reference is compiled against here evaluates it)
    const float hgt = (float)(2 * tan((double)(vrad / 2)) * dist[f]), wid = (float)(2 * tan((double)(hrad / 2)) * dist[f]);
    for (int i = 0; i < 3; ++i) {
      const float c = T.v[i] + view.v[i] * dist[f], u = up.v[i] * hgt / 2, r = right.v[i] * wid / 2;
      centre[f].v[i] = c;
      corner[f][0].v[i] = c + u - r;
      corner[f][1].v[i] = c + u + r;
      corner[f][2].v[i] = c - u - r;
      corner[f][3].v[i] = c - u + r;
    }
  }
  auto minus = [](const V3 &a, const V3 &b) { V3 o; for (int i = 0; i < 3; ++i) o.v[i] = a.v[i] - b.v[i]; return o; };
  auto put = [&](int k, const V3 &e1, const V3 &e2, const V3 &through) {  // plane k: normal e1 x e2 through a point
    const float n[3] = {e1.v[1] * e2.v[2] - e1.v[2] * e2.v[1], e1.v[2] * e2.v[0] - e1.v[0] * e2.v[2], e1.v[0] * e2.v[1] - e1.v[1] * e2.v[0]};
    planes[4 * k] = n[0], planes[4 * k + 1] = n[1], planes[4 * k + 2] = n[2];
    planes[4 * k + 3] = -(through.v[0] * n[0] + (through.v[1] * n[1] + through.v[2] * n[2]));  // Vector3f::dot, 3-term tree
  };
  const V3 a = minus(corner[0][2], T), b = minus(corner[0][3], T), c = minus(corner[0][1], T), d = minus(corner[0][0], T);
  put(0, d, a, T);  // left
  put(1, b, c, T);  // right
  put(2, c, d, T);  // top
  put(3, a, b, T);  // bottom
  put(4, minus(corner[0][2], corner[0][3]), minus(corner[0][1], corner[0][3]), centre[0]);  // far:  (bl - br) x (tr - br)
  put(5, minus(corner[1][1], corner[1][3]), minus(corner[1][2], corner[1][3]), centre[1]);  // near: (tr - br) x (bl - br)
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_create(const tsdf_params *p, tsdf_handle *out) {
  if (!p || !out) return TSDF_HIP_E_INVALID;
  *out = nullptr;
  for (int a = 0; a < 3; ++a)
    if (p->res[a] <= 0 || !(p->size[a] > 0.f)) {
      tsdf_set_error("resolution and grid size must be positive");
      return TSDF_HIP_E_INVALID;
    }
  if (p->image_width <= 0 || p->image_height <= 0 || !(p->max_dist_neg > 0.f) || p->halo < 0 ||
      (p->xform_order != TSDF_XFORM_PCL_SSE && p->xform_order != TSDF_XFORM_LEFT_TO_RIGHT) ||
      p->layout < TSDF_LAYOUT_AUTO || p->layout > TSDF_LAYOUT_PACKED ||
      p->color_mode < TSDF_COLOR_RGB || p->color_mode > TSDF_COLOR_LAB) {
    tsdf_set_error("bad image size / truncation / halo / xform_order / layout / color_mode");
    return TSDF_HIP_E_INVALID;
  }
  const bool lab = p->integrate_color && p->color_mode == TSDF_COLOR_LAB;
  const bool rgbn = (p->integrate_color && p->color_mode == TSDF_COLOR_RGB_NORMALIZED) || lab;  // float colour state
  if (rgbn && p->layout == TSDF_LAYOUT_PACKED) {
    tsdf_set_error("RGB_NORMALIZED / LAB colour keeps float weights: no PACKED layout");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  const bool packable = !rgbn && p->max_weight >= 0.f && p->max_weight <= 255.f;  // false for NaN
  if (p->layout == TSDF_LAYOUT_PACKED && !packable) {
    tsdf_set_error("the PACKED layout needs 0 <= max_weight <= 255");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  int zb = p->z_begin, ze = p->z_end;
  if (zb == 0 && ze == 0) ze = p->res[2];
  if (zb < 0 || ze > p->res[2] || zb >= ze) {
    tsdf_set_error("z slab outside the grid");
    return TSDF_HIP_E_INVALID;
  }
  if (tsdf_hip_device_count() <= 0) {
    tsdf_set_error("no HIP device visible");
    return TSDF_HIP_E_NODEVICE;
  }
  int dev = p->device;
  if (dev < 0) TSDF_HIP_TRY(hipGetDevice(&dev));
  TSDF_ON_DEVICE(dev);

  tsdf_hip_volume *v = new tsdf_hip_volume;
  v->p = *p;
  v->p.z_begin = zb;
  v->p.z_end = ze;
  v->device = dev;
  v->nx = p->res[0];
  v->ny = p->res[1];
  v->nz = p->res[2];
  v->z_begin = zb;
  v->z_end = ze;
  v->z_first = zb - p->halo < 0 ? 0 : zb - p->halo;
  const int z_last = ze + p->halo > v->nz ? v->nz : ze + p->halo;
  v->nz_alloc = z_last - v->z_first;
  v->pitch = ((int64_t)v->nx + 3) / 4 * 4;
  const int64_t n = v->pitch * v->ny * v->nz_alloc;
  v->packed = p->layout == TSDF_LAYOUT_PACKED || (p->layout == TSDF_LAYOUT_AUTO && packable);
  v->p.layout = v->packed ? TSDF_LAYOUT_PACKED : TSDF_LAYOUT_F32W;
  v->kmax = v->packed ? (unsigned)ceilf(p->max_weight) : 0u;

  int rc = TSDF_HIP_OK;
  auto bail = [&](int code) {
    free_volume(v);
    return code;
  };
#define TRY_OR_BAIL(expr)                                                  \
  do {                                                                     \
    hipError_t _e = (expr);                                                \
    if (_e != hipSuccess) return bail(tsdf_hip_fail(_e, #expr, __FILE__, __LINE__)); \
  } while (0)
  auto alloc_planes = [&](PlaneSet &s) -> hipError_t {
    hipError_t e = hipMalloc(&s.d, n * sizeof(float));
    if (e == hipSuccess && !v->packed) e = hipMalloc(&s.w, n * sizeof(float));
    if (e == hipSuccess && p->integrate_color) e = hipMalloc(&s.rgb, n * sizeof(uint32_t));
    if (e == hipSuccess && v->packed && !p->integrate_color) e = hipMalloc(&s.k8, n);
    if (e != hipSuccess) release_planes(s);
    return e;
  };
  PlaneSet best;
  TRY_OR_BAIL(alloc_planes(best));
  {
    const size_t plane_bytes = (size_t)n * (4 + (v->packed ? 0 : 4) + (p->integrate_color ? 4 : 0) + (v->packed && !p->integrate_color ? 1 : 0));
    const int tries = plane_bytes >= ((size_t)4 << 30) ? std::max(1, tsdf_tuning().alloc_tries) : 1;  // small volumes: nothing to gain
    float best_ms = tries > 1 ? probe_planes(best, n, v->stream) : -1.f;
    v->alloc_probe_ms[0] = best_ms;
    v->alloc_tried = 1;
    // A sweep moves every plane byte once each way.  Placements fall into classes (2 x 68.7 GB with the non-temporal
    // sweep: 24.4-24.8, 25.2-25.5, 26-26.5 and 29-30 ms on MI355X; k_integrate follows: 16.5 / 16.7 / 16.9-17.0 / 17.6 ms);
    // one that streams at >= 5.55 TB/s is in the fastest class and ends the search, and while none has reached
    // 5.15 TB/s the search goes on for up to `tries` more candidates (round 3: a volume whose three candidates were all
    // of the slowest class integrated at 17.6 instead of 16.5 ms).
    const double swept = (double)n * (4 + (v->packed ? 0 : 4) + (p->integrate_color ? 4 : 0));  // (the probe leaves a count-byte plane alone)
    auto rate_tbps = [&](float ms) { return ms > 0.f ? 2.0 * swept / ((double)ms * 1e9) : 0.0; };
    for (int t = 1; t < std::min(8, 2 * tries) && best_ms > 0.f && tries > 1; ++t) {
      if (rate_tbps(best_ms) >= 5.55) break;
      if (t >= tries && rate_tbps(best_ms) >= 5.15) break;
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < plane_bytes + ((size_t)2 << 30)) break;
      PlaneSet cand;
      if (alloc_planes(cand) != hipSuccess) {
        (void)hipGetLastError();
        break;
      }
      const float ms = probe_planes(cand, n, v->stream);
      v->alloc_probe_ms[t < 8 ? t : 7] = ms;
      v->alloc_tried = t + 1;
      if (ms > 0.f && ms < best_ms) {
        release_planes(best);
        best = cand;
        best_ms = ms;
        v->alloc_chosen = t;
      } else {
        release_planes(cand);
      }
    }
  }
  v->d = best.d;
  v->w = best.w;
  v->rgb = best.rgb;
  v->k8 = best.k8;
  if (rgbn)
    for (int c = 0; c < (lab ? 3 : 4); ++c) TRY_OR_BAIL(hipMalloc(&v->cn[c], n * sizeof(float)));
  if (lab) {
    float lut[256];
    tsdf_lab_curve(lut);
    TRY_OR_BAIL(hipMalloc(&v->lab_lut, sizeof lut));
    TRY_OR_BAIL(hipMemcpy(v->lab_lut, lut, sizeof lut, hipMemcpyHostToDevice));
    TRY_OR_BAIL(hipMalloc(&v->lab_img, (size_t)p->image_width * p->image_height * sizeof(float4)));
  }
  for (int a = 0; a < 3; ++a) {
    tsdf_build_centers(p->res[a], tsdf_node_size(*p, a), v->h_ctr[a], &v->levels[a]);
    v->ctr_increasing[a] = true;
    for (size_t i = 1; i < v->h_ctr[a].size(); ++i) v->ctr_increasing[a] &= v->h_ctr[a][i - 1] < v->h_ctr[a][i];
    // pad the tables so float4 loads of the last (partial) quad stay in bounds; the pad is NaN, which
    // fails k_integrate's sensor-range test, so voxels of the pitch padding are never observed
    std::vector<float> padded(v->h_ctr[a]);
    padded.resize((size_t)p->res[a] + 4, NAN);
    TRY_OR_BAIL(hipMalloc(&v->ctr[a], padded.size() * sizeof(float)));
    TRY_OR_BAIL(hipMemcpy(v->ctr[a], padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  const size_t npx = (size_t)p->image_width * p->image_height;
  // one allocation [depth | bgra]: the integrate kernel addresses the frame through ONE buffer descriptor
  TRY_OR_BAIL(hipMalloc(&v->frame_depth, 2 * npx * sizeof(float)));
  v->frame_bgra = reinterpret_cast<uint32_t *>(v->frame_depth + npx);
  {
    const double cam[4] = {p->fx, p->fy, p->cx, p->cy};  // read by the rare exact-projection path only
    TRY_OR_BAIL(hipMalloc(&v->cam64, sizeof cam));
    TRY_OR_BAIL(hipMemcpy(v->cam64, cam, sizeof cam, hipMemcpyHostToDevice));
  }
  v->band_fx = (v->nx + 63) / 64;
  v->band_fy = (v->ny + 3) / 4;
  TRY_OR_BAIL(hipMalloc(&v->band, (size_t)v->band_fx * v->band_fy * v->nz_alloc));
  TRY_OR_BAIL(hipMalloc(&v->counter, 4096 * sizeof(unsigned long long)));
  TRY_OR_BAIL(hipMemset(v->counter, 0, 4096 * sizeof(unsigned long long)));
#undef TRY_OR_BAIL
  rc = tsdf_hip_reset(v);
  if (rc != TSDF_HIP_OK) return bail(rc);
  *out = v;
  return TSDF_HIP_OK;
}

// reset(): every voxel (d=-1, w=0) -- tsdf_volume_octree.cpp:213-218; RGBNode starts at 0,0,0
// (octree.h:177-180).
extern "C" int tsdf_hip_reset(tsdf_handle h) {
  if (!h) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_reset(h);
  TSDF_ENTER(h);
  const int64_t n = h->pitch * h->ny * h->nz_alloc;
  const float minus_one = -1.f;
  uint32_t bits;
  memcpy(&bits, &minus_one, 4);
  int rc = fill_u32(h, h->d, bits, n);
  if (rc) return rc;
  if (h->w) rc = fill_u32(h, h->w, 0u, n);
  if (rc) return rc;
  if (h->k8) TSDF_HIP_TRY(hipMemsetAsync(h->k8, 0, (size_t)n, h->stream));
  if (h->rgb) rc = fill_u32(h, h->rgb, 0u, n);
  if (rc) return rc;
  for (int c = 0; c < 4 && !rc; ++c)  // RGBNormalized starts at r_n = g_n = b_n = i = 0 (octree.h:217-222)
    if (h->cn[c]) rc = fill_u32(h, h->cn[c], 0u, n);
  if (rc) return rc;
  if (h->vm) rc = fill_u32(h, reinterpret_cast<float *>(h->vm), 0u, n);  // OctreeNode(): M_ (0), nsample_ (0)
  if (!rc && h->vn) rc = fill_u32(h, reinterpret_cast<float *>(h->vn), 0u, n);
  if (rc) return rc;
  TSDF_HIP_TRY(hipMemsetAsync(h->band, 0, (size_t)h->band_fx * h->band_fy * h->nz_alloc, h->stream));
  h->band_exact = true;  // every distance is -1: no voxel inside the band
  h->rest_state = 0;
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_destroy(tsdf_handle h) {
  if (!h) return TSDF_HIP_E_INVALID;
  TsdfDeviceScope scope(h->device);
  (void)hipStreamSynchronize(h->stream);
  free_volume(h);
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_set_stream(tsdf_handle h, void *hip_stream) {
  if (!h) return TSDF_HIP_E_INVALID;
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_set_stream (a stream belongs to one device)");
  {
    TSDF_ENTER(h);  // a frame that frame pairing holds back is launched on the stream it was committed for
  }
  h->stream = (hipStream_t)hip_stream;
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_synchronize(tsdf_handle h) {
  if (!h) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_synchronize(h);
  TSDF_ENTER(h);
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_centers(tsdf_handle h, int axis, float *out) {
  if (!h || axis < 0 || axis > 2 || !out) return TSDF_HIP_E_INVALID;
  memcpy(out, h->h_ctr[axis].data(), h->h_ctr[axis].size() * sizeof(float));
  return TSDF_HIP_OK;
}

// Which placement tsdf_hip_create kept: ms[i] = probe sweep of candidate i (up to 8; negative = not probed), *chosen = its
// index, return value = candidates tried (1 for small volumes, for alloc_tries = 1 and for multi handles' slabs in turn).
extern "C" int tsdf_hip_alloc_probe(tsdf_handle h, float ms[8], int32_t *chosen) {
  if (!h) return 0;
  const tsdf_hip_volume *s = h->multi ? tsdf_multi_first(h) : h;
  if (ms)
    for (int i = 0; i < 8; ++i) ms[i] = s->alloc_probe_ms[i];
  if (chosen) *chosen = s->alloc_chosen;
  return s->alloc_tried;
}

extern "C" int tsdf_hip_layout(tsdf_handle h) {
  return !h ? -1 : (h->packed ? TSDF_LAYOUT_PACKED : TSDF_LAYOUT_F32W);
}

extern "C" int tsdf_hip_device_planes(tsdf_handle h, float **d, float **w, uint32_t **rgb, int64_t *pitch,
                                      int32_t *z_first, int32_t *nz_alloc) {
  if (!h) return TSDF_HIP_E_INVALID;
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_device_planes");
  {
    TSDF_ENTER(h);  // a frame that frame pairing holds back is launched before the caller looks at the planes
  }
  h->band_exact = false;  // the caller may write through these pointers
  if (d) *d = h->d;
  if (w) *w = h->w;
  if (rgb) *rgb = h->rgb;
  if (pitch) *pitch = h->pitch;
  if (z_first) *z_first = h->z_first;
  if (nz_alloc) *nz_alloc = h->nz_alloc;
  return TSDF_HIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Block transfer: pack / unpack kernels through a contiguous scratch buffer, chunked along z.
struct BlockArgs {
  int x0, y0, zl0;  // zl0: local (allocated) plane index of the block's first plane
  int bx, by, bz;
  int ny;
  int64_t pitch;
};

template <bool TO_BLOCK>
static __global__ void __launch_bounds__(256)
k_block_f32(BlockArgs a, float *__restrict__ vol, float *__restrict__ blk) {
  const int64_t n = (int64_t)a.bx * a.by * a.bz;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % a.bx);
    const int64_t r = i / a.bx;
    const int y = (int)(r % a.by);
    const int z = (int)(r / a.by);
    const int64_t vi = ((int64_t)(a.zl0 + z) * a.ny + (a.y0 + y)) * a.pitch + (a.x0 + x);
    if (TO_BLOCK)
      blk[i] = vol[vi];
    else
      vol[vi] = blk[i];
  }
}

// Weights as floats, whatever the layout.  Storing into a PACKED volume fails (bad[0]++) for a weight that
// no observation count represents; the voxel is then left unchanged.
template <bool TO_BLOCK>
static __global__ void __launch_bounds__(256)
k_block_w(BlockArgs a, PlaneView pv, float *__restrict__ blk, unsigned *__restrict__ bad) {
  const int64_t n = (int64_t)a.bx * a.by * a.bz;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % a.bx);
    const int64_t r = i / a.bx;
    const int y = (int)(r % a.by);
    const int z = (int)(r / a.by);
    const int64_t vi = ((int64_t)(a.zl0 + z) * a.ny + (a.y0 + y)) * a.pitch + (a.x0 + x);
    if (TO_BLOCK) {
      blk[i] = tsdf_load_w(pv, vi);
    } else if (!pv.packed) {
      const_cast<float *>(pv.w)[vi] = blk[i];
    } else {
      unsigned k;
      if (!tsdf_encode_w(blk[i], pv.wmax, pv.kmax, k)) {
        atomicAdd(bad, 1u);
      } else if (pv.rgb) {
        uint32_t *c = const_cast<uint32_t *>(pv.rgb) + vi;
        *c = (*c & 0xffffffu) | (k << 24);
      } else {
        const_cast<uint8_t *>(pv.k8)[vi] = (uint8_t)k;
      }
    }
  }
}

template <bool TO_BLOCK>
static __global__ void __launch_bounds__(256)
k_block_rgb(BlockArgs a, uint32_t *__restrict__ vol, uint8_t *__restrict__ blk) {
  const int64_t n = (int64_t)a.bx * a.by * a.bz;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % a.bx);
    const int64_t r = i / a.bx;
    const int y = (int)(r % a.by);
    const int z = (int)(r / a.by);
    const int64_t vi = ((int64_t)(a.zl0 + z) * a.ny + (a.y0 + y)) * a.pitch + (a.x0 + x);
    if (TO_BLOCK) {
      const uint32_t c = vol[vi];
      blk[3 * i + 0] = (uint8_t)(c & 255u);
      blk[3 * i + 1] = (uint8_t)((c >> 8) & 255u);
      blk[3 * i + 2] = (uint8_t)((c >> 16) & 255u);
    } else {
      // byte 3 (the observation count of the PACKED layout, else zero) is preserved
      vol[vi] = (vol[vi] & 0xff000000u) | (uint32_t)blk[3 * i] | ((uint32_t)blk[3 * i + 1] << 8) |
                ((uint32_t)blk[3 * i + 2] << 16);
    }
  }
}

static int check_block(tsdf_handle h, int x0, int y0, int z0, int nx, int ny, int nz) {
  if (!h || nx <= 0 || ny <= 0 || nz <= 0 || x0 < 0 || y0 < 0 || x0 + nx > h->nx || y0 + ny > h->ny ||
      z0 < h->z_first || z0 + nz > h->z_first + h->nz_alloc) {
    tsdf_set_error("block outside the allocated slab");
    return TSDF_HIP_E_INVALID;
  }
  return TSDF_HIP_OK;
}

template <bool DOWN>
static int block_transfer(tsdf_handle h, int x0, int y0, int z0, int nx, int ny, int nz, float *d, float *w,
                          uint8_t *rgb) {
  int rc = check_block(h, x0, y0, z0, nx, ny, nz);
  if (rc) return rc;
  if (rgb && !h->rgb) {
    tsdf_set_error("volume has no colour plane");
    return TSDF_HIP_E_INVALID;
  }
  if (!DOWN && rgb && h->cn[0]) {
    tsdf_set_error("RGB_NORMALIZED / LAB colour state cannot be set from r,g,b bytes");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  TSDF_ENTER(h);
  const int64_t plane = (int64_t)nx * ny;
  int64_t max_planes = (int64_t)(64 << 20) / plane;  // <= 64 Mi voxels (256 MiB of floats) per chunk
  if (max_planes < 1) max_planes = 1;
  rc = tsdf_ensure_scratch(h, (size_t)std::min<int64_t>(max_planes, nz) * plane * sizeof(float));
  if (rc) return rc;
  for (int zc = 0; zc < nz; zc += (int)max_planes) {
    const int bz = (int)std::min<int64_t>(max_planes, nz - zc);
    const int64_t n = plane * bz;
    BlockArgs a{x0, y0, z0 + zc - h->z_first, nx, ny, bz, h->ny, h->pitch};
    unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    const PlaneView pv = tsdf_plane_view(h);
    float *hosts[2] = {d, w};
    for (int k = 0; k < 2; ++k) {
      if (!hosts[k]) continue;
      float *hp = hosts[k] + (int64_t)zc * plane;
      float *dev = (float *)h->scratch;
      if (DOWN) {
        if (k == 0)
          hipLaunchKernelGGL(k_block_f32<true>, dim3(blocks), dim3(256), 0, h->stream, a, h->d, dev);
        else
          hipLaunchKernelGGL(k_block_w<true>, dim3(blocks), dim3(256), 0, h->stream, a, pv, dev, (unsigned *)h->counter);
        TSDF_HIP_TRY(hipGetLastError());
        if ((rc = tsdf_to_host(h, hp, dev, n * sizeof(float)))) return rc;
      } else {
        if ((rc = tsdf_to_device(h, dev, hp, n * sizeof(float)))) return rc;
        if (k == 0) {
          hipLaunchKernelGGL(k_block_f32<false>, dim3(blocks), dim3(256), 0, h->stream, a, h->d, dev);
          TSDF_HIP_TRY(hipGetLastError());
          TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
        } else {
          unsigned bad = 0;
          TSDF_HIP_TRY(hipMemsetAsync(h->counter, 0, sizeof(unsigned), h->stream));
          hipLaunchKernelGGL(k_block_w<false>, dim3(blocks), dim3(256), 0, h->stream, a, pv, dev, (unsigned *)h->counter);
          TSDF_HIP_TRY(hipGetLastError());
          TSDF_HIP_TRY(hipMemcpyAsync(&bad, h->counter, sizeof bad, hipMemcpyDeviceToHost, h->stream));
          TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
          if (bad) {
            tsdf_set_error(std::to_string(bad) + " weights are not of the form min(k, max_weight): this volume "
                           "needs the F32W layout (tsdf_params.layout)");
            return TSDF_HIP_E_UNSUPPORTED;
          }
        }
      }
    }
    if (rgb && DOWN && h->lab_img) {
      // TSDF_COLOR_LAB: getRGB() is LAB2RGB of the voxel's float means through the HOST's pow (tsdf_lab2rgb_host)
      uint8_t *hp = rgb + (int64_t)zc * plane * 3;
      std::vector<float> lab((size_t)n * 3);
      for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL(k_block_f32<true>, dim3(blocks), dim3(256), 0, h->stream, a, h->cn[c], (float *)h->scratch);
        TSDF_HIP_TRY(hipGetLastError());
        if ((rc = tsdf_to_host(h, lab.data() + (size_t)c * n, h->scratch, n * sizeof(float)))) return rc;
      }
      std::vector<uint32_t> words((size_t)n);
      tsdf_lab2rgb_host_many(lab.data(), lab.data() + n, lab.data() + 2 * n, (size_t)n, words.data());
      for (int64_t i = 0; i < n; ++i) {
        hp[3 * i] = (uint8_t)(words[i] & 255u);
        hp[3 * i + 1] = (uint8_t)((words[i] >> 8) & 255u);
        hp[3 * i + 2] = (uint8_t)((words[i] >> 16) & 255u);
      }
    } else if (rgb) {
      uint8_t *hp = rgb + (int64_t)zc * plane * 3;
      if (DOWN) {
        hipLaunchKernelGGL(k_block_rgb<true>, dim3(blocks), dim3(256), 0, h->stream, a, h->rgb,
                           (uint8_t *)h->scratch);
        TSDF_HIP_TRY(hipGetLastError());
        if ((rc = tsdf_to_host(h, hp, h->scratch, n * 3))) return rc;
      } else {
        if ((rc = tsdf_to_device(h, h->scratch, hp, n * 3))) return rc;
        hipLaunchKernelGGL(k_block_rgb<false>, dim3(blocks), dim3(256), 0, h->stream, a, h->rgb,
                           (uint8_t *)h->scratch);
        TSDF_HIP_TRY(hipGetLastError());
        TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
      }
    }
  }
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_download(tsdf_handle h, int x0, int y0, int z0, int nx, int ny, int nz, float *d,
                                 float *w, uint8_t *rgb) {
  if (h && h->multi) return tsdf_multi_block(h, true, x0, y0, z0, nx, ny, nz, d, w, rgb);
  return block_transfer<true>(h, x0, y0, z0, nx, ny, nz, d, w, rgb);
}

// OctreeNode::M_ / nsample_ of a block of voxels ([z][y][x]; either pointer may be NULL): the state weight_by_variance_
// integrates with (hpp:203-204, octree.cpp:160-161,281-287).  Only volumes that weight by variance keep it.
template <bool DOWN>
static int variance_block(tsdf_handle h, int x0, int y0, int z0, int nx, int ny, int nz, float *M, int32_t *ns) {
  if (!h) return TSDF_HIP_E_INVALID;
  if (h->multi) return tsdf_multi_variance_block(h, DOWN, x0, y0, z0, nx, ny, nz, M, ns);
  int rc = check_block(h, x0, y0, z0, nx, ny, nz);
  if (rc) return rc;
  if (!h->vm || !h->vn) {
    tsdf_set_error("this volume keeps no M_ / nsample_ state (only volumes loaded with weight_by_variance do)");
    return TSDF_HIP_E_INVALID;
  }
  TSDF_ENTER(h);
  const int64_t plane = (int64_t)nx * ny;
  const int64_t max_planes = std::max<int64_t>(1, (int64_t)(64 << 20) / plane);
  rc = tsdf_ensure_scratch(h, (size_t)std::min<int64_t>(max_planes, nz) * plane * sizeof(float));
  if (rc) return rc;
  float *planes[2] = {h->vm, reinterpret_cast<float *>(h->vn)};  // (the int32 counts move as 4-byte words)
  float *hosts[2] = {M, reinterpret_cast<float *>(ns)};
  for (int zc = 0; zc < nz; zc += (int)max_planes) {
    const int bz = (int)std::min<int64_t>(max_planes, nz - zc);
    const int64_t n = plane * bz;
    BlockArgs a{x0, y0, z0 + zc - h->z_first, nx, ny, bz, h->ny, h->pitch};
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    for (int k = 0; k < 2; ++k) {
      if (!hosts[k]) continue;
      float *hp = hosts[k] + (int64_t)zc * plane, *dev = (float *)h->scratch;
      if (DOWN) {
        hipLaunchKernelGGL(k_block_f32<true>, dim3(blocks), dim3(256), 0, h->stream, a, planes[k], dev);
        TSDF_HIP_TRY(hipGetLastError());
        if ((rc = tsdf_to_host(h, hp, dev, n * sizeof(float)))) return rc;
      } else {
        if ((rc = tsdf_to_device(h, dev, hp, n * sizeof(float)))) return rc;
        hipLaunchKernelGGL(k_block_f32<false>, dim3(blocks), dim3(256), 0, h->stream, a, planes[k], dev);
        TSDF_HIP_TRY(hipGetLastError());
        TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
      }
    }
  }
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_download_variance_state(tsdf_handle h, int x0, int y0, int z0, int nx, int ny, int nz, float *M,
                                                int32_t *nsample) {
  return variance_block<true>(h, x0, y0, z0, nx, ny, nz, M, nsample);
}
extern "C" int tsdf_hip_upload_variance_state(tsdf_handle h, int x0, int y0, int z0, int nx, int ny, int nz, const float *M,
                                              const int32_t *nsample) {
  return variance_block<false>(h, x0, y0, z0, nx, ny, nz, const_cast<float *>(M), const_cast<int32_t *>(nsample));
}

// The float colour state of RGB_NORMALIZED (planes r_n, g_n, b_n, i) and LAB (planes L, A, B) voxels, which no
// reference accessor exposes (the members are public: octree.h:217-222, :296-298); the parity tests read it here.
extern "C" int tsdf_hip_download_color_state(tsdf_handle h, int plane_index, int z0, int nz, float *out) {
  if (!h || !out) return TSDF_HIP_E_INVALID;
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_download_color_state");
  if (plane_index < 0 || plane_index > 3 || !h->cn[plane_index]) {
    tsdf_set_error("this volume has no such colour-state plane");
    return TSDF_HIP_E_INVALID;
  }
  int rc = check_block(h, 0, 0, z0, h->nx, h->ny, nz);
  if (rc) return rc;
  TSDF_ENTER(h);
  const int64_t plane = (int64_t)h->nx * h->ny;
  const int64_t max_planes = std::max<int64_t>(1, (int64_t)(64 << 20) / plane);
  rc = tsdf_ensure_scratch(h, (size_t)std::min<int64_t>(max_planes, nz) * plane * sizeof(float));
  if (rc) return rc;
  for (int zc = 0; zc < nz; zc += (int)max_planes) {
    const int bz = (int)std::min<int64_t>(max_planes, nz - zc);
    const int64_t n = plane * bz;
    BlockArgs a{0, 0, z0 + zc - h->z_first, h->nx, h->ny, bz, h->ny, h->pitch};
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_block_f32<true>, dim3(blocks), dim3(256), 0, h->stream, a, h->cn[plane_index], (float *)h->scratch);
    TSDF_HIP_TRY(hipGetLastError());
    if ((rc = tsdf_to_host(h, out + (int64_t)zc * plane, h->scratch, n * sizeof(float)))) return rc;
  }
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_upload(tsdf_handle h, int x0, int y0, int z0, int nx, int ny, int nz, const float *d,
                               const float *w, const uint8_t *rgb) {
  if (h && h->multi)
    return tsdf_multi_block(h, false, x0, y0, z0, nx, ny, nz, const_cast<float *>(d), const_cast<float *>(w),
                            const_cast<uint8_t *>(rgb));
  if (h) h->band_exact = false;  // arbitrary distances arrive: the "band seen" flags no longer describe the planes
  return block_transfer<false>(h, x0, y0, z0, nx, ny, nz, const_cast<float *>(d), const_cast<float *>(w),
                               const_cast<uint8_t *>(rgb));
}

// ---------------------------------------------------------------------------------------------
// Whole-plane transfer between the SoA volume and packed DEVICE buffers ([nz][ny][nx], rgb as the
// volume's uint32 r|g<<8|b<<16), asynchronous on the handle's stream: the halo-exchange primitive.  The
// caller owns the buffers (e.g. torch CUDA tensors handed to RCCL send/recv).
template <bool TO_PACKED>
static __global__ void __launch_bounds__(256)
k_planes_u32(uint32_t *__restrict__ vol, uint32_t *__restrict__ packed, int nx, int ny, int nz, int zl0,
             int64_t pitch) {
  const int64_t n = (int64_t)nx * ny * nz;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % nx);
    const int64_t r = i / nx;
    const int64_t vi = ((int64_t)zl0 * ny + r) * pitch + x;  // r = z*ny + y
    if (TO_PACKED)
      packed[i] = vol[vi];
    else
      vol[vi] = packed[i];
  }
}

// Weight planes as floats / colour planes as r|g<<8|b<<16 for any layout (the exchanged buffers keep the
// documented format; a PACKED volume converts on the fly).  Planes coming from a peer slab of the same
// volume are always representable, so an unrepresentable weight is stored as count 0 and flagged.
template <bool TO_PACKED>
static __global__ void __launch_bounds__(256)
k_planes_w(PlaneView pv, float *__restrict__ packed, int nx, int ny, int nz, int zl0, int64_t pitch,
           unsigned *__restrict__ bad) {
  const int64_t n = (int64_t)nx * ny * nz;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % nx);
    const int64_t r = i / nx;
    const int64_t vi = ((int64_t)zl0 * ny + r) * pitch + x;
    if (TO_PACKED) {
      packed[i] = tsdf_load_w(pv, vi);
    } else {
      unsigned k;
      if (!tsdf_encode_w(packed[i], pv.wmax, pv.kmax, k)) atomicAdd(bad, 1u);
      if (pv.rgb) {
        uint32_t *c = const_cast<uint32_t *>(pv.rgb) + vi;
        *c = (*c & 0xffffffu) | (k << 24);
      } else {
        const_cast<uint8_t *>(pv.k8)[vi] = (uint8_t)k;
      }
    }
  }
}

template <bool TO_PACKED>
static __global__ void __launch_bounds__(256)
k_planes_rgb(uint32_t *__restrict__ vol, uint32_t *__restrict__ packed, int nx, int ny, int nz, int zl0,
             int64_t pitch) {
  const int64_t n = (int64_t)nx * ny * nz;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % nx);
    const int64_t r = i / nx;
    const int64_t vi = ((int64_t)zl0 * ny + r) * pitch + x;
    if (TO_PACKED)
      packed[i] = vol[vi] & 0xffffffu;
    else
      vol[vi] = (vol[vi] & 0xff000000u) | (packed[i] & 0xffffffu);
  }
}

template <bool TO_PACKED>
static int planes_device(tsdf_handle h, int z0, int nz, void *d, void *w, void *rgb) {
  if (!h || nz <= 0 || z0 < h->z_first || z0 + nz > h->z_first + h->nz_alloc) {
    tsdf_set_error("planes outside the allocated slab");
    return TSDF_HIP_E_INVALID;
  }
  if (rgb && !h->rgb) {
    tsdf_set_error("volume has no colour plane");
    return TSDF_HIP_E_INVALID;
  }
  TSDF_ENTER(h);
  const int64_t n = (int64_t)h->nx * h->ny * nz;
  const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 8192);
  const int zl0 = z0 - h->z_first;
  if (d) {
    hipLaunchKernelGGL(k_planes_u32<TO_PACKED>, dim3(blocks), dim3(256), 0, h->stream, (uint32_t *)h->d,
                       (uint32_t *)d, h->nx, h->ny, nz, zl0, h->pitch);
    TSDF_HIP_TRY(hipGetLastError());
  }
  if (w) {
    if (!h->packed) {
      hipLaunchKernelGGL(k_planes_u32<TO_PACKED>, dim3(blocks), dim3(256), 0, h->stream, (uint32_t *)h->w,
                         (uint32_t *)w, h->nx, h->ny, nz, zl0, h->pitch);
    } else {
      hipLaunchKernelGGL(k_planes_w<TO_PACKED>, dim3(blocks), dim3(256), 0, h->stream, tsdf_plane_view(h),
                         (float *)w, h->nx, h->ny, nz, zl0, h->pitch, (unsigned *)(h->counter + 1023));
    }
    TSDF_HIP_TRY(hipGetLastError());
  }
  if (rgb) {
    hipLaunchKernelGGL(k_planes_rgb<TO_PACKED>, dim3(blocks), dim3(256), 0, h->stream, h->rgb, (uint32_t *)rgb,
                       h->nx, h->ny, nz, zl0, h->pitch);
    TSDF_HIP_TRY(hipGetLastError());
  }
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_get_planes_device(tsdf_handle h, int z0, int nz, float *d, float *w, uint32_t *rgb) {
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_get_planes_device");
  return planes_device<true>(h, z0, nz, d, w, rgb);
}

extern "C" int tsdf_hip_set_planes_device(tsdf_handle h, int z0, int nz, const float *d, const float *w,
                                          const uint32_t *rgb) {
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_set_planes_device");
  // halo planes are always read by marching cubes (their flags are never consulted); owned planes written from outside
  // invalidate the flags
  if (h && z0 < h->z_end && z0 + nz > h->z_begin) h->band_exact = false;
  return planes_device<false>(h, z0, nz, const_cast<float *>(d), const_cast<float *>(w), const_cast<uint32_t *>(rgb));
}

#ifdef TSDF_HIP_TEST_HOOKS
// Test hook: position-dependent checksums of the owned planes, computed on the device: for each plane array the sum over
// all owned words of word * odd(index) mod 2^64 (a swap of two unequal words or a change of any word changes it).  How
// the tests compare WHOLE 2048^3 volumes integrated through different kernel instances / launch orders without moving
// 69 GB to the host.  out[0] = d, out[1] = w (0 in the PACKED layout), out[2] = rgb | count words (0 without colour),
// out[3] = the count bytes of a colourless PACKED volume (0 otherwise).
static __global__ void __launch_bounds__(256)
k_checksum(const uint32_t *__restrict__ p, int64_t n, unsigned long long *__restrict__ out) {
  unsigned long long acc = 0ull;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc += (unsigned long long)p[i] * (((unsigned long long)i * 2ull + 1ull) * 0x9E3779B97F4A7C15ull | 1ull);
  for (int off = 32; off; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63u) == 0u) atomicAdd(out, acc);
}

extern "C" int tsdf_hip_selftest_checksum(tsdf_handle h, uint64_t out[4]) {
  if (!h || !out) return TSDF_HIP_E_INVALID;
  TSDF_NOT_ON_MULTI(h, "tsdf_hip_selftest_checksum");
  TSDF_ENTER(h);
  const int64_t plane = h->pitch * h->ny, first = (int64_t)(h->z_begin - h->z_first) * plane,
                n = (int64_t)(h->z_end - h->z_begin) * plane;
  TSDF_HIP_TRY(hipMemsetAsync(h->counter, 0, 4 * sizeof(unsigned long long), h->stream));
  const unsigned grid = 256u * (unsigned)tsdf_tuning().blocks_per_cu;
  const void *arr[4] = {h->d, h->w, h->rgb, nullptr};
  for (int k = 0; k < 3; ++k)
    if (arr[k])
      hipLaunchKernelGGL(k_checksum, dim3(grid), dim3(256), 0, h->stream, reinterpret_cast<const uint32_t *>(arr[k]) + first, n, h->counter + k);
  if (h->k8)  // bytes, four to a word (pitch is a multiple of 4)
    hipLaunchKernelGGL(k_checksum, dim3(grid), dim3(256), 0, h->stream, reinterpret_cast<const uint32_t *>(h->k8 + first), n / 4, h->counter + 3);
  TSDF_HIP_TRY(hipGetLastError());
  unsigned long long c[4];
  TSDF_HIP_TRY(hipMemcpyAsync(c, h->counter, sizeof c, hipMemcpyDeviceToHost, h->stream));
  TSDF_HIP_TRY(hipStreamSynchronize(h->stream));
  for (int k = 0; k < 4; ++k) out[k] = c[k];
  return TSDF_HIP_OK;
}
#endif  // TSDF_HIP_TEST_HOOKS
