// libtsdf_hip.so -- one volume over several GPUs of one node, behind the SAME handle type.
//
// tsdf_hip_create_multi makes a tsdf_handle that owns no voxels itself but N Z-slab handles ("slabs"), one per entry
// of the device list, each an ordinary handle with `halo` extra planes on both sides.  Every C-ABI entry point that
// takes a handle forwards to the functions below when the handle is such a set, so cpu_tsdf::TSDFVolumeOctree (and
// anything else written against include/tsdf_hip.h) drives all GPUs of the node from ONE process without knowing it:
//
//   integrateCloud  the frame fans out to every slab's [depth | bgra] staging buffer -- from pinned host memory over
//                   each GPU's own PCIe link (host entry points), or from the GPU that holds it by
//                   hipMemcpyPeerAsync over xGMI (device / staged entry points) -- and every slab runs k_integrate
//                   on its own planes on its own stream.  No voxel ever crosses a link.
//   reconstruct     plane z_end of each slab comes from its upper neighbour (one raw plane per array, peer copy), the
//                   slabs mesh concurrently (one host thread each), the per-slab triangle lists -- each already in
//                   the reference's order -- are merged by Morton key on the host.
//   renderView      ray hand-off (tsdf_query.hip): halos refreshed on both sides, then rounds of
//                   tsdf_hip_raycast_advance per slab with the ray records merged on the first slab's GPU.
//   getFxn etc.     the slab that owns a point's lower-corner plane answers.
//   save / load     through the block callbacks; a block is split at slab boundaries.
//
// The multi-PROCESS form of the same partition (one rank per GPU, RCCL) is cpu_tsdf_amd/zslab.py.
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "tsdf_common.h"

struct tsdf_hip_multi {
  std::vector<tsdf_handle> slab;
  std::vector<hipStream_t> stream;  // per slab: its own non-blocking stream (slabs overlap; every dependency is an event)
  int halo = 0;
  // pinned frame staging for the host entry points: two slots so that tsdf_hip_integrate_async can return while the
  // uploads of the previous frame are still in flight
  float *pinned[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> uploaded[2];  // per slot, per slab: the slab's H2D copy out of the slot has finished
  unsigned long long frames = 0;
  bool pairing = false;                 // tsdf_hip_set_frame_pairing on the set: every slab pairs the frames of its own ring
  std::vector<hipEvent_t> ev, ev2;      // per slab: general cross-device ordering (two, so a wait may follow a wait)
  bool halo1_fresh = false, halo_all_fresh = false;  // plane z_end of every slab / the whole halo is current
  int frame_staged = 0;
  // renderView: compact ray lists (tsdf_query.hip).  Per slab, on its device: the records it is responsible for, the
  // same sorted by destination after a round, the finished rays of the round, routing counters + one "touched a plane
  // this slab does not hold" counter; on the first slab: the finished rays in transit and the image.  Pinned: the
  // count table the host sizes the copies with.
  std::vector<int *> ray_list, ray_outbox, ray_finbox;
  std::vector<unsigned *> ray_counters;  // 2 * (n_slab + 1) routing words + 1 incomplete word
  int *ray_fin_in = nullptr;
  float *ray_image = nullptr;
  unsigned *ray_table = nullptr;         // pinned: [n_slab][n_slab + 2] (counts per destination, finished, incomplete)
  size_t ray_cap = 0;                    // rays the buffers above hold
  uint64_t rv_stats[4] = {0, 0, 0, 0};   // last renderView: rounds, records handed between slabs, bytes between slabs, host waits
  // per-slab k_integrate timing (tsdf_hip_multi_timing): event pairs around every slab's launches while enabled
  bool timing = false;
  std::vector<std::vector<hipEvent_t>> t_ev;  // per slab: start, stop, start, stop, ...
  std::vector<size_t> t_used;
  // Copies between two slabs' devices: hipMemcpyPeerAsync where the driver grants peer access (xGMI), else -- access
  // refused, or TSDF_HIP_NO_PEER=1, which routes EVERY cross-slab copy this way so that a one-GPU box can test it --
  // through a pinned relay buffer on the host: device -> host on a relay stream of the source device, host -> device on
  // the receiving slab's stream, chunk by chunk, every step ordered by events (tsdf_multi_copy).
  bool relay_all = false;
  std::vector<std::pair<int, int>> no_peer;  // device pairs whose peer access was refused
  char *relay = nullptr;                     // pinned
  size_t relay_cap = 0;
  // per device (an event may only be recorded on a stream of the device it was created on; waiting across devices is fine):
  // a relay stream for the device-to-host half, and three events -- receiver's marker, relay filled, relay drained
  struct RelayDev {
    int dev = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  };
  std::vector<RelayDev> relay_dev;
  hipEvent_t relay_last_drain = nullptr;  // the `drained` event of the copy that used the relay last (any device)
  bool relay_used = false;
  uint64_t relay_bytes = 0;                  // bytes that took the relay (tsdf_hip_multi_render_stats-style report)
  // merged mesh of the last tsdf_hip_march (host)
  std::vector<float> verts;
  std::vector<uint8_t> rgb;
  std::vector<uint64_t> cell;
  bool mesh_has_rgb = false;
  float mc_ms[3] = {0.f, 0.f, 0.f};
  uint64_t mc_ncells = 0;
};

static void slab_range(int nz, int n, int k, int *zb, int *ze) {  // contiguous, balanced (zslab.py slab_range)
  const int base = nz / n, extra = nz % n;
  *zb = k * base + std::min(k, extra);
  *ze = *zb + base + (k < extra ? 1 : 0);
}

static int owner_of(const tsdf_hip_multi *m, int z) {
  for (size_t k = 0; k < m->slab.size(); ++k)
    if (z >= m->slab[k]->z_begin && z < m->slab[k]->z_end) return (int)k;
  return -1;
}

void tsdf_multi_free(tsdf_hip_volume *v) {
  tsdf_hip_multi *m = v->multi;
  if (!m) return;
  for (tsdf_handle s : m->slab) (void)tsdf_hip_synchronize(s);
  for (size_t k = 0; k < m->slab.size(); ++k) {
    TsdfDeviceScope scope(m->slab[k]->device);
    if (k < m->ray_list.size() && m->ray_list[k]) (void)hipFree(m->ray_list[k]);
    if (k < m->ray_outbox.size() && m->ray_outbox[k]) (void)hipFree(m->ray_outbox[k]);
    if (k < m->ray_finbox.size() && m->ray_finbox[k]) (void)hipFree(m->ray_finbox[k]);
    if (k < m->ray_counters.size() && m->ray_counters[k]) (void)hipFree(m->ray_counters[k]);
    if (k < m->ev.size() && m->ev[k]) (void)hipEventDestroy(m->ev[k]);
    if (k < m->ev2.size() && m->ev2[k]) (void)hipEventDestroy(m->ev2[k]);
    for (int s = 0; s < 2; ++s)
      if (k < m->uploaded[s].size() && m->uploaded[s][k]) (void)hipEventDestroy(m->uploaded[s][k]);
    if (k < m->t_ev.size())
      for (hipEvent_t e : m->t_ev[k]) (void)hipEventDestroy(e);
  }
  if (!m->slab.empty()) {
    TsdfDeviceScope scope(m->slab[0]->device);
    if (m->ray_fin_in) (void)hipFree(m->ray_fin_in);
    if (m->ray_image) (void)hipFree(m->ray_image);
  }
  if (m->ray_table) (void)hipHostFree(m->ray_table);
  if (m->relay) (void)hipHostFree(m->relay);
  for (auto &rd : m->relay_dev) {
    TsdfDeviceScope scope(rd.dev);
    if (rd.stream) (void)hipStreamDestroy(rd.stream);
    for (hipEvent_t e : rd.ev)
      if (e) (void)hipEventDestroy(e);
  }
  for (int s = 0; s < 2; ++s)
    if (m->pinned[s]) (void)hipHostFree(m->pinned[s]);
  for (size_t k = 0; k < m->slab.size(); ++k) {
    const int dev = m->slab[k]->device;
    (void)tsdf_hip_destroy(m->slab[k]);
    if (k < m->stream.size() && m->stream[k]) {
      TsdfDeviceScope scope(dev);
      (void)hipStreamDestroy(m->stream[k]);
    }
  }
  delete m;
  v->multi = nullptr;
}

extern "C" int tsdf_hip_create_multi(const tsdf_params *p, const int32_t *devices, int n_devices, tsdf_handle *out) {
  if (!p || !devices || n_devices < 1 || !out) return TSDF_HIP_E_INVALID;
  *out = nullptr;
  if (p->z_begin != 0 || (p->z_end != 0 && p->z_end != p->res[2])) {
    tsdf_set_error("tsdf_hip_create_multi partitions the WHOLE grid: leave z_begin / z_end at 0");
    return TSDF_HIP_E_INVALID;
  }
  if (p->res[2] < n_devices) {
    tsdf_set_error("fewer z planes than devices");
    return TSDF_HIP_E_INVALID;
  }
  const int ndev = tsdf_hip_device_count();
  if (ndev <= 0) {
    tsdf_set_error("no HIP device visible");
    return TSDF_HIP_E_NODEVICE;
  }
  for (int k = 0; k < n_devices; ++k)
    if (devices[k] < 0 || devices[k] >= ndev) {
      tsdf_set_error("device ordinal out of range");
      return TSDF_HIP_E_INVALID;
    }
  tsdf_hip_volume *v = new tsdf_hip_volume;
  tsdf_hip_multi *m = new tsdf_hip_multi;
  v->multi = m;
  v->p = *p;
  v->p.z_begin = 0;
  v->p.z_end = p->res[2];
  v->p.device = devices[0];
  v->device = devices[0];
  v->nx = p->res[0], v->ny = p->res[1], v->nz = p->res[2];
  v->z_begin = 0, v->z_end = v->nz, v->z_first = 0, v->nz_alloc = 0;
  v->pitch = ((int64_t)v->nx + 3) / 4 * 4;
  for (int a = 0; a < 3; ++a) {
    if (p->res[a] <= 0 || !(p->size[a] > 0.f)) {
      tsdf_multi_free(v);
      delete v;
      tsdf_set_error("resolution and grid size must be positive");
      return TSDF_HIP_E_INVALID;
    }
    tsdf_build_centers(p->res[a], tsdf_node_size(*p, a), v->h_ctr[a], &v->levels[a]);
  }
  if (n_devices > TSDF_MAX_SLABS) {  // (before any slab is allocated: ADVICE r03)
    tsdf_set_error("too many slabs");
    delete v->multi;
    v->multi = nullptr;
    delete v;
    return TSDF_HIP_E_INVALID;
  }
  // halo: what renderView's ray hand-off needs (the refinement walk and the normal's samples look back / ahead);
  // marching cubes and sampling use the first plane of it
  m->halo = n_devices > 1 ? std::max(1, tsdf_hip_render_halo(p)) : 0;
  auto fail = [&](int rc) {
    tsdf_multi_free(v);
    delete v;
    return rc;
  };
  for (int k = 0; k < n_devices; ++k) {
    tsdf_params q = *p;
    slab_range(p->res[2], n_devices, k, &q.z_begin, &q.z_end);
    q.halo = m->halo;
    q.device = devices[k];
    tsdf_handle s = nullptr;
    const int rc = tsdf_hip_create(&q, &s);
    if (rc) return fail(rc);
    m->slab.push_back(s);
    m->stream.push_back(nullptr);
    // Every slab works on its OWN non-blocking stream -- also when several slabs share a device -- so the slabs
    // overlap and every cross-slab dependency has to be an explicit event (on the null stream a missing one would be
    // hidden by the implicit serialisation).  What tsdf_hip_create queued on the null stream finishes first.
    TsdfDeviceScope scope(devices[k]);
    if (hipDeviceSynchronize() != hipSuccess || hipStreamCreateWithFlags(&m->stream[k], hipStreamNonBlocking) != hipSuccess) {
      tsdf_set_error("hipStreamCreate failed");
      return fail(TSDF_HIP_E_HIP);
    }
    s->stream = m->stream[k];
  }
  v->packed = m->slab[0]->packed;
  v->kmax = m->slab[0]->kmax;
  v->p.layout = m->slab[0]->p.layout;
  // peer access between every pair of distinct devices (xGMI).  A pair the driver refuses is remembered and its copies
  // take the host relay (tsdf_multi_copy): slower, never a hang, and said once on stderr.
  {
    const char *e = getenv("TSDF_HIP_NO_PEER");
    m->relay_all = e && *e && atoi(e) != 0;
  }
  for (int a = 0; a < n_devices; ++a)
    for (int b = 0; b < n_devices; ++b)
      if (devices[a] != devices[b]) {
        int can = 0;
        bool ok = false;
        if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
          TsdfDeviceScope scope(devices[a]);
          const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
          ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
          if (e != hipSuccess) (void)hipGetLastError();
        } else {
          (void)hipGetLastError();
        }
        if (!ok && std::find(m->no_peer.begin(), m->no_peer.end(), std::make_pair(devices[a], devices[b])) == m->no_peer.end()) {
          m->no_peer.push_back(std::make_pair((int)devices[a], (int)devices[b]));
          fprintf(stderr, "libtsdf_hip: no peer access from GPU %d to GPU %d: frames, halo planes and ray records between their "
                          "slabs go through pinned host memory\n", devices[a], devices[b]);
        }
      }
  m->ev.resize(n_devices, nullptr);
  m->ev2.resize(n_devices, nullptr);
  m->uploaded[0].resize(n_devices, nullptr);
  m->uploaded[1].resize(n_devices, nullptr);
  m->ray_list.resize(n_devices, nullptr);
  m->ray_outbox.resize(n_devices, nullptr);
  m->ray_finbox.resize(n_devices, nullptr);
  m->ray_counters.resize(n_devices, nullptr);
  m->t_ev.resize(n_devices);
  m->t_used.resize(n_devices, 0);
  for (int k = 0; k < n_devices; ++k) {
    TsdfDeviceScope scope(devices[k]);
    if (hipEventCreateWithFlags(&m->ev[k], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev2[k], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->uploaded[0][k], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->uploaded[1][k], hipEventDisableTiming) != hipSuccess) {
      tsdf_set_error("hipEventCreate failed");
      return fail(TSDF_HIP_E_HIP);
    }
  }
  *out = v;
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_slab_count(tsdf_handle h) { return !h ? 0 : (h->multi ? (int)h->multi->slab.size() : 1); }

extern "C" int tsdf_hip_slab_info(tsdf_handle h, int k, int32_t *device, int32_t *z_begin, int32_t *z_end, int32_t *halo) {
  if (!h) return TSDF_HIP_E_INVALID;
  const tsdf_hip_volume *s = h;
  if (h->multi) {
    if (k < 0 || k >= (int)h->multi->slab.size()) return TSDF_HIP_E_INVALID;
    s = h->multi->slab[k];
  } else if (k != 0) {
    return TSDF_HIP_E_INVALID;
  }
  if (device) *device = s->device;
  if (z_begin) *z_begin = s->z_begin;
  if (z_end) *z_end = s->z_end;
  if (halo) *halo = s->p.halo;
  return TSDF_HIP_OK;
}

int tsdf_multi_reset(tsdf_handle h) {
  if (const int rc_flush = tsdf_multi_flush(h)) return rc_flush;  // (frame pairing: slabs launch what they hold first)
  tsdf_hip_multi *m = h->multi;
  for (tsdf_handle s : m->slab) {
    const int rc = tsdf_hip_reset(s);
    if (rc) return rc;
  }
  m->halo1_fresh = m->halo_all_fresh = true;  // every plane, halo included, is (d = -1, w = 0)
  m->frame_staged = 0;
  return TSDF_HIP_OK;
}

// ---- frame pairing on a set (round 6) -------------------------------------------------------------
// Every slab has the ring of a single handle (tsdf_hip_pipeline, tsdf_integrate.hip) and pairs the frames that pass through
// it by itself: a committed frame is copied into the slab's next ring slot at once -- from the set's pinned slot over the
// slab's own PCIe link, or from a device buffer over xGMI -- and its kernel waits for the partner; with both at hand the slab
// runs k_integrate2 where BOTH poses see all of THAT slab (two launches in order otherwise: slabs decide independently).
// tsdf_multi_flush: every entry point of the set that reads or writes voxels first lets the slabs launch what they hold.
int tsdf_pipeline_commit_from(tsdf_handle s, const void *src, int src_dev, const float T[12], bool pairing, hipEvent_t uploaded,
                              int (*copy)(void *ctx, void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t st), void *ctx);
int tsdf_pipeline_pair_from(tsdf_handle s, const void *src_a, const void *src_b, int src_dev, const float TA[12], const float *planes_a,
                            const float TB[12], const float *planes_b, bool count, bool *fused,
                            int (*copy)(void *ctx, void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t st), void *ctx);
static int tsdf_multi_copy(tsdf_hip_multi *m, void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t stream, bool cross_slab);
static int copy_thunk(void *ctx, void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t st) {
  return tsdf_multi_copy(static_cast<tsdf_hip_multi *>(ctx), dst, dst_dev, src, src_dev, bytes, st, true);
}

int tsdf_multi_flush(tsdf_handle h) {
  for (tsdf_handle s : h->multi->slab)
    if (s->pair_pending) {
      TSDF_ON_DEVICE(s->device);
      const int rc = tsdf_pipeline_flush(s);
      if (rc) return rc;
    }
  return TSDF_HIP_OK;
}

int tsdf_multi_set_frame_pairing(tsdf_handle h, int on) {
  h->multi->pairing = on != 0;
  return on ? TSDF_HIP_OK : tsdf_multi_flush(h);  // (switching it off launches what was waiting)
}

int tsdf_multi_synchronize(tsdf_handle h) {
  for (tsdf_handle s : h->multi->slab) {
    const int rc = tsdf_hip_synchronize(s);
    if (rc) return rc;
  }
  return TSDF_HIP_OK;
}

int tsdf_multi_set_weighting(tsdf_handle h, int by_depth, int by_variance) {
  for (tsdf_handle s : h->multi->slab) {
    const int rc = tsdf_hip_set_weighting(s, by_depth, by_variance);
    if (rc) return rc;
  }
  h->weight_by_depth = by_depth != 0;
  h->weight_by_variance = by_variance != 0;
  return TSDF_HIP_OK;
}

int tsdf_multi_set_reference_cull(tsdf_handle h, const float planes[24]) {
  for (tsdf_handle s : h->multi->slab) {
    const int rc = tsdf_hip_set_reference_cull(s, planes);
    if (rc) return rc;
  }
  return TSDF_HIP_OK;
}

// ---- integrateCloud ------------------------------------------------------------------------------------------------
static int integrate_all(tsdf_handle h, const float T[12], uint64_t *n_observed) {
  tsdf_hip_multi *m = h->multi;
  m->halo1_fresh = m->halo_all_fresh = false;
  // every slab's launch is queued before any count is read back: the slabs (GPUs) integrate concurrently whether or
  // not the caller asked for n_observed
  for (size_t k = 0; k < m->slab.size(); ++k) {
    tsdf_handle s = m->slab[k];
    TSDF_ON_DEVICE(s->device);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (m->timing) {
      std::vector<hipEvent_t> &ev = m->t_ev[k];
      while (ev.size() < m->t_used[k] + 2) {
        hipEvent_t e = nullptr;
        TSDF_HIP_TRY(hipEventCreate(&e));
        ev.push_back(e);
      }
      t0 = ev[m->t_used[k]], t1 = ev[m->t_used[k] + 1];
      m->t_used[k] += 2;
      TSDF_HIP_TRY(hipEventRecord(t0, s->stream));
    }
    const int rc = tsdf_integrate_launch(s, s->frame_depth, s->p.integrate_color ? s->frame_bgra : nullptr, T, n_observed != nullptr);
    if (rc) return rc;
    if (t1) TSDF_HIP_TRY(hipEventRecord(t1, s->stream));
  }
  if (n_observed) {
    unsigned long long total = 0, changed = 0, implied = 0, read_bytes = 0;
    bool implied_on = true;
    for (tsdf_handle s : m->slab) {
      TSDF_ON_DEVICE(s->device);
      uint64_t n = 0;
      const int rc = tsdf_integrate_collect(s, &n);
      if (rc) return rc;
      total += n;
      changed += s->last_changed_bytes;
      implied += s->last_implied;
      read_bytes += s->last_read_bytes;
      implied_on = implied_on && s->last_implied_on;
    }
    *n_observed = total;
    h->last_observed = total;
    h->last_changed_bytes = changed;
    h->last_implied = implied;
    h->last_read_bytes = read_bytes;
    h->last_implied_on = implied_on;
  }
  return TSDF_HIP_OK;
}

// Per-slab k_integrate time (bench.py --host inprocess): while enabled, every slab's launches are bracketed by HIP events
// on the slab's own stream.  tsdf_hip_multi_kernel_ms synchronises and returns the summed milliseconds and the
// number of launches of slab k since timing was enabled (or last read), then forgets them.
extern "C" int tsdf_hip_multi_timing(tsdf_handle h, int enable) {
  if (!h || !h->multi) return TSDF_HIP_E_INVALID;
  h->multi->timing = enable != 0;
  for (size_t k = 0; k < h->multi->t_used.size(); ++k) h->multi->t_used[k] = 0;
  return TSDF_HIP_OK;
}

extern "C" int tsdf_hip_multi_kernel_ms(tsdf_handle h, int k, float *ms_sum, int32_t *launches) {
  if (!h || !h->multi || k < 0 || k >= (int)h->multi->slab.size() || !ms_sum) return TSDF_HIP_E_INVALID;
  tsdf_hip_multi *m = h->multi;
  TSDF_ON_DEVICE(m->slab[k]->device);
  TSDF_HIP_TRY(hipStreamSynchronize(m->slab[k]->stream));
  float sum = 0.f;
  for (size_t i = 0; i + 1 < m->t_used[k]; i += 2) {
    float ms = 0.f;
    TSDF_HIP_TRY(hipEventElapsedTime(&ms, m->t_ev[k][i], m->t_ev[k][i + 1]));
    sum += ms;
  }
  *ms_sum = sum;
  if (launches) *launches = (int32_t)(m->t_used[k] / 2);
  m->t_used[k] = 0;
  return TSDF_HIP_OK;
}

// Host frame -> pinned slot -> every slab's staging buffer, each over its own GPU's PCIe link and on its own stream.
int tsdf_multi_frame_begin(tsdf_handle h, float **depth, uint8_t **bgra) {
  tsdf_hip_multi *m = h->multi;
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  const int slot = (int)(m->frames & 1ull);
  if (!m->pinned[slot]) TSDF_HIP_TRY(hipHostMalloc((void **)&m->pinned[slot], npx * 8, hipHostMallocPortable));
  if (m->frames >= 2)  // the slot's previous uploads (two frames ago) must have left it
    for (size_t k = 0; k < m->slab.size(); ++k) TSDF_HIP_TRY(hipEventSynchronize(m->uploaded[slot][k]));
  *depth = m->pinned[slot];
  if (bgra) *bgra = h->p.integrate_color ? reinterpret_cast<uint8_t *>(m->pinned[slot] + npx) : nullptr;
  return TSDF_HIP_OK;
}

static int upload_slot(tsdf_handle h) {
  tsdf_hip_multi *m = h->multi;
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  const bool color = h->p.integrate_color != 0;
  const int slot = (int)(m->frames & 1ull);
  if (!m->pinned[slot]) {
    tsdf_set_error("tsdf_hip_frame_commit without tsdf_hip_frame_begin");
    return TSDF_HIP_E_INVALID;
  }
  for (size_t k = 0; k < m->slab.size(); ++k) {
    tsdf_handle s = m->slab[k];
    TSDF_ON_DEVICE(s->device);
    TSDF_HIP_TRY(hipMemcpyAsync(s->frame_depth, m->pinned[slot], npx * 4 * (color ? 2 : 1), hipMemcpyHostToDevice, s->stream));
    TSDF_HIP_TRY(hipEventRecord(m->uploaded[slot][k], s->stream));
  }
  m->frames++;
  return TSDF_HIP_OK;
}

int tsdf_multi_frame_commit(tsdf_handle h, const float T[12]) {
  tsdf_hip_multi *m = h->multi;
  if (m->pairing) {  // through every slab's own ring: the slab holds the frame back for a partner, or sweeps once for both
    const int slot = (int)(m->frames & 1ull);
    if (!m->pinned[slot]) {
      tsdf_set_error("tsdf_hip_frame_commit without tsdf_hip_frame_begin");
      return TSDF_HIP_E_INVALID;
    }
    m->halo1_fresh = m->halo_all_fresh = false;
    for (size_t k = 0; k < m->slab.size(); ++k) {
      const int rc = tsdf_pipeline_commit_from(m->slab[k], m->pinned[slot], -1, T, true, m->uploaded[slot][k], copy_thunk, m);
      if (rc) return rc;
    }
    m->frames++;
    return TSDF_HIP_OK;
  }
  int rc = tsdf_multi_flush(h);
  if (!rc) rc = upload_slot(h);
  return rc ? rc : integrate_all(h, T, nullptr);
}

// tsdf_hip_integrate_device2 on a set: both device frames go to every slab's ring, every slab sweeps once where both poses
// see all of it.  *fused = 1 when EVERY slab did; n_observed[0..1] = the two frames' observed voxels over all slabs.
int tsdf_multi_integrate_device2(tsdf_handle h, const float *da, const uint32_t *ca, const float TA[12], const float *planes_a, const float *db,
                                 const uint32_t *cb, const float TB[12], const float *planes_b, uint64_t *n_observed, int32_t *fused) {
  tsdf_hip_multi *m = h->multi;
  const bool color = h->p.integrate_color != 0;
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  if (color && (!ca || !cb)) {
    tsdf_set_error("integrate_color is set but no colour image was given");
    return TSDF_HIP_E_INVALID;
  }
  // the ring slots hold [depth | bgra] back to back, like the frames bench.py and zslab.py hand over; anything else goes
  // frame by frame through the staging path
  const bool packed_frames = !color || (reinterpret_cast<const float *>(ca) == da + npx && reinterpret_cast<const float *>(cb) == db + npx);
  hipPointerAttribute_t attr;
  int src_dev = -1;
  if (hipPointerGetAttributes(&attr, da) == hipSuccess)
    src_dev = attr.device;
  else
    (void)hipGetLastError();
  if (!packed_frames || src_dev < 0) {
    int rc = tsdf_multi_flush(h);
    if (!rc) rc = tsdf_multi_set_reference_cull(h, planes_a);  // (nullptr switches the cull off, as on a single handle)
    if (!rc) rc = tsdf_multi_integrate_device(h, da, ca, TA, n_observed);
    if (!rc) rc = tsdf_multi_set_reference_cull(h, planes_b);
    if (!rc) rc = tsdf_multi_integrate_device(h, db, cb, TB, n_observed ? n_observed + 1 : nullptr);
    return rc;
  }
  m->halo1_fresh = m->halo_all_fresh = false;
  bool all_fused = true;
  for (tsdf_handle s : m->slab) {
    bool f = false;
    const int rc = tsdf_pipeline_pair_from(s, da, db, src_dev, TA, planes_a, TB, planes_b, n_observed != nullptr, &f, copy_thunk, m);
    if (rc) return rc;
    all_fused = all_fused && f;
  }
  if (fused) *fused = all_fused ? 1 : 0;
  if (n_observed) {
    n_observed[0] = n_observed[1] = 0;
    unsigned long long observed = 0, changed = 0, implied = 0, read_bytes = 0;
    for (tsdf_handle s : m->slab) {
      TSDF_ON_DEVICE(s->device);
      uint64_t n2[2] = {0, 0};
      const int rc = tsdf_integrate_collect2(s, n2);
      if (rc) return rc;
      n_observed[0] += n2[0], n_observed[1] += n2[1];
      observed += s->last_observed, changed += s->last_changed_bytes, implied += s->last_implied, read_bytes += s->last_read_bytes;
    }
    h->last_observed = observed, h->last_changed_bytes = changed, h->last_implied = implied, h->last_read_bytes = read_bytes;
  }
  return TSDF_HIP_OK;
}

int tsdf_multi_integrate(tsdf_handle h, const float *depth, const uint8_t *bgra, const float T[12], uint64_t *n_observed,
                         bool asynchronous) {
  if (const int rc_flush = tsdf_multi_flush(h)) return rc_flush;  // (frame pairing: slabs launch what they hold first)
  if (h->p.integrate_color && !bgra) {
    tsdf_set_error("integrate_color is set but no colour image was given");
    return TSDF_HIP_E_INVALID;
  }
  float *sd = nullptr;
  uint8_t *sc = nullptr;
  int rc = tsdf_multi_frame_begin(h, &sd, &sc);
  if (rc) return rc;
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  memcpy(sd, depth, npx * 4);
  if (sc) memcpy(sc, bgra, npx * 4);
  if ((rc = upload_slot(h))) return rc;
  if ((rc = integrate_all(h, T, n_observed))) return rc;
  return asynchronous ? TSDF_HIP_OK : tsdf_multi_synchronize(h);
}

// One copy between (possibly) two devices, queued on `stream`, a stream of the RECEIVING device that has already been
// made to wait for the source data (every caller orders source -> receiver by an event first).
static int tsdf_multi_copy(tsdf_hip_multi *m, void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t stream,
                           bool cross_slab = true) {
  if (!bytes) return TSDF_HIP_OK;
  if (!cross_slab) {  // within one slab: never a link
    TSDF_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
    return TSDF_HIP_OK;
  }
  const bool refused = std::find(m->no_peer.begin(), m->no_peer.end(), std::make_pair(dst_dev, src_dev)) != m->no_peer.end() ||
                       std::find(m->no_peer.begin(), m->no_peer.end(), std::make_pair(src_dev, dst_dev)) != m->no_peer.end();
  if (!m->relay_all && !refused) {
    if (dst_dev == src_dev) {
      TSDF_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
    } else {
      TSDF_HIP_TRY(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, stream));
    }
    return TSDF_HIP_OK;
  }
  // ---- host relay ----
  const size_t chunk_max = (size_t)32 << 20;
  const size_t need = std::min(bytes, chunk_max);
  if (need > m->relay_cap) {
    if (m->relay) {
      if (m->relay_last_drain) TSDF_HIP_TRY(hipEventSynchronize(m->relay_last_drain));  // the last drain of the old buffer
      TSDF_HIP_TRY(hipHostFree(m->relay));
      m->relay = nullptr, m->relay_cap = 0;
    }
    TSDF_HIP_TRY(hipHostMalloc((void **)&m->relay, need, hipHostMallocPortable));
    m->relay_cap = need;
  }
  auto relay_of = [&](int dev, tsdf_hip_multi::RelayDev **out) -> int {  // the device's relay stream and events, made on first use
    for (auto &rd : m->relay_dev)
      if (rd.dev == dev) {
        *out = &rd;
        return TSDF_HIP_OK;
      }
    TsdfDeviceScope scope(dev);
    TSDF_HIP_TRY(scope.err);
    tsdf_hip_multi::RelayDev rd;
    rd.dev = dev;
    TSDF_HIP_TRY(hipStreamCreateWithFlags(&rd.stream, hipStreamNonBlocking));
    for (hipEvent_t &e : rd.ev) TSDF_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    m->relay_dev.reserve(TSDF_MAX_SLABS + 8);  // (pointers into the vector stay valid)
    m->relay_dev.push_back(rd);
    *out = &m->relay_dev.back();
    return TSDF_HIP_OK;
  };
  tsdf_hip_multi::RelayDev *S = nullptr, *R = nullptr;
  int rc = relay_of(src_dev, &S);
  if (!rc) rc = relay_of(dst_dev, &R);
  if (rc) return rc;
  for (size_t off = 0; off < bytes; off += chunk_max) {
    const size_t n = std::min(chunk_max, bytes - off);
    {
      TsdfDeviceScope scope(dst_dev);
      TSDF_HIP_TRY(scope.err);
      TSDF_HIP_TRY(hipEventRecord(R->ev[0], stream));  // the receiver is ready (and the source data complete: see above)
    }
    {
      TsdfDeviceScope scope(src_dev);
      TSDF_HIP_TRY(scope.err);
      TSDF_HIP_TRY(hipStreamWaitEvent(S->stream, R->ev[0], 0));
      if (m->relay_last_drain) TSDF_HIP_TRY(hipStreamWaitEvent(S->stream, m->relay_last_drain, 0));  // the relay's last content has left it
      TSDF_HIP_TRY(hipMemcpyAsync(m->relay, (const char *)src + off, n, hipMemcpyDeviceToHost, S->stream));
      TSDF_HIP_TRY(hipEventRecord(S->ev[1], S->stream));
    }
    {
      TsdfDeviceScope scope(dst_dev);
      TSDF_HIP_TRY(scope.err);
      TSDF_HIP_TRY(hipStreamWaitEvent(stream, S->ev[1], 0));
      TSDF_HIP_TRY(hipMemcpyAsync((char *)dst + off, m->relay, n, hipMemcpyHostToDevice, stream));
      TSDF_HIP_TRY(hipEventRecord(R->ev[2], stream));
      m->relay_last_drain = R->ev[2];
    }
    m->relay_used = true;
    m->relay_bytes += n;
  }
  return TSDF_HIP_OK;
}

// Device frame (anywhere on the node) -> every slab's staging buffer by peer copy, ordered on the receiving slab's
// stream.  A frame in slab `src`'s own staging buffer (tsdf_hip_organize) is ordered both ways by events: the copies
// wait for that slab's stream, and that slab's later work waits for the copies.  A CALLER's buffer (src_slab < 0) must
// be complete when the call is made and stay untouched until tsdf_hip_synchronize (or any synchronising call): the
// library cannot order a stream it does not know.
static int fan_out_device(tsdf_handle h, const float *d_depth, const uint32_t *d_bgra, int src_slab) {
  tsdf_hip_multi *m = h->multi;
  const size_t npx = (size_t)h->p.image_width * h->p.image_height;
  const bool color = h->p.integrate_color != 0;
  if (color && !d_bgra) {
    tsdf_set_error("integrate_color is set but no colour image was given");
    return TSDF_HIP_E_INVALID;
  }
  int src_dev = -1;
  {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, d_depth) == hipSuccess)
      src_dev = attr.device;
    else
      (void)hipGetLastError();
  }
  if (src_dev < 0) {
    tsdf_set_error("d_depth is not a device pointer");
    return TSDF_HIP_E_INVALID;
  }
  if (src_slab >= 0) {
    TSDF_ON_DEVICE(m->slab[src_slab]->device);
    TSDF_HIP_TRY(hipEventRecord(m->ev[src_slab], m->slab[src_slab]->stream));
  }
  for (size_t k = 0; k < m->slab.size(); ++k) {
    tsdf_handle s = m->slab[k];
    if ((int)k == src_slab) continue;
    TSDF_ON_DEVICE(s->device);
    if (src_slab >= 0) TSDF_HIP_TRY(hipStreamWaitEvent(s->stream, m->ev[src_slab], 0));
    if (d_depth != s->frame_depth) {
      const int rc = tsdf_multi_copy(m, s->frame_depth, s->device, d_depth, src_dev, npx * 4, s->stream);
      if (rc) return rc;
    }
    if (color && d_bgra != s->frame_bgra) {
      const int rc = tsdf_multi_copy(m, s->frame_bgra, s->device, d_bgra, src_dev, npx * 4, s->stream);
      if (rc) return rc;
    }
    // ... and the source after the receivers: whatever the source slab queues next (the next tsdf_hip_organize, the
    // next upload into its staging buffer) must not overwrite the frame while another GPU is still copying it
    if (src_slab >= 0) TSDF_HIP_TRY(hipEventRecord(m->ev2[k], s->stream));
  }
  if (src_slab >= 0) {
    TSDF_ON_DEVICE(m->slab[src_slab]->device);
    for (size_t k = 0; k < m->slab.size(); ++k)
      if ((int)k != src_slab) TSDF_HIP_TRY(hipStreamWaitEvent(m->slab[src_slab]->stream, m->ev2[k], 0));
  }
  return TSDF_HIP_OK;
}

int tsdf_multi_integrate_device(tsdf_handle h, const float *d_depth, const uint32_t *d_bgra, const float T[12],
                                uint64_t *n_observed) {
  if (const int rc_flush = tsdf_multi_flush(h)) return rc_flush;  // (frame pairing: slabs launch what they hold first)
  int rc = fan_out_device(h, d_depth, d_bgra, -1);
  if (rc) return rc;
  return integrate_all(h, T, n_observed);
}

int tsdf_multi_organize(tsdf_handle h, const float *xyz, size_t xyz_stride, const uint8_t *bgra, size_t bgra_stride, size_t n,
                        float cloud_units, int zero_nans, const double world_to_cam[12], float *depth_out, uint8_t *bgra_out,
                        uint64_t *n_valid) {
  // the z-buffer runs on the first slab's GPU (the "ingest GPU"); integrate_staged fans its result out over xGMI
  const int rc = tsdf_hip_organize(h->multi->slab[0], xyz, xyz_stride, bgra, bgra_stride, n, cloud_units, zero_nans, world_to_cam,
                                   depth_out, bgra_out, n_valid);
  if (!rc) h->multi->frame_staged = 1;
  return rc;
}

int tsdf_multi_integrate_staged(tsdf_handle h, const float T[12], uint64_t *n_observed) {
  if (const int rc_flush = tsdf_multi_flush(h)) return rc_flush;  // (frame pairing: slabs launch what they hold first)
  tsdf_hip_multi *m = h->multi;
  if (!m->frame_staged) {
    tsdf_set_error("no staged frame: call tsdf_hip_organize first");
    return TSDF_HIP_E_INVALID;
  }
  tsdf_handle s0 = m->slab[0];
  int rc = fan_out_device(h, s0->frame_depth, s0->p.integrate_color ? s0->frame_bgra : nullptr, 0);
  if (rc) return rc;
  return integrate_all(h, T, n_observed);
}

int tsdf_multi_last_read_detail(tsdf_handle h, uint64_t out[3]) {
  out[0] = h->last_implied;
  out[1] = h->last_implied_on ? 1 : 0;
  out[2] = h->last_read_bytes;
  return TSDF_HIP_OK;
}

int tsdf_multi_last_count_detail(tsdf_handle h, uint64_t out[2]) {
  out[0] = h->last_observed;
  out[1] = h->last_changed_bytes;
  return TSDF_HIP_OK;
}

// ---- halo exchange ---------------------------------------------------------------------------------------------------
// Copy global planes [z0, z0 + nz) from the slab that owns them into `dst`'s halo: the slabs share pitch and layout,
// so a plane is one contiguous run per array.  Ordered after the owner's stream, queued on the receiver's.
static int copy_planes(tsdf_hip_multi *m, int src, int dst, int z0, int nz) {
  tsdf_handle a = m->slab[src], b = m->slab[dst];
  const int64_t plane = a->pitch * a->ny;
  const int64_t oa = (int64_t)(z0 - a->z_first) * plane, ob = (int64_t)(z0 - b->z_first) * plane;
  {
    TSDF_ON_DEVICE(a->device);
    TSDF_HIP_TRY(hipEventRecord(m->ev[src], a->stream));
  }
  TSDF_ON_DEVICE(b->device);
  TSDF_HIP_TRY(hipStreamWaitEvent(b->stream, m->ev[src], 0));
  auto cp = [&](void *dst_p, const void *src_p, size_t bytes) -> int {
    return tsdf_multi_copy(m, dst_p, b->device, src_p, a->device, bytes, b->stream);
  };
  const size_t n = (size_t)(plane * nz);
  int rc = cp(b->d + ob, a->d + oa, n * 4);
  if (!rc && a->w) rc = cp(b->w + ob, a->w + oa, n * 4);
  if (!rc && a->rgb) rc = cp(b->rgb + ob, a->rgb + oa, n * 4);
  if (!rc && a->k8) rc = cp(b->k8 + ob, a->k8 + oa, n);
  for (int c = 0; c < 4 && !rc; ++c)
    if (a->cn[c]) rc = cp(b->cn[c] + ob, a->cn[c] + oa, n * 4);
  return rc;
}

// Every slab's stream waits for what every other slab has queued so far.
static int all_wait_all(tsdf_hip_multi *m) {
  const int n = (int)m->slab.size();
  for (int k = 0; k < n; ++k) {
    TSDF_ON_DEVICE(m->slab[k]->device);
    TSDF_HIP_TRY(hipEventRecord(m->ev2[k], m->slab[k]->stream));
  }
  for (int k = 0; k < n; ++k) {
    TSDF_ON_DEVICE(m->slab[k]->device);
    for (int j = 0; j < n; ++j)
      if (j != k) TSDF_HIP_TRY(hipStreamWaitEvent(m->slab[k]->stream, m->ev2[j], 0));
  }
  return TSDF_HIP_OK;
}

// Refresh `planes` halo planes above every slab (and below, with `both`) from their owners.
static int exchange_halo(tsdf_handle h, int planes, bool both) {
  tsdf_hip_multi *m = h->multi;
  const int n = (int)m->slab.size();
  if (n == 1) return TSDF_HIP_OK;
  planes = std::min(planes, m->halo);
  if (both ? m->halo_all_fresh : (m->halo1_fresh && planes <= 1)) return TSDF_HIP_OK;
  for (int k = 0; k < n; ++k) {
    tsdf_handle s = m->slab[k];
    const int ranges[2][2] = {{s->z_end, std::min(h->nz, s->z_end + planes)},
                              {both ? std::max(0, s->z_begin - planes) : s->z_begin, s->z_begin}};
    for (int r = 0; r < 2; ++r)
      for (int z = ranges[r][0]; z < ranges[r][1];) {
        const int o = owner_of(m, z);
        if (o < 0) return TSDF_HIP_E_INVALID;
        const int run = std::min(ranges[r][1], m->slab[o]->z_end) - z;
        const int rc = copy_planes(m, o, k, z, run);
        if (rc) return rc;
        z += run;
      }
  }
  // the owners must not run ahead and change planes that are still being read: receivers' copies are ordered after
  // the owners' streams above; order the owners' NEXT work after the copies
  if (int rc = all_wait_all(m)) return rc;
  m->halo1_fresh = true;
  if (both && planes >= m->halo) m->halo_all_fresh = true;
  return TSDF_HIP_OK;
}

// ---- raw voxel blocks --------------------------------------------------------------------------------------------------
int tsdf_multi_block(tsdf_handle h, bool down, int x0, int y0, int z0, int nx, int ny, int nz, float *d, float *w, uint8_t *rgb) {
  if (const int rc_flush = tsdf_multi_flush(h)) return rc_flush;  // (frame pairing: slabs launch what they hold first)
  tsdf_hip_multi *m = h->multi;
  if (nx <= 0 || ny <= 0 || nz <= 0 || z0 < 0 || z0 + nz > h->nz) {
    tsdf_set_error("block outside the grid");
    return TSDF_HIP_E_INVALID;
  }
  if (!down) m->halo1_fresh = m->halo_all_fresh = false;
  const size_t per_plane = (size_t)nx * ny;
  for (int z = z0; z < z0 + nz;) {
    const int o = owner_of(m, z);
    const int run = std::min(z0 + nz, m->slab[o]->z_end) - z;
    const size_t off = (size_t)(z - z0) * per_plane;
    const int rc = down ? tsdf_hip_download(m->slab[o], x0, y0, z, nx, ny, run, d ? d + off : nullptr, w ? w + off : nullptr,
                                            rgb ? rgb + 3 * off : nullptr)
                        : tsdf_hip_upload(m->slab[o], x0, y0, z, nx, ny, run, d ? d + off : nullptr, w ? w + off : nullptr,
                                          rgb ? rgb + 3 * off : nullptr);
    if (rc) return rc;
    z += run;
  }
  return TSDF_HIP_OK;
}

int tsdf_multi_variance_block(tsdf_handle h, bool down, int x0, int y0, int z0, int nx, int ny, int nz, float *M, int32_t *ns) {
  if (const int rc_flush = tsdf_multi_flush(h)) return rc_flush;  // (frame pairing: slabs launch what they hold first)
  tsdf_hip_multi *m = h->multi;
  if (nx <= 0 || ny <= 0 || nz <= 0 || z0 < 0 || z0 + nz > h->nz) {
    tsdf_set_error("block outside the grid");
    return TSDF_HIP_E_INVALID;
  }
  const size_t per_plane = (size_t)nx * ny;
  for (int z = z0; z < z0 + nz;) {
    const int o = owner_of(m, z);
    const int run = std::min(z0 + nz, m->slab[o]->z_end) - z;
    const size_t off = (size_t)(z - z0) * per_plane;
    const int rc = down ? tsdf_hip_download_variance_state(m->slab[o], x0, y0, z, nx, ny, run, M ? M + off : nullptr, ns ? ns + off : nullptr)
                        : tsdf_hip_upload_variance_state(m->slab[o], x0, y0, z, nx, ny, run, M ? M + off : nullptr, ns ? ns + off : nullptr);
    if (rc) return rc;
    z += run;
  }
  return TSDF_HIP_OK;
}

// ---- getFxn / getGradient / getHessian, renderColoredView's lookup -------------------------------------------------------
int tsdf_multi_sample(tsdf_handle h, const float *xyz, size_t n, float *val, float *grad, float *hess, uint8_t *ok) {
  if (const int rc_flush = tsdf_multi_flush(h)) return rc_flush;  // (frame pairing: slabs launch what they hold first)
  tsdf_hip_multi *m = h->multi;
  int rc = exchange_halo(h, 1, false);
  if (rc) return rc;
  std::vector<float> v(n), g(grad ? 3 * n : 0), hs(hess ? 9 * n : 0);
  std::vector<uint8_t> o(n);
  std::vector<uint8_t> done(n, 0);
  for (size_t k = 0; k < m->slab.size(); ++k) {
    rc = tsdf_hip_sample(m->slab[k], xyz, n, v.data(), grad ? g.data() : nullptr, hess ? hs.data() : nullptr, o.data());
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) {
      // exactly one slab owns a point's lower-corner plane; a point no slab answers keeps the first slab's NaNs
      if (!(o[i] || (k == 0))) continue;
      if (done[i]) continue;
      if (val) val[i] = v[i];
      if (grad) memcpy(grad + 3 * i, g.data() + 3 * i, 12);
      if (hess) memcpy(hess + 9 * i, hs.data() + 9 * i, 36);
      if (ok) ok[i] = o[i];
      if (o[i]) done[i] = 1;
    }
  }
  return TSDF_HIP_OK;
}

int tsdf_multi_lookup_rgb(tsdf_handle h, const float *xyz, size_t n, uint8_t *rgb, uint8_t *found) {
  if (const int rc_flush = tsdf_multi_flush(h)) return rc_flush;  // (frame pairing: slabs launch what they hold first)
  tsdf_hip_multi *m = h->multi;
  std::vector<uint8_t> c(3 * n), f(n);
  memset(rgb, 0, 3 * n);
  memset(found, 0, n);
  for (tsdf_handle s : m->slab) {
    const int rc = tsdf_hip_lookup_rgb(s, xyz, n, c.data(), f.data());
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i)
      if (f[i] && !found[i]) {
        found[i] = 1;
        memcpy(rgb + 3 * i, c.data() + 3 * i, 3);
      }
  }
  return TSDF_HIP_OK;
}

// ---- renderView: ray hand-off between the slabs, compact lists -----------------------------------------------------------
// Every slab holds only the rays it is responsible for (at the start: pixel index = slab (mod n_slab); later: the
// owner of the plane the ray's loop needs next).  A round = every slab, concurrently on its own stream: advance its
// list (k_raycast<true>), sort the records by destination (tsdf_ray_list_route), copy the n_slab + 2 counters to a
// pinned table.  The host waits for the tables (one event per slab, the slabs having run in parallel), then queues
// the point-to-point copies: suspended records to the owner of their next plane (96 B each), finished rays to the
// first slab (36 B each), where k_ray_deliver writes them into the image and applies :422.  No image-sized buffer
// crosses a link and no stream is synchronised inside a round.
int tsdf_multi_raycast(tsdf_handle h, const float rot[9], const float origin[3], int downsample, const double *inv, float *out) {
  if (const int rc_flush = tsdf_multi_flush(h)) return rc_flush;  // (frame pairing: slabs launch what they hold first)
  tsdf_hip_multi *m = h->multi;
  const int n_slab = (int)m->slab.size();
  if (downsample < 1) return TSDF_HIP_E_INVALID;
  const int nw = h->p.image_width / downsample, nh = h->p.image_height / downsample;
  const int64_t n = (int64_t)nw * nh;
  if (n <= 0 || n >= (1ll << 31)) return TSDF_HIP_E_INVALID;
  int rc = exchange_halo(h, m->halo, true);
  if (rc) return rc;
  const int cols = n_slab + 2;  // table row of a slab: records per destination slab, finished, incomplete
  if ((size_t)n > m->ray_cap) {
    if ((rc = tsdf_multi_synchronize(h))) return rc;
    for (int k = 0; k < n_slab; ++k) {
      TSDF_ON_DEVICE(m->slab[k]->device);
      for (int **p : {&m->ray_list[k], &m->ray_outbox[k], &m->ray_finbox[k]}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
      }
      TSDF_HIP_TRY(hipMalloc(&m->ray_list[k], (size_t)n * TSDF_HIP_RAY_RECORD_INTS * sizeof(int)));
      TSDF_HIP_TRY(hipMalloc(&m->ray_outbox[k], (size_t)n * TSDF_HIP_RAY_RECORD_INTS * sizeof(int)));
      TSDF_HIP_TRY(hipMalloc(&m->ray_finbox[k], (size_t)n * TSDF_RAY_FIN_INTS * sizeof(int)));
      if (!m->ray_counters[k]) TSDF_HIP_TRY(hipMalloc(&m->ray_counters[k], (2 * (size_t)(n_slab + 1) + 1) * sizeof(unsigned)));
    }
    TSDF_ON_DEVICE(m->slab[0]->device);
    if (m->ray_fin_in) (void)hipFree(m->ray_fin_in);
    if (m->ray_image) (void)hipFree(m->ray_image);
    m->ray_fin_in = nullptr, m->ray_image = nullptr;
    TSDF_HIP_TRY(hipMalloc(&m->ray_fin_in, (size_t)n * TSDF_RAY_FIN_INTS * sizeof(int)));
    TSDF_HIP_TRY(hipMalloc(&m->ray_image, (size_t)n * 8 * sizeof(float)));
    if (!m->ray_table) TSDF_HIP_TRY(hipHostMalloc((void **)&m->ray_table, (size_t)n_slab * cols * sizeof(unsigned), hipHostMallocPortable));
    m->ray_cap = (size_t)n;
  }
  std::vector<int> z_end(n_slab);
  for (int k = 0; k < n_slab; ++k) z_end[k] = m->slab[k]->z_end;
  std::vector<unsigned> cnt(n_slab, 0), next(n_slab, 0);
  uint64_t rounds = 0, handed = 0, bytes_between = 0, host_waits = 0;
  for (int k = 0; k < n_slab; ++k) {
    tsdf_handle s = m->slab[k];
    TSDF_ON_DEVICE(s->device);
    unsigned *inc = m->ray_counters[k] + 2 * (n_slab + 1);
    TSDF_HIP_TRY(hipMemsetAsync(inc, 0, sizeof(unsigned), s->stream));
    if ((rc = tsdf_ray_list_begin(s, rot, origin, downsample, k, n_slab, m->ray_list[k], &cnt[k]))) return rc;
  }
  bool done = false;
  for (int round = 0; !done && round < 2 * n_slab + 4; ++round, ++rounds) {
    for (int k = 0; k < n_slab; ++k) {  // every slab: advance, sort by destination, report the counts
      tsdf_handle s = m->slab[k];
      TSDF_ON_DEVICE(s->device);
      unsigned *ctr = m->ray_counters[k], *inc = ctr + 2 * (n_slab + 1);
      if ((rc = tsdf_ray_list_advance(s, rot, origin, downsample, k, n_slab, m->ray_list[k], cnt[k], inc))) return rc;
      if ((rc = tsdf_ray_list_route(s, m->ray_list[k], cnt[k], n_slab, z_end.data(), ctr, m->ray_outbox[k], m->ray_finbox[k]))) return rc;
      unsigned *row = m->ray_table + (size_t)k * cols;
      TSDF_HIP_TRY(hipMemcpyAsync(row, ctr, (size_t)(n_slab + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
      TSDF_HIP_TRY(hipMemcpyAsync(row + n_slab + 1, inc, sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
      TSDF_HIP_TRY(hipEventRecord(m->ev[k], s->stream));
    }
    for (int k = 0; k < n_slab; ++k) {  // (the slabs ran concurrently: this waits for the slowest, once)
      TSDF_HIP_TRY(hipEventSynchronize(m->ev[k]));
      ++host_waits;
      if (m->ray_table[(size_t)k * cols + n_slab + 1]) {
        tsdf_set_error("ray hand-off: the refinement walk / trilinear samples left a slab's halo planes "
                       "(tsdf_hip_render_halo too small for this volume)");
        return TSDF_HIP_E_UNSUPPORTED;
      }
    }
    auto table = [&](int src, int dst) { return m->ray_table[(size_t)src * cols + dst]; };
    // suspended records: from every slab's outbox (segments in destination order) into the owner's list
    uint64_t still = 0;
    for (int d = 0; d < n_slab; ++d) {
      tsdf_handle sd = m->slab[d];
      TSDF_ON_DEVICE(sd->device);
      unsigned at = 0;
      for (int k = 0; k < n_slab; ++k) {
        const unsigned c = table(k, d);
        if (!c) continue;
        unsigned first = 0;
        for (int e = 0; e < d; ++e) first += table(k, e);
        if (k != d) TSDF_HIP_TRY(hipStreamWaitEvent(sd->stream, m->ev[k], 0));
        const size_t rec = TSDF_HIP_RAY_RECORD_INTS * sizeof(int);
        if ((rc = tsdf_multi_copy(m, m->ray_list[d] + (size_t)at * TSDF_HIP_RAY_RECORD_INTS, sd->device,
                                  m->ray_outbox[k] + (size_t)first * TSDF_HIP_RAY_RECORD_INTS, m->slab[k]->device, c * rec, sd->stream, k != d)))
          return rc;
        if (k != d) handed += c, bytes_between += (uint64_t)c * rec;
        at += c;
      }
      next[d] = at;
      still += at;
    }
    // finished rays: to the first slab, into the image
    {
      tsdf_handle s0 = m->slab[0];
      TSDF_ON_DEVICE(s0->device);
      unsigned at = 0;
      for (int k = 0; k < n_slab; ++k) {
        const unsigned c = table(k, n_slab);
        if (!c) continue;
        if (k != 0) TSDF_HIP_TRY(hipStreamWaitEvent(s0->stream, m->ev[k], 0));
        const size_t rec = TSDF_RAY_FIN_INTS * sizeof(int);
        if ((rc = tsdf_multi_copy(m, m->ray_fin_in + (size_t)at * TSDF_RAY_FIN_INTS, s0->device, m->ray_finbox[k], m->slab[k]->device,
                                  c * rec, s0->stream, k != 0)))
          return rc;
        if (k != 0) bytes_between += (uint64_t)c * rec;
        at += c;
      }
      if ((rc = tsdf_ray_deliver(s0, m->ray_fin_in, at, m->ray_image, n, inv))) return rc;
    }
    // a slab's boxes are rewritten by its next round: that must follow the copies the other slabs just queued
    if ((rc = all_wait_all(m))) return rc;
    cnt.swap(next);
    done = still == 0;
  }
  m->rv_stats[0] = rounds, m->rv_stats[1] = handed, m->rv_stats[2] = bytes_between, m->rv_stats[3] = host_waits;
  if (!done) {
    tsdf_set_error("ray hand-off did not converge (rays still suspended after 2 * slabs + 4 rounds)");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  return tsdf_to_host(m->slab[0], out, m->ray_image, (size_t)n * 8 * sizeof(float));  // (synchronises the first slab's stream)
}

// Report-only: rounds, records handed from one slab to another, bytes that moved between slabs (hand-offs + finished rays
// to the first slab) and host waits of the last renderView on this multi handle.
extern "C" int tsdf_hip_multi_render_stats(tsdf_handle h, uint64_t out[4]) {
  if (!h || !h->multi || !out) return TSDF_HIP_E_INVALID;
  for (int i = 0; i < 4; ++i) out[i] = h->multi->rv_stats[i];
  return TSDF_HIP_OK;
}

// Report-only: how the slabs of this set reach each other -- out[0] = device pairs whose peer access the driver refused,
// out[1] = 1 if every cross-slab copy takes the host relay (TSDF_HIP_NO_PEER=1), out[2] = bytes that went through the
// relay since create.
extern "C" int tsdf_hip_multi_link_stats(tsdf_handle h, uint64_t out[3]) {
  if (!h || !h->multi || !out) return TSDF_HIP_E_INVALID;
  out[0] = h->multi->no_peer.size();
  out[1] = h->multi->relay_all ? 1 : 0;
  out[2] = h->multi->relay_bytes;
  return TSDF_HIP_OK;
}

// ---- marching cubes ------------------------------------------------------------------------------------------------------
static inline uint64_t spread3_host(uint64_t v) {
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}
static inline uint64_t morton_of_cell(uint64_t c) {  // cell = x<<42 | y<<21 | z; the reference's order: x is the high bit
  return (spread3_host(c >> 42) << 2) | (spread3_host((c >> 21) & 0x1fffff) << 1) | spread3_host(c & 0x1fffff);
}

int tsdf_multi_march(tsdf_handle h, float w_min, int color_mode, uint64_t *n_tri) {
  if (const int rc_flush = tsdf_multi_flush(h)) return rc_flush;  // (frame pairing: slabs launch what they hold first)
  tsdf_hip_multi *m = h->multi;
  const int n = (int)m->slab.size();
  if (n_tri) *n_tri = 0;
  int rc = exchange_halo(h, 1, false);
  if (rc) return rc;
  if ((rc = tsdf_multi_synchronize(h))) return rc;
  struct Part {
    std::vector<float> verts;
    std::vector<uint8_t> rgb;
    std::vector<uint64_t> cell;
    uint64_t ntri = 0;
    int rc = 0;
    std::string err;
    float ms[3] = {0, 0, 0};
    uint64_t cells = 0;
  };
  std::vector<Part> part(n);
  std::vector<std::thread> th;
  for (int k = 0; k < n; ++k)
    th.emplace_back([&, k]() {  // one host thread per slab: the slabs' kernels and downloads overlap
      Part &p = part[k];
      tsdf_handle s = m->slab[k];
      p.rc = tsdf_hip_march(s, w_min, color_mode, &p.ntri);
      if (!p.rc && p.ntri) {
        p.verts.resize(p.ntri * 9);
        if (color_mode) p.rgb.resize(p.ntri * 9);
        p.cell.resize(p.ntri);
        p.rc = tsdf_hip_march_fetch(s, p.verts.data(), color_mode ? p.rgb.data() : nullptr, p.cell.data());
      }
      if (p.rc) p.err = tsdf_hip_last_error();
      (void)tsdf_hip_march_timing(s, p.ms, &p.cells);
    });
  for (auto &t : th) t.join();
  uint64_t total = 0;
  m->mc_ms[0] = m->mc_ms[1] = m->mc_ms[2] = 0.f;
  m->mc_ncells = 0;
  for (int k = 0; k < n; ++k) {
    if (part[k].rc) {
      tsdf_set_error(part[k].err);
      return part[k].rc;
    }
    total += part[k].ntri;
    m->mc_ncells += part[k].cells;
    for (int i = 0; i < 3; ++i) m->mc_ms[i] = std::max(m->mc_ms[i], part[k].ms[i]);
  }
  // k-way merge by Morton key: every part is sorted, the triangles of one cell are adjacent and stay in order, and no
  // key occurs in two parts (a cell belongs to the slab of its base voxel)
  m->verts.resize(total * 9);
  m->rgb.resize(color_mode ? total * 9 : 0);
  m->cell.resize(total);
  m->mesh_has_rgb = color_mode != 0;
  std::vector<uint64_t> pos(n, 0), key(n, ~0ull);
  for (int k = 0; k < n; ++k)
    if (part[k].ntri) key[k] = morton_of_cell(part[k].cell[0]);
  for (uint64_t t = 0; t < total;) {
    int best = 0;
    for (int k = 1; k < n; ++k)
      if (key[k] < key[best]) best = k;
    Part &p = part[best];
    uint64_t i = pos[best], j = i;
    // take the run of this part up to the smallest key of the other parts
    uint64_t limit = ~0ull;
    for (int k = 0; k < n; ++k)
      if (k != best) limit = std::min(limit, key[k]);
    while (j < p.ntri && morton_of_cell(p.cell[j]) < limit) ++j;
    if (j == i) j = i + 1;  // (cannot happen: keys are unique across parts)
    const uint64_t cnt = j - i;
    memcpy(&m->verts[t * 9], &p.verts[i * 9], cnt * 9 * sizeof(float));
    if (color_mode) memcpy(&m->rgb[t * 9], &p.rgb[i * 9], cnt * 9);
    memcpy(&m->cell[t], &p.cell[i], cnt * sizeof(uint64_t));
    t += cnt;
    pos[best] = j;
    key[best] = j < p.ntri ? morton_of_cell(p.cell[j]) : ~0ull;
  }
  h->mc_ntri = total;
  h->mc_has_rgb = color_mode != 0;
  if (n_tri) *n_tri = total;
  return TSDF_HIP_OK;
}

int tsdf_multi_march_fetch(tsdf_handle h, float *verts, uint8_t *rgb, uint64_t *cell) {
  tsdf_hip_multi *m = h->multi;
  const size_t n = (size_t)h->mc_ntri;
  if (!n) return TSDF_HIP_OK;
  if (rgb && !m->mesh_has_rgb) {
    tsdf_set_error("the last tsdf_hip_march ran without a colour mode");
    return TSDF_HIP_E_INVALID;
  }
  if (verts) memcpy(verts, m->verts.data(), n * 9 * sizeof(float));
  if (rgb) memcpy(rgb, m->rgb.data(), n * 9);
  if (cell) memcpy(cell, m->cell.data(), n * sizeof(uint64_t));
  return TSDF_HIP_OK;
}

int tsdf_multi_march_timing(tsdf_handle h, float ms[3], uint64_t *n_cells) {
  for (int i = 0; i < 3; ++i) ms[i] = h->multi->mc_ms[i];
  if (n_cells) *n_cells = h->multi->mc_ncells;
  return TSDF_HIP_OK;
}

int tsdf_multi_march_stats(tsdf_handle h, uint64_t out[4]) {  // sums over the slabs; "skipped" only if every slab did
  out[0] = out[1] = out[2] = 0, out[3] = 1;
  for (tsdf_handle s : h->multi->slab) {
    uint64_t o[4];
    const int rc = tsdf_hip_march_stats(s, o);
    if (rc) return rc;
    out[0] += o[0], out[1] += o[1], out[2] += o[2], out[3] &= o[3];
  }
  return TSDF_HIP_OK;
}

tsdf_handle tsdf_multi_first(tsdf_handle h) { return h->multi->slab[0]; }
