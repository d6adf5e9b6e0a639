// tsdf_hip_save / tsdf_hip_load (+ the callback forms): the reference's .vol checkpoint
// (src/lib/tsdf_volume_octree.cpp:222-275) over the C ABI.  Host code only; the format lives in
// vol_format.h, the voxels move one cubic block at a time through tsdf_hip_download / tsdf_hip_upload or
// the caller's callbacks.
#include "tsdf_common.h"
#include "vol_format.h"

using cpu_tsdf::VolHeader;

static void default_meta(const tsdf_params &p, tsdf_vol_meta *m) {
  for (int k = 0; k < 3; ++k) m->max_cell_size[k] = p.size[k] / (float)p.res[k];
  m->is_empty = 0;
  m->weight_by_depth = m->weight_by_variance = 0;
  for (int i = 0; i < 16; ++i) m->global_transform[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

static void header_from(const tsdf_params &p, const tsdf_vol_meta &m, VolHeader &hd) {
  for (int k = 0; k < 3; ++k) {
    hd.res[k] = p.res[k];
    hd.size[k] = p.size[k];
    hd.max_cell[k] = m.max_cell_size[k];
  }
  hd.max_dist_pos = p.max_dist_pos;
  hd.max_dist_neg = p.max_dist_neg;
  hd.max_weight = p.max_weight;
  hd.min_sensor_dist = p.min_sensor_dist;
  hd.max_sensor_dist = p.max_sensor_dist;
  hd.fx = p.fx;
  hd.fy = p.fy;
  hd.cx = p.cx;
  hd.cy = p.cy;
  hd.image_width = p.image_width;
  hd.image_height = p.image_height;
  hd.is_empty = m.is_empty != 0;
  hd.weight_by_depth = m.weight_by_depth != 0;
  hd.weight_by_variance = m.weight_by_variance != 0;
  for (int i = 0; i < 16; ++i) hd.global_transform[i] = m.global_transform[i];
  hd.color = p.integrate_color != 0;
}

static void header_to(const VolHeader &f, tsdf_params &p, tsdf_vol_meta &m) {
  for (int k = 0; k < 3; ++k) {
    p.res[k] = f.res[k];
    p.size[k] = f.size[k];
    m.max_cell_size[k] = f.max_cell[k];
  }
  p.max_dist_pos = f.max_dist_pos;
  p.max_dist_neg = f.max_dist_neg;
  p.max_weight = f.max_weight;
  p.min_sensor_dist = f.min_sensor_dist;
  p.max_sensor_dist = f.max_sensor_dist;
  p.fx = f.fx;
  p.fy = f.fy;
  p.cx = f.cx;
  p.cy = f.cy;
  p.image_width = f.image_width;
  p.image_height = f.image_height;
  p.integrate_color = f.color ? 1 : 0;
  p.z_begin = p.z_end = p.halo = 0;
  m.is_empty = f.is_empty;
  m.weight_by_depth = f.weight_by_depth;
  m.weight_by_variance = f.weight_by_variance;
  for (int i = 0; i < 16; ++i) m.global_transform[i] = f.global_transform[i];
}

// the optional M_ / nsample_ side channel of a block (vol_format.h VarFn), C style like tsdf_block_fn
typedef int (*tsdf_var_fn)(void *user, int x0, int y0, int z0, int edge, float *M, int32_t *nsample);

static int save_blocks_impl(const tsdf_params *p, const tsdf_vol_meta *meta, const char *filename, tsdf_block_fn fetch,
                            tsdf_var_fn var, void *user) {
  if (!p || !filename || !fetch) return TSDF_HIP_E_INVALID;
  if (cpu_tsdf::volfmt::log2_exact(p->res[0]) < 0 || p->res[1] != p->res[0] || p->res[2] != p->res[0]) {
    tsdf_set_error("the .vol octree format needs a cubic power-of-two resolution");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  tsdf_vol_meta m;
  if (meta)
    m = *meta;
  else
    default_meta(*p, &m);
  VolHeader hd;
  header_from(*p, m, hd);
  int rc = 0;
  std::string err;
  const bool ok = cpu_tsdf::vol_write_stream(
      filename, hd, tsdf_vol_chunk(),
      [&](int x0, int y0, int z0, int c, float *d, float *w, unsigned char *rgb) {
        rc = fetch(user, x0, y0, z0, c, d, w, rgb);
        return rc == 0;
      },
      &err,
      var ? cpu_tsdf::volfmt::VarFn([&](int x0, int y0, int z0, int c, float *M, int32_t *ns) {
        rc = var(user, x0, y0, z0, c, M, ns);
        return rc == 0;
      })
          : cpu_tsdf::volfmt::VarFn());
  if (ok) return TSDF_HIP_OK;
  if (rc) return rc;  // (the callback's own code; its message, if any, is already set)
  tsdf_set_error(err);
  return TSDF_HIP_E_IO;
}

extern "C" int tsdf_hip_save_blocks(const tsdf_params *p, const tsdf_vol_meta *meta, const char *filename,
                                    tsdf_block_fn fetch, void *user) {
  return save_blocks_impl(p, meta, filename, fetch, nullptr, user);
}

static int load_blocks_impl(const char *filename, const tsdf_params *defaults, tsdf_header_fn on_header, tsdf_block_fn store,
                            tsdf_var_fn var, void *user) {
  if (!filename || (!on_header && !store)) return TSDF_HIP_E_INVALID;
  tsdf_params p;
  if (defaults)
    p = *defaults;
  else
    tsdf_hip_default_params(&p);
  tsdf_vol_meta m;
  VolHeader hd;
  int rc = 0;
  bool header_only = false;
  std::string err;
  const bool ok = cpu_tsdf::vol_read_stream(
      filename, hd, tsdf_vol_chunk(),
      [&](const VolHeader &f) {
        header_to(f, p, m);
        if (on_header) rc = on_header(user, &p, &m);
        header_only = rc == 0 && !store;
        return rc == 0 && store != nullptr;
      },
      [&](int x0, int y0, int z0, int c, float *d, float *w, unsigned char *rgb) {
        rc = store(user, x0, y0, z0, c, d, w, rgb);
        return rc == 0;
      },
      &err,
      var ? cpu_tsdf::volfmt::VarFn([&](int x0, int y0, int z0, int c, float *M, int32_t *ns) {
        rc = var(user, x0, y0, z0, c, M, ns);
        return rc == 0;
      })
          : cpu_tsdf::volfmt::VarFn());
  if (ok || header_only) return TSDF_HIP_OK;
  if (rc) return rc;
  tsdf_set_error(err);
  return TSDF_HIP_E_IO;
}

extern "C" int tsdf_hip_load_blocks(const char *filename, const tsdf_params *defaults, tsdf_header_fn on_header,
                                    tsdf_block_fn store, void *user) {
  return load_blocks_impl(filename, defaults, on_header, store, nullptr, user);
}

// ---- one handle ---------------------------------------------------------------------------------------------
static int fetch_from_handle(void *user, int x0, int y0, int z0, int c, float *d, float *w, uint8_t *rgb) {
  return tsdf_hip_download((tsdf_handle)user, x0, y0, z0, c, c, c, d, w, rgb);
}

static int fetch_var_from_handle(void *user, int x0, int y0, int z0, int c, float *M, int32_t *ns) {
  return tsdf_hip_download_variance_state((tsdf_handle)user, x0, y0, z0, c, c, c, M, ns);
}

extern "C" int tsdf_hip_save(tsdf_handle h, const char *filename, const tsdf_vol_meta *meta) {
  if (!h || !filename) return TSDF_HIP_E_INVALID;
  if (h->z_begin != 0 || h->z_end != h->p.res[2]) {
    tsdf_set_error("save needs a handle that owns the whole grid (Z-slabs: tsdf_hip_save_blocks)");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  if ((h->multi ? tsdf_multi_first(h) : h)->cn[0]) {
    tsdf_set_error("RGB_NORMALIZED / LAB volumes have no usable .vol form (the reference writes one byte of each float, "
                   "octree.cpp:417-433)");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  // a volume that weights by variance carries its M_ / nsample_ state in the file, like the reference's octree nodes
  const bool var = (h->multi ? tsdf_multi_first(h) : h)->weight_by_variance != 0;
  return save_blocks_impl(&h->p, meta, filename, fetch_from_handle, var ? fetch_var_from_handle : nullptr, h);
}

namespace {
struct LoadState {
  tsdf_handle h = nullptr;
  tsdf_params p;
  tsdf_vol_meta m;
  bool force_f32w = false;
  const int32_t *devices = nullptr;  // non-null: build a multi-GPU set (tsdf_hip_create_multi)
  int n_devices = 0;
};
int load_header(void *user, const tsdf_params *p, const tsdf_vol_meta *m) {
  LoadState *s = (LoadState *)user;
  s->p = *p;
  s->m = *m;
  if (s->force_f32w) s->p.layout = TSDF_LAYOUT_F32W;
  // a depth-weighted volume (hpp:201-202) holds weights that are not counts, and keeps integrating that way
  if ((m->weight_by_depth || m->weight_by_variance) && s->p.layout == TSDF_LAYOUT_AUTO) s->p.layout = TSDF_LAYOUT_F32W;
  const int rc = s->devices ? tsdf_hip_create_multi(&s->p, s->devices, s->n_devices, &s->h) : tsdf_hip_create(&s->p, &s->h);
  if (rc) return rc;
  return tsdf_hip_set_weighting(s->h, m->weight_by_depth, m->weight_by_variance);
}
int load_store(void *user, int x0, int y0, int z0, int c, float *d, float *w, uint8_t *rgb) {
  return tsdf_hip_upload(((LoadState *)user)->h, x0, y0, z0, c, c, c, d, w, rgb);
}
int load_store_var(void *user, int x0, int y0, int z0, int c, float *M, int32_t *ns) {  // (only called for files that weight by variance)
  return tsdf_hip_upload_variance_state(((LoadState *)user)->h, x0, y0, z0, c, c, c, M, ns);
}
}  // namespace

static int load_impl(const char *filename, const tsdf_params *defaults, const int32_t *devices, int n_devices, tsdf_handle *out,
                     tsdf_params *params_out, tsdf_vol_meta *meta_out) {
  if (!filename || !out) return TSDF_HIP_E_INVALID;
  *out = nullptr;
  const int asked = defaults ? defaults->layout : TSDF_LAYOUT_AUTO;
  for (int attempt = 0; attempt < 2; ++attempt) {
    LoadState s;
    s.force_f32w = attempt == 1;
    s.devices = devices;
    s.n_devices = n_devices;
    const int rc = load_blocks_impl(filename, defaults, load_header, load_store, load_store_var, &s);
    if (rc == TSDF_HIP_OK) {
      *out = s.h;
      if (params_out) {
        *params_out = s.p;
        params_out->layout = tsdf_hip_layout(s.h);
      }
      if (meta_out) *meta_out = s.m;
      return TSDF_HIP_OK;
    }
    // weights that are not min(k, max_weight) (a file written with other weighting) do not fit the packed
    // layout: with AUTO the file is read again into a float weight plane
    const bool packed_misfit = rc == TSDF_HIP_E_UNSUPPORTED && s.h && tsdf_hip_layout(s.h) == TSDF_LAYOUT_PACKED;
    if (s.h) tsdf_hip_destroy(s.h);
    if (!(packed_misfit && attempt == 0 && asked == TSDF_LAYOUT_AUTO)) return rc;
  }
  return TSDF_HIP_E_UNSUPPORTED;
}

extern "C" int tsdf_hip_load(const char *filename, const tsdf_params *defaults, tsdf_handle *out,
                             tsdf_params *params_out, tsdf_vol_meta *meta_out) {
  return load_impl(filename, defaults, nullptr, 0, out, params_out, meta_out);
}

extern "C" int tsdf_hip_load_multi(const char *filename, const tsdf_params *defaults, const int32_t *devices, int n_devices,
                                   tsdf_handle *out, tsdf_params *params_out, tsdf_vol_meta *meta_out) {
  if (!devices || n_devices < 1) return TSDF_HIP_E_INVALID;
  return load_impl(filename, defaults, devices, n_devices, out, params_out, meta_out);
}
