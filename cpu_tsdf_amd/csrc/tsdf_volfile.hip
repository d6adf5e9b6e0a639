// tsdf_hip_save / tsdf_hip_load: the reference's .vol checkpoint (src/lib/tsdf_volume_octree.cpp:222-275)
// over the C ABI.  Host code only; the format lives in vol_format.h, the voxels move through
// tsdf_hip_download / tsdf_hip_upload one cubic block at a time.
#include "tsdf_common.h"
#include "vol_format.h"

using cpu_tsdf::VolHeader;

static void default_meta(const tsdf_params &p, tsdf_vol_meta *m) {
  for (int k = 0; k < 3; ++k) m->max_cell_size[k] = p.size[k] / (float)p.res[k];
  m->is_empty = 0;
  m->weight_by_depth = m->weight_by_variance = 0;
  for (int i = 0; i < 16; ++i) m->global_transform[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

extern "C" int tsdf_hip_save(tsdf_handle h, const char *filename, const tsdf_vol_meta *meta) {
  if (!h || !filename) return TSDF_HIP_E_INVALID;
  const tsdf_params &p = h->p;
  if (h->z_begin != 0 || h->z_end != p.res[2]) {
    tsdf_set_error("save needs a handle that owns the whole grid (gather the Z-slabs first)");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  if (cpu_tsdf::volfmt::log2_exact(p.res[0]) < 0 || p.res[1] != p.res[0] || p.res[2] != p.res[0]) {
    tsdf_set_error("the .vol octree format needs a cubic power-of-two resolution");
    return TSDF_HIP_E_UNSUPPORTED;
  }
  tsdf_vol_meta m;
  if (meta)
    m = *meta;
  else
    default_meta(p, &m);
  VolHeader hd;
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
  int rc = TSDF_HIP_OK;
  std::string err;
  const bool ok = cpu_tsdf::vol_write_stream(
      filename, hd, tsdf_tuning().vol_chunk,
      [&](int x0, int y0, int z0, int c, float *d, float *w, unsigned char *rgb) {
        rc = tsdf_hip_download(h, x0, y0, z0, c, c, c, d, w, rgb);
        return rc == TSDF_HIP_OK;
      },
      &err);
  if (ok) return TSDF_HIP_OK;
  if (rc) return rc;  // (tsdf_hip_download has set the message)
  tsdf_set_error(err);
  return TSDF_HIP_E_IO;
}

extern "C" int tsdf_hip_load(const char *filename, const tsdf_params *defaults, tsdf_handle *out,
                             tsdf_params *params_out, tsdf_vol_meta *meta_out) {
  if (!filename || !out) return TSDF_HIP_E_INVALID;
  *out = nullptr;
  tsdf_params base;
  if (defaults)
    base = *defaults;
  else
    tsdf_hip_default_params(&base);
  for (int attempt = 0; attempt < 2; ++attempt) {
    tsdf_handle h = nullptr;
    tsdf_params p = base;
    VolHeader hd;
    int rc = TSDF_HIP_OK;
    std::string err;
    const bool ok = cpu_tsdf::vol_read_stream(
        filename, hd, tsdf_tuning().vol_chunk,
        [&](const VolHeader &f) {
          for (int k = 0; k < 3; ++k) {
            p.res[k] = f.res[k];
            p.size[k] = f.size[k];
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
          if (attempt) p.layout = TSDF_LAYOUT_F32W;
          rc = tsdf_hip_create(&p, &h);
          return rc == TSDF_HIP_OK;
        },
        [&](int x0, int y0, int z0, int c, float *d, float *w, unsigned char *rgb) {
          rc = tsdf_hip_upload(h, x0, y0, z0, c, c, c, d, w, rgb);
          return rc == TSDF_HIP_OK;
        },
        &err);
    if (ok) {
      *out = h;
      if (params_out) {
        *params_out = p;
        params_out->layout = tsdf_hip_layout(h);
      }
      if (meta_out) {
        for (int k = 0; k < 3; ++k) meta_out->max_cell_size[k] = hd.max_cell[k];
        meta_out->is_empty = hd.is_empty;
        meta_out->weight_by_depth = hd.weight_by_depth;
        meta_out->weight_by_variance = hd.weight_by_variance;
        for (int i = 0; i < 16; ++i) meta_out->global_transform[i] = hd.global_transform[i];
      }
      return TSDF_HIP_OK;
    }
    const bool packed_misfit = rc == TSDF_HIP_E_UNSUPPORTED && h && tsdf_hip_layout(h) == TSDF_LAYOUT_PACKED;
    if (h) tsdf_hip_destroy(h);
    if (packed_misfit && attempt == 0 && base.layout == TSDF_LAYOUT_AUTO) continue;
    if (rc) return rc;
    tsdf_set_error(err);
    return TSDF_HIP_E_IO;
  }
  return TSDF_HIP_E_UNSUPPORTED;
}
