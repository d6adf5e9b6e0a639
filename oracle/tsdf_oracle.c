/*
 * tsdf_oracle.c -- CPU restatement of the cpu_tsdf hot path on a DENSE grid.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (cpu_tsdf_amd/, include/) may call, link or
 * import this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
 * only as the checker.
 *
 * Each function follows the reference (sdmiller/cpu_tsdf) line by line; citations are relative to
 * the reference tree.  The restatement is validated against the reference's own, unmodified
 * sources compiled with the in-repo PCL/Eigen stand-ins (oracle/_ref, see oracle/Makefile and
 * tests/test_oracle_golden.py).  The reference ships no tests or golden vectors of its own
 * (SURVEY.md section 4), and the PCL/Eigen arithmetic it calls is not vendored: the pieces marked
 * [PCL-recall]/[Eigen-recall] restate upstream behaviour from memory and are "parity unpinned"
 * (see DESIGN.md).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (no -march: like the reference's CMake
 * build, so no FMA contraction; fp32 arithmetic is plain IEEE single).
 *
 * Dense-grid equivalence: with every octree leaf at the finest level, getContainingVoxel
 * (src/lib/octree.cpp:112-133,628-643) is an index computation; octree-only behaviour (split,
 * prune, coarse leaves) has no counterpart here.  Arrays are [z][y][x], x fastest.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct oracle_params {
  int32_t res[3];
  float size[3];
  float max_dist_pos, max_dist_neg, max_weight;
  float min_sensor_dist, max_sensor_dist;
  double fx, fy, cx, cy;
  int32_t image_width, image_height;
  int32_t integrate_color;
  int32_t xform_order; /* 0: x*c0 + (y*c1 + (z*c2 + c3)), 1: ((m0*x + m1*y) + m2*z) + m3 */
} oracle_params;

/* x86 cvttsd2si: NaN / out-of-range -> INT_MIN.  A plain C cast would be UB there; the reference's
 * `int u = <double>` (tsdf_volume_octree.cpp:614-615, :568-570) compiles to cvttsd2si. */
static int cvtt(double v) { return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : INT_MIN; }

static int ilog2_exact(int v) {
  int l = 0;
  if (v <= 0 || (v & (v - 1))) return -1;
  while ((1 << l) < v) ++l;
  return l;
}

/* Octree node centres per axis: OctreeNode::split, src/lib/octree.cpp:244-266 (child centre =
 * ctr -/+ size_/4, child size = size_/2, float), root at 0 (octree.cpp:589-590).  Non power-of-two
 * resolutions have no octree; fall back to getVoxelCenter (tsdf_volume_octree.cpp:553-560). */
void oracle_centers(int res, float size, float *out) {
  const int L = ilog2_exact(res);
  if (L >= 0) {
    for (int i = 0; i < res; ++i) {
      float c = 0.f, s = size;
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

/* getVoxelCenter, tsdf_volume_octree.cpp:553-560 (one axis). */
static float voxel_center(const oracle_params *p, int axis, int i) {
  const float off = p->size[axis] / 2.0;
  return (float)(((size_t)i + 0.5) * p->size[axis] / (double)p->res[axis] - off);
}

void oracle_voxel_center(const oracle_params *p, int i, int j, int k, float out[3]) {
  out[0] = voxel_center(p, 0, i);
  out[1] = voxel_center(p, 1, j);
  out[2] = voxel_center(p, 2, k);
}

/* getVoxelIndex, tsdf_volume_octree.cpp:562-574. */
int oracle_voxel_index(const oracle_params *p, float x, float y, float z, int idx[3]) {
  const float v[3] = {x, y, z};
  int ok = 1;
  for (int a = 0; a < 3; ++a) {
    const double off = (double)p->size[a] / 2.0;
    idx[a] = cvtt(floor(((double)v[a] + off) / (double)p->size[a] * (double)p->res[a]));
    ok &= idx[a] >= 0 && idx[a] < p->res[a];
  }
  return ok;
}

/* One axis of OctreeNode::getContainingVoxel's descent, octree.cpp:112-121:
 * child bit = (x - ctr) > 0 at each level. */
static int descend_axis(float x, float size, int L) {
  float c = 0.f, s = size;
  int i = 0;
  for (int l = 0; l < L; ++l) {
    const int b = (x - c) > 0;
    const float off = s / 4;
    c = b ? c + off : c - off;
    s = s / 2;
    i = i * 2 + b;
  }
  return i;
}

/* The size an axis' node centres are built from.  OctreeNode keeps ONE size_, initialised from size_x
 * (octree.h:63-66), and split() offsets all three centre coordinates by size_/4 (octree.cpp:244-266): on a grid whose
 * setGridSize is not cubic the octree is still a cube of edge size_x.  (The bounds tests and the closed-form
 * getVoxelCenter / getVoxelIndex keep using the per-axis sizes, as in the reference.)  Without an octree equivalent
 * (a resolution that is not a power of two on every axis, or not cubic) the axis' own size is used. */
static float node_size(const oracle_params *p, int axis) {
  const int cubic = p->res[0] == p->res[1] && p->res[1] == p->res[2] && ilog2_exact(p->res[0]) >= 0;
  return cubic ? p->size[0] : p->size[axis];
}

/* Octree::getContainingVoxel, octree.cpp:628-643, on a fully split tree.  Returns 0 for NULL. */
int oracle_containing(const oracle_params *p, float x, float y, float z, int idx[3]) {
  if (isnan(z) || fabsf(x) > p->size[0] / 2 || fabsf(y) > p->size[1] / 2 || fabsf(z) > p->size[2] / 2)
    return 0;
  const float v[3] = {x, y, z};
  for (int a = 0; a < 3; ++a) {
    const int L = ilog2_exact(p->res[a]);
    if (L >= 0) {
      idx[a] = descend_axis(v[a], node_size(p, a), L);
    } else { /* no octree for this resolution: nearest cell by the closed form */
      int i = cvtt(floor(((double)v[a] + (double)p->size[a] / 2.0) / (double)p->size[a] * (double)p->res[a]));
      if (i < 0) i = 0;
      if (i >= p->res[a]) i = p->res[a] - 1;
      idx[a] = i;
    }
  }
  return 1;
}

/* ------------------------------------------------------------------------------------------
 * integrateCloud on the dense grid: updateVoxel leaf branch, include/cpu_tsdf/impl/
 * tsdf_volume_octree.hpp:143-208, reprojectPoint tsdf_volume_octree.cpp:611-617,
 * OctreeNode::addObservation octree.cpp:152-163, RGBNode::addObservation octree.cpp:328-337.
 * T = trans.inverse().cast<float>() (hpp:54), row-major 3x4.  depth = pt.z of cloud(u,v),
 * bgra = PCL PointXYZRGBA byte order.  rgb is 3 bytes per voxel.  Returns the number of voxels
 * that reached addObservation. */
/* The leaf branch of updateVoxel up to the call of addObservation (hpp:143-198): 1 and (*dn, *pixel) if the
 * voxel centre (x, y, z) is observed by this frame. */
static int observe(const oracle_params *p, const float T[12], float x, float y, float z, const float *depth,
                   float *dn_out, size_t *pixel) {
  const int W = p->image_width, H = p->image_height;
  float g[3];
  for (int r = 0; r < 3; ++r) { /* pcl::transformPoint, hpp:145 [PCL-recall] */
    const float *m = T + 4 * r;
    if (p->xform_order == 0)
      g[r] = x * m[0] + (y * m[1] + (z * m[2] + m[3]));
    else
      g[r] = ((m[0] * x + m[1] * y) + m[2] * z) + m[3];
  }
  if (g[2] < p->min_sensor_dist || g[2] > p->max_sensor_dist) return 0; /* hpp:146 */
  const int u = cvtt((double)g[0] * p->fx / (double)g[2] + p->cx);      /* .cpp:614 */
  const int v = cvtt((double)g[1] * p->fy / (double)g[2] + p->cy);      /* .cpp:615 */
  if (!(g[2] > 0 && u >= 0 && u < W && v >= 0 && v < H)) return 0;       /* .cpp:616 */
  const float zs = depth[(size_t)v * W + u];
  if (isnan(zs)) return 0; /* hpp:152 */
  float dn = zs - g[2];    /* hpp:159 */
  if (dn > p->max_dist_pos)
    dn = p->max_dist_pos; /* hpp:189-192 */
  else if (dn < -p->max_dist_neg)
    return 0;             /* hpp:193-196 */
  dn /= p->max_dist_neg;  /* hpp:198 */
  *dn_out = dn;
  *pixel = (size_t)v * W + u;
  return 1;
}

/* getFrustumCulledVoxels, tsdf_volume_octree.cpp:619-652: integrateCloud only visits the leaves pcl::FrustumCulling keeps
 * -- a pyramid of 1.1 x the field of view around the optical axis between the sensor-range planes, six plane tests
 * `pt.dot(plane) <= 0` on the leaf centre [PCL-recall: filters/impl/frustum_culling.hpp, as restated in
 * compat/mini_pcl.h].  For ordinary cameras it keeps everything updateVoxel would accept; with a principal point far off
 * centre (or a non-finite range) it drops voxels that project into the image.  The planes (l, r, t, b, far, near; 4
 * floats each) from the forward pose `trans` (row-major 4x4 doubles); every operation in float where PCL's is. */
static void v3_cross(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static void cull_plane(const float n[3], const float through[3], float out[4]) { /* (n, -through.dot(n)), 3-term tree sum */
  out[0] = n[0], out[1] = n[1], out[2] = n[2];
  out[3] = -(through[0] * n[0] + (through[1] * n[1] + through[2] * n[2]));
}
void oracle_reference_cull_planes(const oracle_params *p, const double trans[16], float planes[24]) {
  /* trans_robot = trans.matrix().cast<float>() * cam2robot (:633-638): columns (view, up, right, T) = (z, -y, x, t) of
   * the pose; the 4-term sums only ever add exact zeros */
  float view[3], up[3], right[3], T[3];
  for (int r = 0; r < 3; ++r) {
    view[r] = (float)trans[4 * r + 2];
    up[r] = -(float)trans[4 * r + 1];
    right[r] = (float)trans[4 * r + 0];
    T[r] = (float)trans[4 * r + 3];
  }
  /* :641-644; setHorizontalFOV / setVerticalFOV / set*PlaneDistance take floats */
  const float hfov = (float)(1.1 * 2 * fabs(atan(0.5 * p->image_width / p->fx) * 180 / M_PI));
  const float vfov = (float)(1.1 * 2 * fabs(atan(0.5 * p->image_height / p->fy) * 180 / M_PI));
  const float np_dist = p->min_sensor_dist, fp_dist = p->max_sensor_dist;
  const float vfov_rad = (float)(vfov * M_PI / 180), hfov_rad = (float)(hfov * M_PI / 180);
  const float np_h = (float)(2 * tan(vfov_rad / 2) * np_dist), np_w = (float)(2 * tan(hfov_rad / 2) * np_dist);
  const float fp_h = (float)(2 * tan(vfov_rad / 2) * fp_dist), fp_w = (float)(2 * tan(hfov_rad / 2) * fp_dist);
  float fp_c[3], fp_tl[3], fp_tr[3], fp_bl[3], fp_br[3], np_c[3], np_tr[3], np_bl[3], np_br[3];
  for (int i = 0; i < 3; ++i) {
    fp_c[i] = T[i] + view[i] * fp_dist;
    fp_tl[i] = fp_c[i] + (up[i] * fp_h / 2) - (right[i] * fp_w / 2);
    fp_tr[i] = fp_c[i] + (up[i] * fp_h / 2) + (right[i] * fp_w / 2);
    fp_bl[i] = fp_c[i] - (up[i] * fp_h / 2) - (right[i] * fp_w / 2);
    fp_br[i] = fp_c[i] - (up[i] * fp_h / 2) + (right[i] * fp_w / 2);
    np_c[i] = T[i] + view[i] * np_dist;
    np_tr[i] = np_c[i] + (up[i] * np_h / 2) + (right[i] * np_w / 2);
    np_bl[i] = np_c[i] - (up[i] * np_h / 2) - (right[i] * np_w / 2);
    np_br[i] = np_c[i] - (up[i] * np_h / 2) + (right[i] * np_w / 2);
  }
  float e1[3], e2[3], n[3], a[3], b[3], c[3], d[3];
  for (int i = 0; i < 3; ++i) e1[i] = fp_bl[i] - fp_br[i], e2[i] = fp_tr[i] - fp_br[i];
  v3_cross(e1, e2, n);
  cull_plane(n, fp_c, planes + 16); /* far */
  for (int i = 0; i < 3; ++i) e1[i] = np_tr[i] - np_br[i], e2[i] = np_bl[i] - np_br[i];
  v3_cross(e1, e2, n);
  cull_plane(n, np_c, planes + 20); /* near */
  for (int i = 0; i < 3; ++i) a[i] = fp_bl[i] - T[i], b[i] = fp_br[i] - T[i], c[i] = fp_tr[i] - T[i], d[i] = fp_tl[i] - T[i];
  v3_cross(b, c, n);
  cull_plane(n, T, planes + 4); /* right */
  v3_cross(d, a, n);
  cull_plane(n, T, planes + 0); /* left */
  v3_cross(c, d, n);
  cull_plane(n, T, planes + 8); /* top */
  v3_cross(a, b, n);
  cull_plane(n, T, planes + 12); /* bottom */
}

static int cull_keeps(const float planes[24], float x, float y, float z) { /* is_in_fov: Vector4f dot = (p0 + p1) + (p2 + p3) */
  for (int k = 0; k < 6; ++k) {
    const float *pl = planes + 4 * k;
    if (!((x * pl[0] + y * pl[1]) + (z * pl[2] + 1.0f * pl[3]) <= 0)) return 0;
  }
  return 1;
}

/* weight_by_depth: hpp:200-202, `w_new *= (1 - std::min(pt.z / 10., 1.))` -- a float times a double, stored
 * back into the float; the flag has no setter and only arrives through load() (tsdf_volume_octree.cpp:265). */
/* weight_by_variance: hpp:203-204, `w_new *= std::exp(logNormal(d_new, voxel->d_, voxel->getVariance()))` once the
 * voxel has more than 5 samples.  logNormal (hpp:106-110) = -std::pow(x - mean, 2) / (2 * var): pow(float, int) is
 * the double pow, the quotient a double, the result stored in a float; getVariance (octree.cpp:281-287) =
 * (M_ / w_) * (nsample_ / (nsample_ - 1)) with an INTEGER quotient (1 for every count it is reached with); std::exp of
 * a float is expf.  M_ and nsample_ (Mv, nv: per voxel, like d) are updated by every addObservation (octree.cpp:160-161). */
static uint64_t integrate_impl(const oracle_params *p, float *d, float *w, uint8_t *rgb, const float *depth,
                               const uint8_t *bgra, const float T[12], int z_begin, int z_end, int weight_by_depth,
                               const float *cull_planes, float *Mv, int32_t *nv) {
  const int nx = p->res[0], ny = p->res[1], nz = p->res[2];
  float *cx = (float *)malloc(sizeof(float) * nx), *cy = (float *)malloc(sizeof(float) * ny),
        *cz = (float *)malloc(sizeof(float) * nz);
  oracle_centers(nx, node_size(p, 0), cx);
  oracle_centers(ny, node_size(p, 1), cy);
  oracle_centers(nz, node_size(p, 2), cz);
  if (z_begin == 0 && z_end == 0) z_end = nz;
  uint64_t n_obs = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_obs)
  for (int k = z_begin; k < z_end; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        float dn;
        size_t pixel;
        if (cull_planes && !cull_keeps(cull_planes, cx[i], cy[j], cz[k])) continue; /* hpp:93-94 */
        if (!observe(p, T, cx[i], cy[j], cz[k], depth, &dn, &pixel)) continue;
        float wn = 1;           /* hpp:200 (the variance weighting of :203-204 needs M_ / nsample_: not restated) */
        if (weight_by_depth) {  /* hpp:201-202; std::min(a, b) = (b < a) ? b : a */
          const double a = depth[pixel] / 10.;
          wn = (float)((double)wn * (1 - ((1. < a) ? 1. : a)));
        }
        const size_t vi = ((size_t)k * ny + j) * nx + i;
        if (Mv && nv[vi] > 5) { /* hpp:203-204 */
          const float var = (Mv[vi] / w[vi]) * (float)(nv[vi] / (nv[vi] - 1));
          const float ln = (float)(-pow((double)(dn - d[vi]), 2.0) / (double)(2 * var));
          wn *= expf(ln);
        }
        if (p->integrate_color && rgb) { /* octree.cpp:331-335 (old w, truncation) */
          const uint8_t *px = bgra + 4 * pixel;
          const float wsum = w[vi] + wn;
          /* static_cast<uint8_t>(float): cvttss2si, then the low byte (NaN -> 0) */
          rgb[3 * vi + 0] = (uint8_t)(cvtt((double)((w[vi] * rgb[3 * vi + 0] + wn * px[2]) / wsum)) & 255);
          rgb[3 * vi + 1] = (uint8_t)(cvtt((double)((w[vi] * rgb[3 * vi + 1] + wn * px[1]) / wsum)) & 255);
          rgb[3 * vi + 2] = (uint8_t)(cvtt((double)((w[vi] * rgb[3 * vi + 2] + wn * px[0]) / wsum)) & 255);
        }
        const float d_old = d[vi];
        d[vi] = (d[vi] * w[vi] + dn * wn) / (w[vi] + wn); /* octree.cpp:156 */
        w[vi] += wn;                                       /* octree.cpp:157 */
        if (w[vi] > p->max_weight) w[vi] = p->max_weight;  /* octree.cpp:158-159 */
        if (Mv) {
          Mv[vi] += wn * (dn - d[vi]) * (dn - d_old); /* octree.cpp:160 */
          ++nv[vi];                                   /* octree.cpp:161 */
        }
        ++n_obs;
      }
  free(cx);
  free(cy);
  free(cz);
  return n_obs;
}

uint64_t oracle_integrate(const oracle_params *p, float *d, float *w, uint8_t *rgb, const float *depth,
                          const uint8_t *bgra, const float T[12], int z_begin, int z_end) {
  return integrate_impl(p, d, w, rgb, depth, bgra, T, z_begin, z_end, 0, NULL, NULL, NULL);
}

/* integrateCloud INCLUDING the reference's frustum cull (planes from oracle_reference_cull_planes). */
uint64_t oracle_integrate_culled(const oracle_params *p, float *d, float *w, uint8_t *rgb, const float *depth,
                                 const uint8_t *bgra, const float T[12], int z_begin, int z_end, const float planes[24]) {
  return integrate_impl(p, d, w, rgb, depth, bgra, T, z_begin, z_end, 0, planes, NULL, NULL);
}

uint64_t oracle_integrate_weighted(const oracle_params *p, float *d, float *w, uint8_t *rgb, const float *depth,
                                   const uint8_t *bgra, const float T[12], int z_begin, int z_end, int weight_by_depth) {
  return integrate_impl(p, d, w, rgb, depth, bgra, T, z_begin, z_end, weight_by_depth, NULL, NULL, NULL);
}

/* std::exp(float) as the reference calls it (hpp:204): the host libm's expf, for the GPU tests' sweep of the device's form */
void oracle_expf_many(const float *in, size_t n, float *out) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; ++i) out[i] = expf(in[i]);
}

/* integrateCloud of a volume whose header carries weight_by_variance_ (and possibly weight_by_depth_): M / nsample are the
 * per-voxel OctreeNode::M_ / nsample_ planes ([z][y][x] like d). */
uint64_t oracle_integrate_variance(const oracle_params *p, float *d, float *w, uint8_t *rgb, float *M, int32_t *nsample,
                                   const float *depth, const uint8_t *bgra, const float T[12], int z_begin, int z_end,
                                   int weight_by_depth) {
  return integrate_impl(p, d, w, rgb, depth, bgra, T, z_begin, z_end, weight_by_depth, NULL, M, nsample);
}

/* The same with RGBNormalized voxels (setColorMode("RGBNormalized")): RGBNormalized::addObservation,
 * octree.cpp:380-393, and getRGB, octree.cpp:396-402.  cn = four planes (r_n, g_n, b_n, i), each nz*ny*nx
 * floats starting at 0 (octree.h:217-222); rgb receives what getRGB() returns for every voxel touched. */
uint64_t oracle_integrate_rgbn(const oracle_params *p, float *d, float *w, float *cn, uint8_t *rgb,
                               const float *depth, const uint8_t *bgra, const float T[12], int z_begin, int z_end) {
  const int nx = p->res[0], ny = p->res[1], nz = p->res[2];
  const size_t n = (size_t)nx * ny * nz;
  float *r_n = cn, *g_n = cn + n, *b_n = cn + 2 * n, *i_m = cn + 3 * n;
  float *cx = (float *)malloc(sizeof(float) * nx), *cy = (float *)malloc(sizeof(float) * ny),
        *cz = (float *)malloc(sizeof(float) * nz);
  oracle_centers(nx, node_size(p, 0), cx);
  oracle_centers(ny, node_size(p, 1), cy);
  oracle_centers(nz, node_size(p, 2), cz);
  if (z_begin == 0 && z_end == 0) z_end = nz;
  uint64_t n_obs = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_obs)
  for (int k = z_begin; k < z_end; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        float dn;
        size_t pixel;
        if (!observe(p, T, cx[i], cy[j], cz[k], depth, &dn, &pixel)) continue;
        const float w_new = 1;
        const size_t vi = ((size_t)k * ny + j) * nx + i;
        const uint8_t *px = bgra + 4 * pixel;
        const uint8_t r = px[2], g = px[1], b = px[0];
        const float wsum = w[vi] + w_new;                                                           /* :383 */
        const float in = sqrtf((float)r * (float)r + (float)g * (float)g + (float)b * (float)b);    /* :384 */
        const float r_f = r / in, g_f = g / in, b_f = b / in;                                       /* :385-387 */
        r_n[vi] = (w[vi] * r_n[vi] + w_new * r_f) / wsum;                                           /* :388 */
        g_n[vi] = (w[vi] * g_n[vi] + w_new * g_f) / wsum;
        b_n[vi] = (w[vi] * b_n[vi] + w_new * b_f) / wsum;
        i_m[vi] = (w[vi] * i_m[vi] + w_new * in) / wsum;                                            /* :391 */
        /* getRGB: `uint8_t r = r_n_ * i_` is cvttss2si + low byte on x86-64 (NaN -> 0x80000000 -> 0) */
        rgb[3 * vi + 0] = (uint8_t)cvtt((double)(r_n[vi] * i_m[vi]));
        rgb[3 * vi + 1] = (uint8_t)cvtt((double)(g_n[vi] * i_m[vi]));
        rgb[3 * vi + 2] = (uint8_t)cvtt((double)(b_n[vi] * i_m[vi]));
        d[vi] = (d[vi] * w[vi] + dn * w_new) / (w[vi] + w_new); /* octree.cpp:156 */
        w[vi] += w_new;
        if (w[vi] > p->max_weight) w[vi] = p->max_weight;
        ++n_obs;
      }
  free(cx);
  free(cy);
  free(cz);
  return n_obs;
}

/* ------------------------------------------------------------------------------------------
 * setColorMode("LAB"): RGB2LAB octree.cpp:436-481, LAB2RGB octree.cpp:483-527, LABNode octree.cpp:531-551.
 * Every literal below is a double in the reference too, so the mixed float/double arithmetic is the same in C;
 * std::pow(float, int) is the C++11 <cmath> overload that promotes both to double (octree.cpp:491-502). */
void oracle_rgb2lab(uint8_t r, uint8_t g, uint8_t b, float *L, float *A, float *B) {
  float rf = ((float)r / 255.), gf = ((float)g / 255.), bf = ((float)b / 255.); /* :441-443 */
  if (rf > 0.0405) rf = pow(((rf + 0.055) / 1.055), 2.4); else rf /= 12.92;     /* :444-447 */
  if (gf > 0.0405) gf = pow(((gf + 0.055) / 1.055), 2.4); else gf /= 12.92;
  if (bf > 0.0405) bf = pow(((bf + 0.055) / 1.055), 2.4); else bf /= 12.92;
  rf *= 100;
  gf *= 100;
  bf *= 100;
  float X = rf * 0.4124 + gf * 0.3576 + bf * 0.1805; /* :459-461 */
  float Y = rf * 0.2126 + gf * 0.7152 + bf * 0.0722;
  float Z = rf * 0.0193 + gf * 0.1192 + bf * 0.9505;
  X /= 95.047;
  Y /= 100.;
  Z /= 108.883;
  if (X > 0.008856) X = pow((double)X, 1 / 3.); else X = 7.787 * X + (16 / 116.); /* :466-477 */
  if (Y > 0.008856) Y = pow((double)Y, 1 / 3.); else Y = 7.787 * Y + (16 / 116.);
  if (Z > 0.008856) Z = pow((double)Z, 1 / 3.); else Z = 7.787 * Z + (16 / 116.);
  *L = (116 * Y) - 16; /* :478-480: float arithmetic */
  *A = 500 * (X - Y);
  *B = 200 * (Y - Z);
}

void oracle_lab2rgb(float L, float A, float B, uint8_t *r, uint8_t *g, uint8_t *b) {
  float Y = (L + 16) / 116.; /* :488-490 */
  float X = A / 500. + Y;
  float Z = Y - (B / 200.);
  if (pow((double)X, 3.0) > 0.008856) X = pow((double)X, 3.0); else X = (X - 16 / 116.) / 7.787; /* :491-502 */
  if (pow((double)Y, 3.0) > 0.008856) Y = pow((double)Y, 3.0); else Y = (Y - 16 / 116.) / 7.787;
  if (pow((double)Z, 3.0) > 0.008856) Z = pow((double)Z, 3.0); else Z = (Z - 16 / 116.) / 7.787;
  X *= 95.047;
  Y *= 100.;
  Z *= 108.883;
  X /= 100;
  Y /= 100;
  Z /= 100;
  float rf = X * +3.2406 + Y * -1.5372 + Z * -0.4986; /* :511-513 */
  float gf = X * -0.9689 + Y * +1.8758 + Z * +0.0415;
  float bf = X * +0.0557 + Y * -0.2040 + Z * +1.0570;
  if (rf > 0.0031308) rf = 1.055 * pow((double)rf, 1. / 2.4) - 0.055; else rf *= 12.92; /* :514-525 */
  if (gf > 0.0031308) gf = 1.055 * pow((double)gf, 1. / 2.4) - 0.055; else gf *= 12.92;
  if (bf > 0.0031308) bf = 1.055 * pow((double)bf, 1. / 2.4) - 0.055; else bf *= 12.92;
  /* static_cast<uint8_t>(float) is cvttss2si + low byte on x86-64, as in getRGB above */
  *r = (uint8_t)cvtt((double)(rf * 255));
  *g = (uint8_t)cvtt((double)(gf * 255));
  *b = (uint8_t)cvtt((double)(bf * 255));
}

void oracle_rgb2lab_many(const uint8_t *rgb, size_t n, float *lab) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; ++i) oracle_rgb2lab(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], lab + 3 * i, lab + 3 * i + 1, lab + 3 * i + 2);
}
void oracle_lab2rgb_many(const float *lab, size_t n, uint8_t *rgb) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; ++i) oracle_lab2rgb(lab[3 * i], lab[3 * i + 1], lab[3 * i + 2], rgb + 3 * i, rgb + 3 * i + 1, rgb + 3 * i + 2);
}

/* integrate with LABNode voxels.  cn = three planes (L, A, B) starting at 0 (octree.h:267-270); rgb receives what
 * getRGB() returns (LAB2RGB of the means, octree.cpp:547-551) for every voxel touched. */
uint64_t oracle_integrate_lab(const oracle_params *p, float *d, float *w, float *cn, uint8_t *rgb,
                              const float *depth, const uint8_t *bgra, const float T[12], int z_begin, int z_end) {
  const int nx = p->res[0], ny = p->res[1], nz = p->res[2];
  const size_t n = (size_t)nx * ny * nz;
  float *Lm = cn, *Am = cn + n, *Bm = cn + 2 * n;
  float *cx = (float *)malloc(sizeof(float) * nx), *cy = (float *)malloc(sizeof(float) * ny),
        *cz = (float *)malloc(sizeof(float) * nz);
  oracle_centers(nx, node_size(p, 0), cx);
  oracle_centers(ny, node_size(p, 1), cy);
  oracle_centers(nz, node_size(p, 2), cz);
  if (z_begin == 0 && z_end == 0) z_end = nz;
  uint64_t n_obs = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_obs)
  for (int k = z_begin; k < z_end; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        float dn;
        size_t pixel;
        if (!observe(p, T, cx[i], cy[j], cz[k], depth, &dn, &pixel)) continue;
        const float w_new = 1;
        const size_t vi = ((size_t)k * ny + j) * nx + i;
        const uint8_t *px = bgra + 4 * pixel;
        const float wsum = w[vi] + w_new; /* octree.cpp:535 */
        float Ln, An, Bn;
        oracle_rgb2lab(px[2], px[1], px[0], &Ln, &An, &Bn);
        Lm[vi] = (w[vi] * Lm[vi] + w_new * Ln) / wsum; /* :540-542 */
        Am[vi] = (w[vi] * Am[vi] + w_new * An) / wsum;
        Bm[vi] = (w[vi] * Bm[vi] + w_new * Bn) / wsum;
        oracle_lab2rgb(Lm[vi], Am[vi], Bm[vi], rgb + 3 * vi, rgb + 3 * vi + 1, rgb + 3 * vi + 2);
        d[vi] = (d[vi] * w[vi] + dn * w_new) / (w[vi] + w_new); /* octree.cpp:156 */
        w[vi] += w_new;
        if (w[vi] > p->max_weight) w[vi] = p->max_weight;
        ++n_obs;
      }
  free(cx);
  free(cy);
  free(cz);
  return n_obs;
}

/* ------------------------------------------------------------------------------------------
 * interpolateTrilinearly, tsdf_volume_octree.cpp:486-541.  *valid is AND-ed (never set to 1). */
/* Slab bookkeeping for the ray hand-off restatement (oracle_raycast_advance): planes a slab may read. */
static _Thread_local int tl_zlo = INT_MIN, tl_zhi = INT_MAX, tl_bad = 0;
static void touch_plane(int z) {
  if (z < tl_zlo || z >= tl_zhi) tl_bad = 1;
}

static float trilinear(const oracle_params *p, const float *d, const float *w, float x, float y, float z,
                       int *valid) {
  int id[3];
  const int nx = p->res[0], ny = p->res[1], nz = p->res[2];
  const int exists = oracle_voxel_index(p, x, y, z, id);
  if (!exists || id[0] <= 0 || id[0] >= nx - 1 || id[1] <= 0 || id[1] >= ny - 1 || id[2] <= 0 ||
      id[2] >= nz - 1) {
    if (valid) *valid = 0;
    return NAN;
  }
  int xi = id[0], yi = id[1], zi = id[2];
  float vx = voxel_center(p, 0, xi), vy = voxel_center(p, 1, yi), vz = voxel_center(p, 2, zi);
  if (x < vx) xi -= 1;
  if (y < vy) yi -= 1;
  if (z < vz) zi -= 1;
  vx = voxel_center(p, 0, xi);
  vy = voxel_center(p, 1, yi);
  vz = voxel_center(p, 2, zi);
  const float a = (x - vx) * nx / p->size[0];
  const float b = (y - vy) * ny / p->size[1];
  const float c = (z - vz) * nz / p->size[2];
  touch_plane(zi);
  touch_plane(zi + 1);
#define VI(i, j, k) (((size_t)(k) * ny + (j)) * nx + (i))
  const size_t o = VI(xi, yi, zi), ox = VI(xi + 1, yi, zi), oy = VI(xi, yi + 1, zi), oz = VI(xi, yi, zi + 1),
               oxy = VI(xi + 1, yi + 1, zi), oxz = VI(xi + 1, yi, zi + 1), oyz = VI(xi, yi + 1, zi + 1),
               oxyz = VI(xi + 1, yi + 1, zi + 1);
  if (valid) {
    *valid &= (w[o] > 0);
    *valid &= (w[ox] > 0);
    *valid &= (w[oy] > 0);
    *valid &= (w[oz] > 0);
    *valid &= (w[oxy] > 0);
    *valid &= (w[oxz] > 0);
    *valid &= (w[oyz] > 0);
    *valid &= (w[oxyz] > 0);
  }
  return (d[o] * (1 - a) * (1 - b) * (1 - c) + d[oz] * (1 - a) * (1 - b) * (c) +
          d[oy] * (1 - a) * (b) * (1 - c) + d[oyz] * (1 - a) * (b) * (c) + d[ox] * (a) * (1 - b) * (1 - c) +
          d[oxz] * (a) * (1 - b) * (c) + d[oxy] * (a) * (b) * (1 - c) + d[oxyz] * (a) * (b) * (c));
}

float oracle_trilinear(const oracle_params *p, const float *d, const float *w, float x, float y, float z,
                       int *valid) {
  return trilinear(p, d, w, x, y, z, valid);
}

/* Eigen::Vector3f::normalize() [Eigen-recall, 3.3+]: z = squaredNorm(); if (z > 0) v /= sqrt(z);
 * squaredNorm of a 3-vector unrolls as x*x + (y*y + z*z); operator/= is a true division. */
static void normalize3(float v[3]) {
  const float n2 = v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]);
  if (n2 > 0.f) {
    const float n = sqrtf(n2);
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
  }
}

static float leaf_size(const oracle_params *p, int axis) { /* size_ after L halvings */
  float s = p->size[axis];
  const int L = ilog2_exact(p->res[axis]);
  if (L < 0) return p->size[axis] / p->res[axis];
  for (int l = 0; l < L; ++l) s = s / 2;
  return s;
}

/* renderView, tsdf_volume_octree.cpp:278-421, WITHOUT the final transformPointCloudWithNormals
 * (:422): output stays in the volume frame.  rot = trans.rotation().cast<float>() (row-major 3x3),
 * org = trans.translation().cast<float>().  out: 8 floats per pixel x,y,z,nx,ny,nz,t,niter; a miss
 * has NaN xyz and zero normal (PointNormal's default constructor).
 *
 * One ray.  With st == NULL this is the reference's loop start to finish.  With st != NULL it is the
 * multi-slab restatement used by the gloo tests (same protocol as tsdf_hip_raycast_advance,
 * include/tsdf_hip.h): the loop state is loaded from the ray's record, the main loop stops BEFORE it would
 * read a voxel whose plane is outside [slab[2], slab[3]), and the new record goes to dl. */
#define RAY_REC 24
static float i2f(int32_t v) { float f; memcpy(&f, &v, 4); return f; }
static int32_t f2i(float f) { int32_t v; memcpy(&v, &f, 4); return v; }

static void ray_direction(const oracle_params *p, const float rot[9], int ds, int64_t i, float du[3]) {
  const int nw = p->image_width / ds;
  const double nfx = p->fx / ds, nfy = p->fy / ds, ncx = p->cx / ds, ncy = p->cy / ds;
  const size_t x = i % nw, y = i / nw;
  du[0] = (float)((x - ncx) / nfx);
  du[1] = (float)((y - ncy) / nfy);
  du[2] = 1.f;
  normalize3(du);
  /* du = R * du  [Eigen-recall 3.3: each coefficient is a 3-term reduction p0 + (p1 + p2)] */
  const float a = du[0], b = du[1], c = du[2];
  for (int r = 0; r < 3; ++r) du[r] = rot[3 * r] * a + (rot[3 * r + 1] * b + rot[3 * r + 2] * c);
}

static void ray_one(const oracle_params *p, const float *d, const float *w, const float rot[9],
                    const float org[3], int ds, int64_t i, float *o, const int32_t *st, int32_t *dl,
                    const int *slab /* rank, world, z_begin, z_end */) {
  /* list mode (slab[6] > 0): i is the position in a compact list, the ray's pixel index is word 11 */
  const int64_t slot = i;
  if (st && slab[6] > 0) i = st[RAY_REC * slot + 11];
  const int nx = p->res[0], ny = p->res[1];
  const float min_step = p->max_dist_neg * 3 / 4.; /* :289 */
  /* leaf->getMinSize() is size_ = size_x for every axis (octree.h:63-66) */
  const float lsz = leaf_size(p, 0);
  int found_crossing = 0, finish_only = 0;
  float du[3];
  ray_direction(p, rot, ds, i, du);
  float pt[3] = {org[0], org[1], org[2]};
  float dd = 0, ww = 0, last_w = 0, last_d = 0;
  float t = p->min_sensor_dist;
  float step = min_step;
  int hit_voxel = 0, niter = 0;
  if (st) {
    const int32_t *r = st + RAY_REC * slot;
    if (r[0] != 1) return;
    const int mine = r[1] < 0 ? (int)(i % slab[1]) == slab[0] : (r[1] >= slab[2] && r[1] < slab[3]);
    if (!mine) return;
    niter = r[2];
    hit_voxel = r[3];
    t = i2f(r[4]);
    for (int k = 0; k < 3; ++k) pt[k] = i2f(r[5 + k]);
    last_d = i2f(r[8]);
    last_w = i2f(r[9]);
    step = i2f(r[10]);
    finish_only = r[12] != 0; /* the main loop is done; only the normal at t_star = t is left (see below) */
  } else {
    for (int k = 0; k < 3; ++k) pt[k] += t * du[k];
  }
  int suspend_z = -1;
  if (finish_only) found_crossing = 1;
  while (!finish_only && t < p->max_sensor_dist) {
    int id[3];
    if (oracle_containing(p, pt[0], pt[1], pt[2], id)) {
      if (st && (id[2] < slab[2] || id[2] >= slab[3])) {
        suspend_z = id[2];
        break;
      }
      const size_t vi = ((size_t)id[2] * ny + id[1]) * nx + id[0];
      hit_voxel = 1;
      dd = d[vi];
      ww = w[vi];
      if (((dd < 0 && last_d > 0) || (dd > 0 && last_d < 0)) && last_w && ww) {
        found_crossing = 1;
        const float old_t = t - step;
        step = (p->size[2] / p->res[2]) / 2.; /* :329 */
        float new_d, new_w;
        float last_new_d = dd, last_new_w = ww;
        while (t >= old_t) {
          t -= step;
          for (int k = 0; k < 3; ++k) pt[k] -= step * du[k];
          if (!oracle_containing(p, pt[0], pt[1], pt[2], id)) break;
          touch_plane(id[2]);
          const size_t vj = ((size_t)id[2] * ny + id[1]) * nx + id[0];
          new_d = d[vj];
          new_w = w[vj];
          if ((last_d > 0 && new_d > 0) || (last_d < 0 && new_d < 0)) {
            last_d = new_d;
            last_w = new_w;
            dd = last_new_d;
            ww = last_new_w;
            t += step;
            for (int k = 0; k < 3; ++k) pt[k] += step * du[k];
            break;
          }
          last_new_d = dd; /* sic: the reference assigns d, not new_d (:352-353) */
          last_new_w = ww;
        }
        break;
      }
      last_d = dd;
      last_w = ww;
      { /* :360  step = max(size/4, (float)(fabs(d)*max_dist_neg_)) */
        const float s1 = lsz / 4.f, s2 = (float)(fabs(dd) * p->max_dist_neg);
        step = s1 < s2 ? s2 : s1; /* std::max(a,b) = (a<b)?b:a */
      }
    } else if (hit_voxel) {
      break;
    }
    t += step;
    for (int k = 0; k < 3; ++k) pt[k] += step * du[k];
    niter++;
  }
  if (st) {
    int32_t *r = dl + RAY_REC * slot;
    r[0] = suspend_z >= 0 ? 1 : 2;
    r[1] = suspend_z;
    r[2] = niter;
    r[3] = hit_voxel;
    r[4] = f2i(t);
    for (int k = 0; k < 3; ++k) r[5 + k] = f2i(pt[k]);
    r[8] = f2i(last_d);
    r[9] = f2i(last_w);
    r[10] = f2i(step);
    r[12] = 0;
    if (suspend_z >= 0) return;
    o = (float *)(r + 16);
  }
  o[3] = o[4] = o[5] = 0.f;
  o[6] = t;
  o[7] = (float)niter;
  if (!found_crossing) {
    o[0] = o[1] = o[2] = NAN;
    return;
  }
  float t_star = t; /* finish_only: the record carries t_star in the t word */
  if (!finish_only) {
    int has_data = 1;
    const float tcurr = t, tprev = t - step;
    last_d = trilinear(p, d, w, org[0] + tprev * du[0], org[1] + tprev * du[1], org[2] + tprev * du[2], &has_data);
    dd = trilinear(p, d, w, org[0] + tcurr * du[0], org[1] + tcurr * du[1], org[2] + tcurr * du[2], &has_data);
    /* :385-388 sets NaN but does not `continue`; :389-390 then overwrites xyz anyway */
    /* unqualified fabs(float) resolves to double fabs(double) with <cmath> only (g++), so the
     * update of t_star is evaluated in double */
    t_star = t + step * (-1 + fabs(last_d / (last_d - dd)));
  }
  for (int k = 0; k < 3; ++k) o[k] = org[k] + t_star * du[k];
  o[6] = t_star;
  int id[3];
  if (!oracle_containing(p, o[0], o[1], o[2], id)) {
    o[3] = o[4] = o[5] = NAN;
    return;
  }
  /* Slab hand-off only: t_star extrapolates from two trilinear samples and can land ANY distance ahead when they
   * are nearly equal, so the six samples of the normal may need planes of another slab.  The ray then travels once
   * more: suspended for the owner of the hit point's plane with the finish flag (word 12) and t_star in the t word;
   * that rank recomputes o = org + t_star * du (same arithmetic) and the normal.  (Only when the planes the normal
   * reads, id[2]-2 .. id[2]+2, are not all held here: slab[4..5] = the allocated range.) */
  if (st && !finish_only && ((id[2] - 2 < slab[4] && slab[4] > 0) || (id[2] + 2 >= slab[5] && slab[5] < p->res[2]))) {
    int32_t *r = dl + RAY_REC * slot;
    r[0] = 1;
    r[1] = id[2];
    r[4] = f2i(t_star);
    r[12] = 1;
    return;
  }
  const float xs = lsz, ys = lsz, zs = lsz; /* getSize returns size_ thrice (octree.cpp:58-64) */
  int valid = 1;
  const float d_xm = trilinear(p, d, w, o[0] - xs, o[1], o[2], &valid);
  const float d_xp = trilinear(p, d, w, o[0] + xs, o[1], o[2], &valid);
  const float d_ym = trilinear(p, d, w, o[0], o[1] - ys, o[2], &valid);
  const float d_yp = trilinear(p, d, w, o[0], o[1] + ys, o[2], &valid);
  const float d_zm = trilinear(p, d, w, o[0], o[1], o[2] - zs, &valid);
  const float d_zp = trilinear(p, d, w, o[0], o[1], o[2] + zs, &valid);
  if (!valid) {
    o[3] = o[4] = o[5] = NAN;
    return;
  }
  float dF[3];
  dF[0] = (d_xp - d_xm) * p->max_dist_neg / (2 * xs);
  dF[1] = (d_yp - d_ym) * p->max_dist_neg / (2 * ys);
  dF[2] = (d_zp - d_zm) * p->max_dist_neg / (2 * zs);
  normalize3(dF);
  o[3] = dF[0];
  o[4] = dF[1];
  o[5] = dF[2];
}

void oracle_raycast(const oracle_params *p, const float *d, const float *w, const float rot[9],
                    const float org[3], int ds, float *out) {
  const int nw = p->image_width / ds, nh = p->image_height / ds;
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < (int64_t)nw * nh; ++i) {
    tl_zlo = INT_MIN;
    tl_zhi = INT_MAX;
    ray_one(p, d, w, rot, org, ds, i, out + 8 * i, NULL, NULL, NULL);
  }
}

/* Start record of every ray (same layout as tsdf_hip_raycast_begin). */
void oracle_raycast_begin(const oracle_params *p, const float rot[9], const float org[3], int ds, int32_t *state) {
  const int nw = p->image_width / ds, nh = p->image_height / ds;
  for (int64_t i = 0; i < (int64_t)nw * nh; ++i) {
    float du[3];
    ray_direction(p, rot, ds, i, du);
    int32_t *r = state + RAY_REC * i;
    memset(r, 0, RAY_REC * sizeof(int32_t));
    const float t = p->min_sensor_dist;
    r[0] = 1;
    r[1] = -1;
    r[4] = f2i(t);
    for (int k = 0; k < 3; ++k) {
      float pt = org[k];
      pt += t * du[k];
      r[5 + k] = f2i(pt);
    }
    r[10] = f2i(p->max_dist_neg * 3 / 4.);
    r[11] = (int32_t)i;
  }
}

/* The compact-list form (tsdf_hip_raycast_advance_list): `count` records updated in place. */
int oracle_raycast_advance_list(const oracle_params *p, const float *d, const float *w, const float rot[9],
                                const float org[3], int ds, const int *slab6, int32_t *records, int count) {
  int bad = 0;
  int slab[7];
  memcpy(slab, slab6, 6 * sizeof(int));
  slab[6] = count;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : bad)
  for (int64_t i = 0; i < (int64_t)count; ++i) {
    tl_zlo = slab[4];
    tl_zhi = slab[5];
    tl_bad = 0;
    ray_one(p, d, w, rot, org, ds, i, NULL, records, records, slab);
    bad += tl_bad;
  }
  return bad;
}

/* One hand-off round of one slab.  slab = {rank, world, z_begin, z_end, alloc_lo, alloc_hi}; returns the
 * number of rays whose refinement walk / trilinear samples read a plane outside [alloc_lo, alloc_hi). */
int oracle_raycast_advance(const oracle_params *p, const float *d, const float *w, const float rot[9],
                           const float org[3], int ds, const int *slab, const int32_t *state, int32_t *delta) {
  const int nw = p->image_width / ds, nh = p->image_height / ds;
  int bad = 0;
  int slab7[7];
  memcpy(slab7, slab, 6 * sizeof(int));
  slab7[6] = 0;
  memset(delta, 0, (size_t)nw * nh * RAY_REC * sizeof(int32_t));
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : bad)
  for (int64_t i = 0; i < (int64_t)nw * nh; ++i) {
    tl_zlo = slab[4];
    tl_zhi = slab[5];
    tl_bad = 0;
    ray_one(p, d, w, rot, org, ds, i, NULL, state, delta, slab7);
    bad += tl_bad;
  }
  return bad;
}

/* ------------------------------------------------------------------------------------------
 * The `integrate` program's per-cloud preparation, src/prog/integrate.cpp:559-618, as the serial loop it
 * is there: units (:559-568), zero -> NaN (:570-578), world -> camera by pcl::transformPointCloud with
 * poses[i].inverse() (:580-581) [PCL-recall: Transformer<double>::se3 = x*c0 + (y*c1 + (z*c2 + c3)) in
 * double, cast to float; non-finite points of a non-dense cloud are skipped], then the z-buffer
 * reprojection into a width x height organised cloud (:583-617) with reprojectPoint (:201-207), whose
 * intrinsics are FLOAT globals.  tf = first three rows of poses[i].inverse().matrix() or NULL.
 * depth[v*W+u] = pt.z of the organised cloud (NaN where no point landed); bgra = its colour bytes
 * (PointXYZRGBA default 0,0,0,255 where empty).  Returns the number of filled pixels. */
static int cvtt_f(float v) { return (v >= -2147483648.f && v < 2147483648.f) ? (int)v : INT_MIN; } /* cvttss2si */

uint64_t oracle_organize(const oracle_params *p, const float *xyz, size_t xyz_stride, const uint8_t *bgra,
                         size_t bgra_stride, size_t n, float units, int zero_nans, const double *tf, float *depth,
                         uint8_t *bgra_out) {
  const int W = p->image_width, H = p->image_height;
  const float fx = (float)p->fx, fy = (float)p->fy, cx = (float)p->cx, cy = (float)p->cy;
  for (size_t j = 0; j < (size_t)W * H; ++j) {
    depth[j] = NAN;
    if (bgra_out) {
      bgra_out[4 * j] = bgra_out[4 * j + 1] = bgra_out[4 * j + 2] = 0;
      bgra_out[4 * j + 3] = 255;
    }
  }
  uint64_t filled = 0;
  for (size_t j = 0; j < n; ++j) {
    float x = xyz[j * xyz_stride], y = xyz[j * xyz_stride + 1], z = xyz[j * xyz_stride + 2];
    if (units != 1) {
      x *= units;
      y *= units;
      z *= units;
    }
    if (zero_nans && x == 0 && y == 0 && z == 0) x = y = z = NAN;
    if (tf && isfinite(x) && isfinite(y) && isfinite(z)) {
      const double dx = x, dy = y, dz = z;
      float o[3];
      for (int r = 0; r < 3; ++r)
        o[r] = (float)(dx * tf[4 * r] + (dy * tf[4 * r + 1] + (dz * tf[4 * r + 2] + tf[4 * r + 3])));
      x = o[0], y = o[1], z = o[2];
    }
    const int u = cvtt_f((x * fx / z) + cx), v = cvtt_f((y * fy / z) + cy);
    if (!(!isnan(z) && z > 0 && u >= 0 && u < W && v >= 0 && v < H)) continue;
    const size_t px = (size_t)v * W + u;
    if (isnan(depth[px]) || depth[px] > z) {
      if (isnan(depth[px])) filled++;
      depth[px] = z;
      if (bgra_out && bgra) memcpy(bgra_out + 4 * px, bgra + j * bgra_stride, 4);
    }
  }
  return filled;
}

/* ------------------------------------------------------------------------------------------
 * getNeighbors tsdf_volume_octree.cpp:796-828, getFxn :655-672, getGradient :681-700,
 * getHessian :703-726.  Order of the 8 neighbours: dx outer, dy, dz inner.  getFxn/getGradient use
 * the octree NODE centre (vox->getCenter), getHessian uses getVoxelCenter (`centers[i]`).
 * `fabs` on a float is double fabs(double) (see above), so each term is a double product. */
static int sgn(float x) { return x > 0 ? 1 : -1; } /* :674-678 */

int oracle_sample(const oracle_params *p, const float *d, const float pt[3], float *val, float grad[3],
                  float hess[9]) {
  int id[3];
  const int nx = p->res[0], ny = p->res[1], nz = p->res[2];
  if (!oracle_voxel_index(p, pt[0], pt[1], pt[2], id)) return 0;
  int xi = id[0], yi = id[1], zi = id[2];
  if (pt[0] < voxel_center(p, 0, xi)) xi -= 1;
  if (pt[1] < voxel_center(p, 1, yi)) yi -= 1;
  if (pt[2] < voxel_center(p, 2, zi)) zi -= 1;
  if (xi < 0 || xi >= nx - 1 || yi < 0 || yi >= ny - 1 || zi < 0 || zi >= nz - 1) return 0;
  float *tx = (float *)malloc(sizeof(float) * (nx + ny + nz)), *ty = tx + nx, *tz = ty + ny;
  oracle_centers(nx, node_size(p, 0), tx);
  oracle_centers(ny, node_size(p, 1), ty);
  oracle_centers(nz, node_size(p, 2), tz);
  const float c = p->size[0] / p->res[0];
  float v = 0, g[3] = {0, 0, 0}, h01 = 0, h02 = 0, h12 = 0;
  for (int dx = 0; dx <= 1; dx++)
    for (int dy = 0; dy <= 1; dy++)
      for (int dz = 0; dz <= 1; dz++) {
        const int i = xi + dx, j = yi + dy, k = zi + dz;
        const float dv = d[((size_t)k * ny + j) * nx + i];
        const float nc[3] = {tx[i], ty[j], tz[k]};                                       /* node centre */
        const float fc[3] = {voxel_center(p, 0, i), voxel_center(p, 1, j), voxel_center(p, 2, k)}; /* centers[i] */
        v += (c - fabs(pt[0] - nc[0])) * (c - fabs(pt[1] - nc[1])) * (c - fabs(pt[2] - nc[2])) * dv;
        g[0] += -sgn(pt[0] - nc[0]) * (c - fabs(pt[1] - nc[1])) * (c - fabs(pt[2] - nc[2])) * dv;
        g[1] += (c - fabs(pt[0] - nc[0])) * -sgn(pt[1] - nc[1]) * (c - fabs(pt[2] - nc[2])) * dv;
        g[2] += (c - fabs(pt[0] - nc[0])) * (c - fabs(pt[1] - nc[1])) * -sgn(pt[2] - nc[2]) * dv;
        h01 += sgn(pt[0] - fc[0]) * sgn(pt[1] - fc[1]) * (c - fabs(pt[2] - fc[2])) * dv;
        h02 += sgn(pt[0] - fc[0]) * (c - fabs(pt[1] - fc[1])) * sgn(pt[2] - fc[2]) * dv;
        h12 += (c - fabs(pt[0] - fc[0])) * sgn(pt[1] - fc[1]) * sgn(pt[2] - fc[2]) * dv;
      }
  free(tx);
  const float c3 = c * c * c;
  if (val) *val = v / c3;
  if (grad)
    for (int k = 0; k < 3; ++k) grad[k] = g[k] / c3;
  if (hess) {
    memset(hess, 0, 9 * sizeof(float));
    hess[1] = hess[3] = h01 / c3;
    hess[2] = hess[6] = h02 / c3;
    hess[5] = hess[7] = h12 / c3;
  }
  return 1;
}

void oracle_sample_batch(const oracle_params *p, const float *d, const float *xyz, size_t n, float *val,
                         float *grad, float *hess, uint8_t *ok) {
  for (size_t i = 0; i < n; ++i) {
    float v = NAN, g[3] = {NAN, NAN, NAN}, h[9];
    for (int k = 0; k < 9; ++k) h[k] = NAN;
    const int r = oracle_sample(p, d, xyz + 3 * i, &v, g, h);
    if (ok) ok[i] = (uint8_t)r;
    if (val) val[i] = v;
    if (grad) memcpy(grad + 3 * i, g, sizeof g);
    if (hess) memcpy(hess + 9 * i, h, sizeof h);
  }
}

/* ------------------------------------------------------------------------------------------
 * Marching cubes: MarchingCubesTSDFOctree, src/lib/marching_cubes_tsdf_octree.cpp:44-236, on top of
 * pcl::MarchingCubes<PointXYZ>::createSurface / interpolateEdge [PCL-recall] with Bourke's tables
 * (oracle/mc_tables.h, generated by tools/gen_mc_tables.py and cross-checked there). */
#include "mc_tables.h"

static uint64_t morton_x_major(uint32_t x, uint32_t y, uint32_t z) {
  uint64_t k = 0;
  for (int l = 20; l >= 0; --l)
    k = (k << 3) | (uint64_t)((((x >> l) & 1u) << 2) | (((y >> l) & 1u) << 1) | ((z >> l) & 1u));
  return k;
}

typedef struct {
  uint64_t key;
  uint32_t x, y, z;
} mc_cell;

static int cmp_cell(const void *a, const void *b) {
  const uint64_t ka = ((const mc_cell *)a)->key, kb = ((const mc_cell *)b)->key;
  return ka < kb ? -1 : ka > kb;
}

/* The voxel arrays may hold only a BOX of the grid: voxels [org, org + dim) per axis, x fastest (the whole grid is
 * org = 0, dim = res).  Full-size volumes are checked box by box (tests/test_fullsize_gpu.py). */
typedef struct {
  int org[3], dim[3];
} mc_window;

static size_t win_index(const mc_window *win, int x, int y, int z) {
  return ((size_t)(z - win->org[2]) * win->dim[1] + (size_t)(y - win->org[1])) * win->dim[0] + (size_t)(x - win->org[0]);
}

/* getGridValue, marching_cubes_tsdf_octree.cpp:91-106 */
static float grid_value(const oracle_params *p, const mc_window *win, const float *d, const float *w, float w_min, int x, int y,
                        int z) {
  const size_t vi = win_index(win, x, y, z);
  if (w[vi] < w_min || fabs(d[vi]) >= 1) return NAN;
  return d[vi] * p->max_dist_neg;
}

/* Returns the number of triangles; writes at most cap triangles (9 floats each, volume frame, before
 * the global transform), rgb 9 bytes per triangle (color_mode 1: setColorByRGB, 2:
 * setColorByConfidence), cell keys (x<<42 | y<<21 | z) per triangle.  Order = octree pre-order. */
static uint64_t march_window(const oracle_params *p, const mc_window *win, const int clo[3], const int chi[3], const float *d,
                             const float *w, const uint8_t *rgb, float w_min, int color_mode, float *verts, uint8_t *rgb_out,
                             uint64_t *cell_out, uint64_t cap) {
  /* setInputTSDF :44-83: the two +- terms at :64-66 cancel, so the bounding box is
   * [centre(voxel 0), centre(voxel res)]; size_voxel_ = (upper - lower) * (1 / res) in float */
  float lower[3], size_voxel[3];
  for (int a = 0; a < 3; ++a) {
    lower[a] = voxel_center(p, a, 0);
    const float upper = voxel_center(p, a, p->res[a]);
    size_voxel[a] = (upper - lower[a]) * (1.0f / (float)p->res[a]);
  }
  const float iso = 0.f;
  size_t n_cells = 0, cap_cells = 1 << 16;
  mc_cell *cells = (mc_cell *)malloc(cap_cells * sizeof(mc_cell));
  /* reconstructVoxel :179-207 candidate test */
  /* (the reference walks every leaf and keeps those 0 < index < res - 1, :186-191; a window keeps the base voxels in
   * [clo, chi) of them) */
  int lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {
    lo[a] = clo[a] > 1 ? clo[a] : 1;
    hi[a] = chi[a] < p->res[a] - 1 ? chi[a] : p->res[a] - 1;
  }
  for (int z = lo[2]; z < hi[2]; ++z)
    for (int y = lo[1]; y < hi[1]; ++y)
      for (int x = lo[0]; x < hi[0]; ++x) {
        const size_t vi = win_index(win, x, y, z);
        if (!(w[vi] >= w_min && fabs(d[vi]) < 1)) continue;
        if (n_cells == cap_cells) {
          cap_cells *= 2;
          cells = (mc_cell *)realloc(cells, cap_cells * sizeof(mc_cell));
        }
        cells[n_cells].key = morton_x_major(x, y, z);
        cells[n_cells].x = x;
        cells[n_cells].y = y;
        cells[n_cells].z = z;
        ++n_cells;
      }
  qsort(cells, n_cells, sizeof(mc_cell), cmp_cell);
  uint64_t n_tri = 0;
  static const int off[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 0, 1}, {0, 0, 1}, {0, 1, 0}, {1, 1, 0}, {1, 1, 1}, {0, 1, 1}};
  static const int edge_v[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
  for (size_t ci = 0; ci < n_cells; ++ci) {
    const int x = cells[ci].x, y = cells[ci].y, z = cells[ci].z;
    float leaf[8];
    int ok = 1;
    for (int k = 0; k < 8 && ok; ++k) { /* getValidNeighborList1D :145-177 */
      leaf[k] = grid_value(p, win, d, w, w_min, x + off[k][0], y + off[k][1], z + off[k][2]);
      if (isnan(leaf[k])) ok = 0;
    }
    if (!ok) continue;
    int cubeindex = 0; /* createSurface [PCL-recall] */
    for (int k = 0; k < 8; ++k)
      if (leaf[k] < iso) cubeindex |= 1 << k;
    if (mc_edge_table[cubeindex] == 0) continue;
    const int idx[3] = {x, y, z};
    float center[3], pc[8][3];
    for (int a = 0; a < 3; ++a) center[a] = lower[a] + size_voxel[a] * (float)idx[a];
    for (int k = 0; k < 8; ++k) {
      pc[k][0] = center[0];
      pc[k][1] = center[1];
      pc[k][2] = center[2];
      if (k & 0x4) pc[k][1] = center[1] + size_voxel[1];
      if (k & 0x2) pc[k][2] = center[2] + size_voxel[2];
      if ((k & 0x1) ^ ((k >> 1) & 0x1)) pc[k][0] = center[0] + size_voxel[0];
    }
    float vl[12][3];
    for (int e = 0; e < 12; ++e)
      if (mc_edge_table[cubeindex] & (1 << e)) { /* interpolateEdge */
        const int a = edge_v[e][0], b = edge_v[e][1];
        const float mu = (iso - leaf[a]) / (leaf[b] - leaf[a]);
        for (int k = 0; k < 3; ++k) vl[e][k] = pc[a][k] + mu * (pc[b][k] - pc[a][k]);
      }
    uint8_t col[3] = {0, 0, 0};
    const size_t vi = win_index(win, x, y, z);
    if (color_mode == 2) { /* :217-224 */
      const float std_dev = (100. - w[vi]) / 100.;
      col[0] = (uint8_t)fmax(0., fmin((1 - std_dev) * 255., 255.));
      col[1] = 0;
      col[2] = (uint8_t)fmax(0., fmin((std_dev)*255., 255.));
    } else if (color_mode == 1 && rgb) { /* :226-231 */
      col[0] = rgb[3 * vi];
      col[1] = rgb[3 * vi + 1];
      col[2] = rgb[3 * vi + 2];
    }
    for (int i = 0; mc_tri_table[cubeindex][i] != -1; i += 3) {
      if (n_tri < cap) {
        for (int v = 0; v < 3; ++v) {
          const int e = mc_tri_table[cubeindex][i + v];
          if (verts) memcpy(verts + 9 * n_tri + 3 * v, vl[e], 3 * sizeof(float));
          if (rgb_out) memcpy(rgb_out + 9 * n_tri + 3 * v, col, 3);
        }
        if (cell_out) cell_out[n_tri] = ((uint64_t)x << 42) | ((uint64_t)y << 21) | (uint64_t)z;
      }
      ++n_tri;
    }
  }
  free(cells);
  return n_tri;
}

uint64_t oracle_march(const oracle_params *p, const float *d, const float *w, const uint8_t *rgb, float w_min,
                      int color_mode, float *verts, uint8_t *rgb_out, uint64_t *cell_out, uint64_t cap) {
  const mc_window win = {{0, 0, 0}, {p->res[0], p->res[1], p->res[2]}};
  const int clo[3] = {1, 1, 1}, chi[3] = {p->res[0] - 1, p->res[1] - 1, p->res[2] - 1};
  return march_window(p, &win, clo, chi, d, w, rgb, w_min, color_mode, verts, rgb_out, cell_out, cap);
}

/* The same for the cells whose base voxel lies in [clo, chi) of a grid of which only the voxels [org, org + dim) are in
 * the arrays (org <= clo and chi + 1 <= org + dim per axis, or -1 is returned): the triangles the whole-grid mesh holds
 * for those cells, in the whole-grid order restricted to them. */
uint64_t oracle_march_box(const oracle_params *p, const int org[3], const int dim[3], const int clo[3], const int chi[3],
                          const float *d, const float *w, const uint8_t *rgb, float w_min, int color_mode, float *verts,
                          uint8_t *rgb_out, uint64_t *cell_out, uint64_t cap) {
  mc_window win;
  for (int a = 0; a < 3; ++a) {
    win.org[a] = org[a];
    win.dim[a] = dim[a];
    const int hi = chi[a] < p->res[a] - 1 ? chi[a] : p->res[a] - 1;
    if (org[a] < 0 || dim[a] < 2 || org[a] > clo[a] || hi + 1 > org[a] + dim[a] || org[a] + dim[a] > p->res[a]) return (uint64_t)-1;
  }
  return march_window(p, &win, clo, chi, d, w, rgb, w_min, color_mode, verts, rgb_out, cell_out, cap);
}
