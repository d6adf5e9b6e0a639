// ref_capi.cpp -- flat C wrapper around the cpu_tsdf C++ API, for ctypes.
//
// TEST INFRASTRUCTURE.  Compiled twice from this one source:
//   * against the REFERENCE's own headers and sources (-I/root/reference/include, -DCT_REFERENCE)
//     -> oracle/_ref/libcpu_tsdf_ref.so: the parity oracle and the "reference" CPU baseline;
//   * against THIS repo's drop-in headers (include/cpu_tsdf/...) -> the drop-in shell under test.
// That the same driver compiles against both is the drop-in claim; tests then compare outputs.
// Only public API is used, except ct_dump_dense which (reference build only) walks the public
// `octree_` member (tsdf_volume_octree.h:298) to read back voxels.
#include <cpu_tsdf/marching_cubes_tsdf_octree.h>
#include <cpu_tsdf/tsdf_volume_octree.h>

#include <chrono>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

using cpu_tsdf::TSDFVolumeOctree;

struct CtVolume {
  TSDFVolumeOctree::Ptr vol;
  pcl::PolygonMesh mesh;
  double fx = 525, fy = 525, cx = 320, cy = 240;
};

static Eigen::Affine3d to_affine(const double *m16) {
  Eigen::Affine3d t = Eigen::Affine3d::Identity();
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) t.matrix()(r, c) = m16[4 * r + c];
  return t;
}

extern "C" {

void *ct_create() {
  CtVolume *c = new CtVolume;
  c->vol.reset(new TSDFVolumeOctree);
  return c;
}
void ct_destroy(void *h) { delete (CtVolume *)h; }
#define V(h) (((CtVolume *)(h))->vol)

void ct_set_resolution(void *h, int x, int y, int z) { V(h)->setResolution(x, y, z); }
void ct_set_grid_size(void *h, float x, float y, float z) { V(h)->setGridSize(x, y, z); }
void ct_set_image_size(void *h, int w, int ht) { V(h)->setImageSize(w, ht); }
void ct_set_intrinsics(void *h, double fx, double fy, double cx, double cy) {
  CtVolume *c = (CtVolume *)h;
  c->fx = fx;
  c->fy = fy;
  c->cx = cx;
  c->cy = cy;
  c->vol->setCameraIntrinsics(fx, fy, cx, cy);
}
void ct_set_sensor_bounds(void *h, float zmin, float zmax) { V(h)->setSensorDistanceBounds(zmin, zmax); }
void ct_set_trunc(void *h, float pos, float neg) { V(h)->setDepthTruncationLimits(pos, neg); }
void ct_set_max_weight(void *h, float w) { V(h)->setWeightTruncationLimit(w); }
void ct_set_max_voxel_size(void *h, float x, float y, float z) { V(h)->setMaxVoxelSize(x, y, z); }
void ct_set_integrate_color(void *h, int f) { V(h)->setIntegrateColor(f != 0); }
void ct_set_color_mode(void *h, const char *mode) { V(h)->setColorMode(mode); }
#ifdef CT_REFERENCE
// the reference's own colour conversions (octree.cpp:436-527), for pinning the oracle's restatement
void ct_rgb2lab_many(const uint8_t *rgb, size_t n, float *lab) {
  for (size_t i = 0; i < n; ++i)
    cpu_tsdf::RGB2LAB(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], lab[3 * i], lab[3 * i + 1], lab[3 * i + 2]);
}
void ct_lab2rgb_many(const float *lab, size_t n, uint8_t *rgb) {
  for (size_t i = 0; i < n; ++i)
    cpu_tsdf::LAB2RGB(lab[3 * i], lab[3 * i + 1], lab[3 * i + 2], rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
}
#endif
void ct_set_num_random_splits(void *h, int n) { V(h)->setNumRandomSplts(n); }
void ct_set_global_transform(void *h, const double *m16) { V(h)->setGlobalTransform(to_affine(m16)); }
void ct_reset(void *h) { V(h)->reset(); }
void ct_get_resolution(void *h, int *r) { V(h)->getResolution(r[0], r[1], r[2]); }
void ct_get_trunc(void *h, float *pn) { V(h)->getDepthTruncationLimits(pn[0], pn[1]); }

// Organised PointXYZRGBA cloud from planar depth + bgra, the way the reference CLI organises its input
// (src/prog/integrate.cpp:592-635): z = depth, x/y back-projected through the intrinsics, NaN where
// there is no return.  Returns the wall time of integrateCloud alone in seconds.
double ct_integrate(void *h, const float *depth, const uint8_t *bgra, int W, int H, const double *trans16) {
  CtVolume *c = (CtVolume *)h;
  pcl::PointCloud<pcl::PointXYZRGBA> cloud(W, H);
  cloud.is_dense = false;
  const float nan = std::numeric_limits<float>::quiet_NaN();
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
      pcl::PointXYZRGBA &p = cloud(u, v);
      const float z = depth[(size_t)v * W + u];
      if (std::isnan(z)) {
        p.x = p.y = p.z = nan;
      } else {
        p.z = z;
        p.x = (float)((u - c->cx) * z / c->fx);
        p.y = (float)((v - c->cy) * z / c->fy);
      }
      if (bgra) {
        const uint8_t *q = bgra + 4 * ((size_t)v * W + u);
        p.b = q[0];
        p.g = q[1];
        p.r = q[2];
        p.a = q[3];
      }
    }
  pcl::PointCloud<pcl::Normal> empty_normals;
  const Eigen::Affine3d trans = to_affine(trans16);
  const auto t0 = std::chrono::steady_clock::now();
  c->vol->integrateCloud(cloud, empty_normals, trans);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// renderView: out is (H/ds)*(W/ds)*8 floats: x,y,z, nx,ny,nz, curvature, 0 (camera frame, as returned).
double ct_render_view(void *h, const double *trans16, int ds, float *out) {
  const auto t0 = std::chrono::steady_clock::now();
  pcl::PointCloud<pcl::PointNormal>::Ptr cloud = V(h)->renderView(to_affine(trans16), ds);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (size_t i = 0; i < cloud->size(); ++i) {
    const pcl::PointNormal &p = cloud->points[i];
    float *o = out + 8 * i;
    o[0] = p.x;
    o[1] = p.y;
    o[2] = p.z;
    o[3] = p.normal_x;
    o[4] = p.normal_y;
    o[5] = p.normal_z;
    o[6] = p.curvature;
    o[7] = 0.f;
  }
  return dt;
}

// renderColoredView (tsdf_volume_octree.cpp:426-450): out as ct_render_view, rgb = 3 bytes r,g,b per pixel.
void ct_render_colored_view(void *h, const double *trans16, int ds, float *out, uint8_t *rgb) {
  pcl::PointCloud<pcl::PointXYZRGBNormal>::Ptr cloud = V(h)->renderColoredView(to_affine(trans16), ds);
  for (size_t i = 0; i < cloud->size(); ++i) {
    const pcl::PointXYZRGBNormal &p = cloud->points[i];
    float *o = out + 8 * i;
    o[0] = p.x;
    o[1] = p.y;
    o[2] = p.z;
    o[3] = p.normal_x;
    o[4] = p.normal_y;
    o[5] = p.normal_z;
    o[6] = o[7] = 0.f;
    rgb[3 * i] = p.r;
    rgb[3 * i + 1] = p.g;
    rgb[3 * i + 2] = p.b;
  }
}

// getFxn / getGradient / getHessian per point (tsdf_volume_octree.cpp:655-726).
void ct_sample(void *h, const float *xyz, size_t n, float *val, float *grad, float *hess, uint8_t *ok) {
  for (size_t i = 0; i < n; ++i) {
    const pcl::PointXYZ pt(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    float v = std::numeric_limits<float>::quiet_NaN();
    Eigen::Vector3f g(v, v, v);
    Eigen::Matrix3f hm;
    for (int k = 0; k < 9; ++k) hm.data()[k] = v;
    const bool o1 = V(h)->getFxn(pt, v);
    const bool o2 = V(h)->getGradient(pt, g);
    const bool o3 = V(h)->getHessian(pt, hm);
    ok[i] = (uint8_t)((o1 ? 1 : 0) | (o2 ? 2 : 0) | (o3 ? 4 : 0));
    val[i] = v;
    for (int k = 0; k < 3; ++k) grad[3 * i + k] = g(k);
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) hess[9 * i + 3 * r + cc] = hm(r, cc);
  }
}

// MarchingCubesTSDFOctree::reconstruct; color_mode 0 none, 1 setColorByRGB, 2 setColorByConfidence.
// Returns the number of mesh vertices (3 per triangle); *seconds = reconstruct() wall time.
uint64_t ct_march(void *h, float w_min, int color_mode, double *seconds) {
  CtVolume *c = (CtVolume *)h;
  cpu_tsdf::MarchingCubesTSDFOctree mc;
  mc.setMinWeight(w_min);
  mc.setInputTSDF(c->vol);
  if (color_mode == 1) mc.setColorByRGB(true);
  if (color_mode == 2) mc.setColorByConfidence(true);
  // used the way PCL code uses any surface reconstruction: through the base class (the reference's mesher IS a
  // pcl::MarchingCubes<pcl::PointXYZ>, marching_cubes_tsdf_octree.h:50; the drop-in must be one too)
  pcl::MarchingCubes<pcl::PointXYZ> &base = mc;
  int rx = 0, ry = 0, rz = 0, vx = 0, vy = 0, vz = 0;
  base.getGridResolution(rx, ry, rz);
  c->vol->getResolution(vx, vy, vz);
  if (rx != vx || ry != vy || rz != vz || base.getIsoLevel() != 0.f || base.getPercentageExtendGrid() != 0.f) return ~0ull;
  const auto t0 = std::chrono::steady_clock::now();
  base.reconstruct(c->mesh);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return (uint64_t)c->mesh.cloud.width * c->mesh.cloud.height;
}

// Copies the last mesh: verts n*3 floats, rgb n*3 bytes (if the cloud has colour and rgb != NULL),
// polygons n_poly*3 uint32.  Returns the polygon count.
uint64_t ct_march_fetch(void *h, float *verts, uint8_t *rgb, uint32_t *polys) {
  CtVolume *c = (CtVolume *)h;
  const pcl::PCLPointCloud2 &pc = c->mesh.cloud;
  const size_t n = (size_t)pc.width * pc.height;
  const bool colored = pc.point_step == sizeof(pcl::PointXYZRGB);
  for (size_t i = 0; i < n; ++i) {
    const uint8_t *p = pc.data.data() + i * pc.point_step;
    if (verts) std::memcpy(verts + 3 * i, p, 12);
    if (rgb && colored) {  // PointXYZRGB bytes at offset 16: b, g, r, a
      rgb[3 * i + 0] = p[18];
      rgb[3 * i + 1] = p[17];
      rgb[3 * i + 2] = p[16];
    }
  }
  if (polys)
    for (size_t i = 0; i < c->mesh.polygons.size(); ++i)
      for (int k = 0; k < 3; ++k) polys[3 * i + k] = c->mesh.polygons[i].vertices[k];
  return c->mesh.polygons.size();
}

int ct_mesh_has_color(void *h) { return ((CtVolume *)h)->mesh.cloud.point_step == sizeof(pcl::PointXYZRGB); }

void ct_save(void *h, const char *path) { V(h)->save(path); }
void ct_load(void *h, const char *path) { V(h)->load(path); }

void ct_voxel_center(void *h, int i, int j, int k, float *out) {
  const pcl::PointXYZ p = V(h)->getVoxelCenter(i, j, k);
  out[0] = p.x;
  out[1] = p.y;
  out[2] = p.z;
}
int ct_voxel_index(void *h, float x, float y, float z, int *idx) {
  return V(h)->getVoxelIndex(x, y, z, idx[0], idx[1], idx[2]) ? 1 : 0;
}

#ifdef CT_REFERENCE
// Reference build only: sample the octree at every fine-grid voxel centre.  leaf_size[i] = size of the
// leaf that contains the centre (== finest size where the octree is fully refined); ctr = that leaf's
// own centre.  Arrays are [z][y][x].
void ct_dump_dense(void *h, float *d, float *w, uint8_t *rgb, float *leaf_size, float *leaf_ctr) {
  int rx, ry, rz;
  V(h)->getResolution(rx, ry, rz);
#pragma omp parallel for
  for (int k = 0; k < rz; ++k)
    for (int j = 0; j < ry; ++j)
      for (int i = 0; i < rx; ++i) {
        const size_t vi = ((size_t)k * ry + j) * rx + i;
        const pcl::PointXYZ c = V(h)->getVoxelCenter(i, j, k);
        const cpu_tsdf::OctreeNode *leaf = V(h)->octree_->getContainingVoxel(c.x, c.y, c.z);
        if (!leaf) {
          d[vi] = w[vi] = std::numeric_limits<float>::quiet_NaN();
          if (leaf_size) leaf_size[vi] = 0;
          continue;
        }
        leaf->getData(d[vi], w[vi]);
        if (rgb) leaf->getRGB(rgb[3 * vi], rgb[3 * vi + 1], rgb[3 * vi + 2]);
        if (leaf_size) leaf_size[vi] = leaf->getMinSize();
        if (leaf_ctr) leaf->getCenter(leaf_ctr[3 * vi], leaf_ctr[3 * vi + 1], leaf_ctr[3 * vi + 2]);
      }
}

uint64_t ct_num_leaves(void *h) {
  std::vector<cpu_tsdf::OctreeNode::Ptr> leaves;
  V(h)->octree_->getLeaves(leaves);
  return leaves.size();
}
int ct_is_reference() { return 1; }
#else
int ct_is_reference() { return 0; }
// Drop-in build only: raw voxel readback through the shell's extension (arrays [z][y][x]).
int ct_download(void *h, float *d, float *w, uint8_t *rgb) {
  int rx, ry, rz;
  V(h)->getResolution(rx, ry, rz);
  return V(h)->downloadBlock(0, 0, 0, rx, ry, rz, d, w, rgb) ? 1 : 0;
}
// Drop-in build only: the multi-GPU extension (takes effect at the next reset / load).
void ct_set_devices(void *h, const int *devices, int n) { V(h)->setDevices(std::vector<int>(devices, devices + n)); }
// Drop-in build only: replicate the reference's frustum cull where it is not a no-op (the reference always culls).
void ct_set_reference_cull(void *h, int flag) { V(h)->setReferenceCull(flag != 0); }
// Drop-in build only: integrateCloud calls two per kernel sweep where the poses allow it (same voxels).
void ct_set_frame_pairing(void *h, int flag) { V(h)->setFramePairing(flag != 0); }
int ct_get_frame_pairing(void *h) { return V(h)->getFramePairing() ? 1 : 0; }
int ct_get_num_devices(void *h) { return (int)V(h)->getDevices().size(); }
#endif

}  // extern "C"
