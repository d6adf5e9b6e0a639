// Host shell of the MI355X drop-in: cpu_tsdf::MarchingCubesTSDFOctree on top of the C ABI.
// Mirrors performReconstruction of the reference (src/lib/marching_cubes_tsdf_octree.cpp:108-143): build
// the vertex cloud (3 vertices per triangle, no sharing), move it by the volume's global transform,
// pack it into the PolygonMesh blob, polygons = {3i, 3i+1, 3i+2}.
#include <cpu_tsdf/marching_cubes_tsdf_octree.h>
#include <pcl/common/transforms.h>
#include <pcl/console/print.h>
#include <pcl/conversions.h>

#include <vector>

namespace cpu_tsdf {

// setInputTSDF (reference: src/lib/marching_cubes_tsdf_octree.cpp:44-83) leaves the pcl::MarchingCubes base in the
// state the reference leaves it in, so that code reading it back (getGridResolution, getIsoLevel, the input cloud's
// bounding box) sees the same values.  What that state IS: the "input cloud" is the eight corners of the box spanned by
// the centres of voxel (0, 0, 0) and of voxel (res_x, res_y, res_z) -- the reference adds a half-voxel offset to each
// centre and subtracts the same number again (:64-66), which leaves the centres themselves; getVoxelCenter is separable
// per axis, so the eight points are the combinations of those two -- no grid extension, iso level 0, and size_voxel_
// the box divided by the resolution.  The kernels derive the same lower boundary / voxel size from the volume's
// parameters (tsdf_march.hip).
void MarchingCubesTSDFOctree::setInputTSDF(TSDFVolumeOctree::ConstPtr tsdf_volume) {
  tsdf_volume_ = tsdf_volume;
  if (!tsdf_volume_) return;
  int res[3];
  tsdf_volume_->getResolution(res[0], res[1], res[2]);
  setGridResolution(res[0], res[1], res[2]);
  const pcl::PointXYZ first = tsdf_volume_->getVoxelCenter(0, 0, 0);
  const pcl::PointXYZ beyond = tsdf_volume_->getVoxelCenter(res[0], res[1], res[2]);
  pcl::PointCloud<pcl::PointXYZ>::Ptr corners(new pcl::PointCloud<pcl::PointXYZ>);
  for (int c = 0; c < 8; ++c)  // x slowest, z fastest, like the reference's three loops
    corners->points.push_back(pcl::PointXYZ((c & 4) ? beyond.x : first.x, (c & 2) ? beyond.y : first.y, (c & 1) ? beyond.z : first.z));
  corners->width = (uint32_t)corners->points.size();
  corners->height = 1;
  setInputCloud(corners);
  setPercentageExtendGrid(0);
  setIsoLevel(0.f);
  getBoundingBox();
  size_voxel_ = (upper_boundary_ - lower_boundary_) * Eigen::Array3f(res_x_, res_y_, res_z_).inverse();
}

static bool run_march(const TSDFVolumeOctree::ConstPtr &vol, float w_min, int mode, std::vector<float> &verts,
                      std::vector<unsigned char> &rgb) {
  verts.clear();
  rgb.clear();
  if (!vol || !vol->handle()) {
    PCL_ERROR("[cpu_tsdf::MarchingCubesTSDFOctree::reconstruct] no TSDF volume set (or reset() not called)\n");
    return false;
  }
  if (!vol->cubicForQueries("MarchingCubesTSDFOctree::reconstruct")) return false;
  uint64_t n_tri = 0;
  int rc = tsdf_hip_march(vol->handle(), w_min, mode, &n_tri);
  if (rc == 0 && n_tri) {
    verts.resize((size_t)n_tri * 9);
    if (mode) rgb.resize((size_t)n_tri * 9);
    rc = tsdf_hip_march_fetch(vol->handle(), verts.data(), mode ? rgb.data() : nullptr, nullptr);
  }
  if (rc) {
    PCL_ERROR("[cpu_tsdf::MarchingCubesTSDFOctree::reconstruct] %s: %s\n", tsdf_hip_error_string(rc),
              tsdf_hip_last_error());
    verts.clear();
    rgb.clear();
    return false;
  }
  return true;
}

static void fill_polygons(size_t n_vertices, std::vector<pcl::Vertices> &polygons) {
  polygons.resize(n_vertices / 3);
  for (size_t i = 0; i < polygons.size(); ++i) {
    pcl::Vertices v;
    v.vertices.resize(3);
    for (int j = 0; j < 3; ++j) v.vertices[j] = static_cast<int>(i) * 3 + j;
    polygons[i] = v;
  }
}

void MarchingCubesTSDFOctree::performReconstruction(pcl::PolygonMesh &output) {
  output.polygons.clear();
  // color_by_confidence_ wins over color_by_rgb_ (marching_cubes_tsdf_octree.cpp:215-231)
  const int mode = color_by_confidence_ ? 2 : (color_by_rgb_ ? 1 : 0);
  std::vector<float> verts;
  std::vector<unsigned char> rgb;
  run_march(tsdf_volume_, w_min_, mode, verts, rgb);
  const size_t n = verts.size() / 3;
  const Eigen::Affine3d g = tsdf_volume_ ? tsdf_volume_->getGlobalTransform() : Eigen::Affine3d::Identity();
  if (mode) {
    pcl::PointCloud<pcl::PointXYZRGB> cloud;
    cloud.points.resize(n);
    cloud.width = (uint32_t)n;
    cloud.height = 1;
    for (size_t i = 0; i < n; ++i) {
      pcl::PointXYZRGB &p = cloud.points[i];
      p.x = verts[3 * i];
      p.y = verts[3 * i + 1];
      p.z = verts[3 * i + 2];
      p.r = rgb[3 * i];
      p.g = rgb[3 * i + 1];
      p.b = rgb[3 * i + 2];
    }
    pcl::transformPointCloud(cloud, cloud, g);
    pcl::toPCLPointCloud2(cloud, output.cloud);
  } else {
    pcl::PointCloud<pcl::PointXYZ> cloud;
    cloud.points.resize(n);
    cloud.width = (uint32_t)n;
    cloud.height = 1;
    for (size_t i = 0; i < n; ++i) {
      pcl::PointXYZ &p = cloud.points[i];
      p.x = verts[3 * i];
      p.y = verts[3 * i + 1];
      p.z = verts[3 * i + 2];
    }
    pcl::transformPointCloud(cloud, cloud, g);
    pcl::toPCLPointCloud2(cloud, output.cloud);
  }
  fill_polygons(n, output.polygons);
}

void MarchingCubesTSDFOctree::performReconstruction(pcl::PointCloud<pcl::PointXYZ> &points,
                                                    std::vector<pcl::Vertices> &polygons) {
  std::vector<float> verts;
  std::vector<unsigned char> rgb;
  run_march(tsdf_volume_, w_min_, 0, verts, rgb);
  const size_t n = verts.size() / 3;
  points.points.resize(n);
  points.width = (uint32_t)n;
  points.height = 1;
  for (size_t i = 0; i < n; ++i) {
    points.points[i].x = verts[3 * i];
    points.points[i].y = verts[3 * i + 1];
    points.points[i].z = verts[3 * i + 2];
  }
  if (tsdf_volume_) pcl::transformPointCloud(points, points, tsdf_volume_->getGlobalTransform());
  fill_polygons(n, polygons);
}

}  // namespace cpu_tsdf
