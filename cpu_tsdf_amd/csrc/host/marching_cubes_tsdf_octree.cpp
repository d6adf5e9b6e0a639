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

// reference: src/lib/marching_cubes_tsdf_octree.cpp:44-83.  The base class is set up exactly as there so that code
// reading it back (getGridResolution, getIsoLevel, ...) sees the reference's values; the kernels take the same
// lower_boundary_ / size_voxel_ from the volume's parameters (tsdf_march.hip), where the two +- terms of :64-66 cancel.
void MarchingCubesTSDFOctree::setInputTSDF(TSDFVolumeOctree::ConstPtr tsdf_volume) {
  tsdf_volume_ = tsdf_volume;
  if (!tsdf_volume_) return;
  int res_x, res_y, res_z;
  tsdf_volume_->getResolution(res_x, res_y, res_z);
  setGridResolution(res_x, res_y, res_z);
  float size_x, size_y, size_z;
  tsdf_volume_->getGridSize(size_x, size_y, size_z);
  pcl::PointCloud<pcl::PointXYZ>::Ptr corner_cloud(new pcl::PointCloud<pcl::PointXYZ>);
  for (int x_i = 0; x_i <= res_x; x_i += res_x)
    for (int y_i = 0; y_i <= res_y; y_i += res_y)
      for (int z_i = 0; z_i <= res_z; z_i += res_z) {
        pcl::PointXYZ center = tsdf_volume_->getVoxelCenter(x_i, y_i, z_i);
        center.x += (x_i == 0 ? -1 : 1) * 0.5 * size_x / res_x + (x_i == 0 ? 1 : -1) * (0.5 * size_x / (double)res_x);
        center.y += (y_i == 0 ? -1 : 1) * 0.5 * size_y / res_y + (y_i == 0 ? 1 : -1) * (0.5 * size_y / (double)res_y);
        center.z += (z_i == 0 ? -1 : 1) * 0.5 * size_z / res_z + (z_i == 0 ? 1 : -1) * (0.5 * size_z / (double)res_z);
        corner_cloud->points.push_back(center);
      }
  corner_cloud->width = (uint32_t)corner_cloud->points.size();
  corner_cloud->height = 1;
  setInputCloud(corner_cloud);
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
