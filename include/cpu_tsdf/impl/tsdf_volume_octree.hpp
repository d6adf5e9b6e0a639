// integrateCloud template of the MI355X drop-in: strips the organised PCL cloud down to the two planar
// images the kernel reads (pt.z, and b,g,r,a when colour is on) and forwards to integratePlanar.
// The reference reads exactly these fields (include/cpu_tsdf/impl/tsdf_volume_octree.hpp:150-159,206);
// like there, PointT needs r,g,b members.
#pragma once

#include <vector>

namespace cpu_tsdf {

template <typename PointT, typename NormalT>
bool TSDFVolumeOctree::integrateCloud(const pcl::PointCloud<PointT> &cloud, const pcl::PointCloud<NormalT> &,
                                      const Eigen::Affine3d &trans) {
  const size_t n = cloud.points.size();
  std::vector<float> depth(n);
  for (size_t i = 0; i < n; ++i) depth[i] = cloud.points[i].z;
  std::vector<unsigned char> bgra;
  if (p_.integrate_color) {
    bgra.resize(4 * n);
    for (size_t i = 0; i < n; ++i) {
      const PointT &pt = cloud.points[i];
      bgra[4 * i + 0] = pt.b;
      bgra[4 * i + 1] = pt.g;
      bgra[4 * i + 2] = pt.r;
      bgra[4 * i + 3] = 255;
    }
  }
  return integratePlanar(depth.data(), bgra.empty() ? nullptr : bgra.data(), (int)cloud.width, (int)cloud.height,
                         trans);
}

}  // namespace cpu_tsdf
