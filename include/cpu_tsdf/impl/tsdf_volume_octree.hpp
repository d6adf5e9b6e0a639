// integrateCloud template of the MI355X drop-in: strips the organised PCL cloud down to the two planar images the
// kernel reads (pt.z, and b,g,r,a when colour is on) -- straight into the volume's pinned staging slot
// (tsdf_hip_frame_begin), in parallel when the caller compiles with OpenMP -- and queues upload + integrate
// (tsdf_hip_frame_commit).  The call returns once the frame is staged: the upload overlaps the previous frame's kernel
// and every later call on the volume is ordered after it, so a stream of clouds runs at the kernel's rate.
// The reference reads exactly these fields (include/cpu_tsdf/impl/tsdf_volume_octree.hpp:150-159,206); like there,
// PointT needs r,g,b members.
#pragma once

namespace cpu_tsdf {

template <typename PointT, typename NormalT>
bool TSDFVolumeOctree::integrateCloud(const pcl::PointCloud<PointT> &cloud, const pcl::PointCloud<NormalT> &,
                                      const Eigen::Affine3d &trans) {
  float *depth = nullptr;
  unsigned char *bgra = nullptr;
  // the staging slot holds image_width x image_height pixels: a cloud whose point count disagrees with its own
  // width x height would overrun it (more points) or leave pixels of an older frame in it (fewer)
  if (cloud.points.size() != (size_t)cloud.width * (size_t)cloud.height) {
    PCL_ERROR("[cpu_tsdf::TSDFVolumeOctree::integrateCloud] cloud has %zu points but says %u x %u\n", cloud.points.size(),
              (unsigned)cloud.width, (unsigned)cloud.height);
    return false;
  }
  if (!beginFrame((int)cloud.width, (int)cloud.height, &depth, &bgra)) return false;
  const long n = (long)cloud.points.size();
  const PointT *pts = n ? &cloud.points[0] : nullptr;
#pragma omp parallel for schedule(static) if (n > 65536)  // (the caller's OpenMP settings decide the team)
  for (long i = 0; i < n; ++i) {
    depth[i] = pts[i].z;
    if (bgra) {
      bgra[4 * i + 0] = pts[i].b;
      bgra[4 * i + 1] = pts[i].g;
      bgra[4 * i + 2] = pts[i].r;
      bgra[4 * i + 3] = 255;
    }
  }
  return commitFrame(trans);
}

}  // namespace cpu_tsdf
