// integrateCloud template of the MI355X drop-in: strips the organised PCL cloud down to the two planar images the
// kernel reads (pt.z, and b,g,r,a when colour is on) -- straight into the volume's pinned staging slot
// (tsdf_hip_frame_begin) -- and queues upload + integrate
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
  // ONE thread, on purpose.  The strip reads 32-byte points and writes 8 bytes per pixel: memory-bound, 0.5-0.6 ms for a
  // 640x480 cloud on one core -- a twentieth of the kernel it runs ahead of (the call returns once the frame is staged).  Until
  // round 6 this loop was an `omp parallel for` over the caller's default team: on a host whose container grants fewer CPUs
  // than it shows (a 256-thread team under a CPU quota) the team's barrier waited for the scheduler, 104 ms per call --
  // the drop-in ran at 8 frames/s where the kernel does 78 (profiles/r06_cpp_path_timing_before.json).
  uint32_t *bgra32 = reinterpret_cast<uint32_t *>(bgra);  // the slot is 16-byte aligned
  if (bgra32) {
    for (long i = 0; i < n; ++i) {
      depth[i] = pts[i].z;
      bgra32[i] = (uint32_t)pts[i].b | ((uint32_t)pts[i].g << 8) | ((uint32_t)pts[i].r << 16) | 0xff000000u;  // PCL order b, g, r, a
    }
  } else {
    for (long i = 0; i < n; ++i) depth[i] = pts[i].z;
  }
  return commitFrame(trans);
}

}  // namespace cpu_tsdf
