// cpu_tsdf::MarchingCubesTSDFOctree -- MI355X drop-in for the reference mesher
// (include/cpu_tsdf/marching_cubes_tsdf_octree.h:50-100).  Same user-facing calls (setInputTSDF,
// setMinWeight, setColorByRGB, setColorByConfidence, reconstruct(pcl::PolygonMesh&)); the per-cell work
// runs in the HIP marching-cubes kernels, so this class does not derive from pcl::MarchingCubes and
// needs no PCL surface module.  Triangle order, vertex values and colours equal the reference's.
#pragma once

#include <cpu_tsdf/tsdf_volume_octree.h>
#include <pcl/PolygonMesh.h>

#include <vector>

namespace cpu_tsdf {

class MarchingCubesTSDFOctree {
 public:
  MarchingCubesTSDFOctree() : color_by_confidence_(false), color_by_rgb_(false), w_min_(2.5f) {}

  void setInputTSDF(TSDFVolumeOctree::ConstPtr tsdf_volume) { tsdf_volume_ = tsdf_volume; }
  void setColorByConfidence(bool color_by_confidence) { color_by_confidence_ = color_by_confidence; }
  void setColorByRGB(bool color_by_rgb) { color_by_rgb_ = color_by_rgb; }
  void setMinWeight(float w_min) { w_min_ = w_min; }

  // pcl::SurfaceReconstruction::reconstruct: fills output.cloud (PointXYZ, or PointXYZRGB when a colour
  // mode is on) and output.polygons ({3i, 3i+1, 3i+2}); vertices are moved by the volume's global transform.
  void reconstruct(pcl::PolygonMesh &output);
  void reconstruct(pcl::PointCloud<pcl::PointXYZ> &points, std::vector<pcl::Vertices> &polygons);

 private:
  TSDFVolumeOctree::ConstPtr tsdf_volume_;
  bool color_by_confidence_, color_by_rgb_;
  float w_min_;
};

}  // namespace cpu_tsdf
