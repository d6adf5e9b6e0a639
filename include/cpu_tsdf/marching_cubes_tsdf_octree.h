// cpu_tsdf::MarchingCubesTSDFOctree -- MI355X drop-in for the reference mesher
// (include/cpu_tsdf/marching_cubes_tsdf_octree.h:50-100).  As there, it IS a pcl::MarchingCubes<pcl::PointXYZ>: code
// that holds it as a pcl::SurfaceReconstruction / pcl::MarchingCubes reference, or calls the inherited setters
// (setIsoLevel, setGridResolution, setPercentageExtendGrid, setInputCloud -- used by the reference's own setInputTSDF,
// src/lib/marching_cubes_tsdf_octree.cpp:71-78), keeps compiling; reconstruct(pcl::PolygonMesh&) is the inherited
// entry point and lands in performReconstruction below, which runs the HIP marching-cubes kernels instead of walking
// octree leaves.  Triangle order, vertex values and colours equal the reference's.
#pragma once

#include <cpu_tsdf/tsdf_volume_octree.h>
#include <pcl/PolygonMesh.h>
#include <pcl/surface/marching_cubes.h>

#include <vector>

namespace cpu_tsdf {

class MarchingCubesTSDFOctree : public pcl::MarchingCubes<pcl::PointXYZ> {
 public:
  MarchingCubesTSDFOctree()
      : pcl::MarchingCubes<pcl::PointXYZ>(), color_by_confidence_(false), color_by_rgb_(false), w_min_(2.5f) {}

  // Mirrors the reference (:44-83): remembers the volume and dresses the base class the same way -- grid resolution,
  // the 8-corner "input cloud", no grid extension, iso level 0, bounding box and size_voxel_.
  void setInputTSDF(TSDFVolumeOctree::ConstPtr tsdf_volume);
  void setColorByConfidence(bool color_by_confidence) { color_by_confidence_ = color_by_confidence; }
  void setColorByRGB(bool color_by_rgb) { color_by_rgb_ = color_by_rgb; }
  void setMinWeight(float w_min) { w_min_ = w_min; }

  using pcl::MarchingCubes<pcl::PointXYZ>::reconstruct;  // reconstruct(PolygonMesh&), reconstruct(points, polygons)

 protected:
  void voxelizeData() override {}  // as in the reference (:86-90): nothing to voxelize, the TSDF is the grid
  // fills output.cloud (PointXYZ, or PointXYZRGB when a colour mode is on) and output.polygons ({3i, 3i+1, 3i+2});
  // vertices are moved by the volume's global transform (:108-143)
  void performReconstruction(pcl::PolygonMesh &output) override;
  void performReconstruction(pcl::PointCloud<pcl::PointXYZ> &points, std::vector<pcl::Vertices> &polygons) override;

  TSDFVolumeOctree::ConstPtr tsdf_volume_;
  bool color_by_confidence_, color_by_rgb_;
  float w_min_;
};

}  // namespace cpu_tsdf
