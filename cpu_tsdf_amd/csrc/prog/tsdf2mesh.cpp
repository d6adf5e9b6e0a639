// `tsdf2mesh` -- the reference's volume-to-mesh program (src/prog/tsdf2mesh.cpp:50-74) on the MI355X path:
// `tsdf2mesh foo.vol foo.ply` loads a volume written by TSDFVolumeOctree::save (either implementation's),
// runs marching cubes on the GPU with the class defaults (min weight 2.5, no colour) and writes a binary PLY.
#include <cpu_tsdf/marching_cubes_tsdf_octree.h>
#include <cpu_tsdf/tsdf_interface.h>
#include <cpu_tsdf/tsdf_volume_octree.h>

#include <pcl/console/print.h>
#include <pcl/io/ply_io.h>

#include <string>

int main(int argc, char **argv) {
  if (argc < 3) {
    PCL_INFO("Renders a mesh from a TSDF volume saved with TSDFVolumeOctree::save.\nUsage: %s foo.vol foo.ply\n", argv[0]);
    return 1;
  }
  cpu_tsdf::TSDFVolumeOctree::Ptr tsdf(new cpu_tsdf::TSDFVolumeOctree);
  tsdf->load(argv[1]);
  if (!tsdf->handle()) return 1;
  cpu_tsdf::MarchingCubesTSDFOctree mc;
  mc.setInputTSDF(tsdf);
  mc.setColorByConfidence(false);
  mc.setColorByRGB(false);
  pcl::PolygonMesh mesh;
  mc.reconstruct(mesh);
  return pcl::io::savePLYFileBinary(argv[2], mesh);
}
