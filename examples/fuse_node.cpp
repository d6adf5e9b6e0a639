// The reference's README flow (README.md:25-51) against the MI355X drop-in, on every GPU of the node.
//
//   g++ -std=c++14 -fopenmp examples/fuse_node.cpp -Iinclude -Icompat            (or $(pkg-config --cflags pcl_common eigen3))
//       -Lcpu_tsdf_amd/lib -lcpu_tsdf_hip -ltsdf_hip -Wl,-rpath,cpu_tsdf_amd/lib -o fuse_node
//   ./fuse_node 256 8 0,1,2,3,4,5,6,7        resolution, frames, GPU ordinals (repeat one ordinal to split a single GPU)
//
// Apart from setDevices() -- the one call the reference does not have -- every line is what a user of
// sdmiller/cpu_tsdf writes: organised PointXYZRGBA clouds in, integrateCloud, renderView, getFxn, marching cubes, save.
#include <cpu_tsdf/marching_cubes_tsdf_octree.h>
#include <cpu_tsdf/tsdf_volume_octree.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>

// A sphere of radius 0.25 S in front of a wall, seen by a camera on the z axis at distance 2.2 S (pinhole, z depth).
static pcl::PointCloud<pcl::PointXYZRGBA>::Ptr synth_cloud(int W, int H, double f, double S, double yaw) {
  pcl::PointCloud<pcl::PointXYZRGBA>::Ptr cloud(new pcl::PointCloud<pcl::PointXYZRGBA>(W, H));
  cloud->is_dense = false;
  const double cx = W / 2.0 - 0.5, cy = H / 2.0 - 0.5, dist = 2.2 * S, r = 0.25 * S, wall = dist + 0.4 * S;
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
      const double dx = (u - cx) / f, dy = (v - cy) / f;  // ray (dx, dy, 1) * z
      // sphere centred at (0, 0, dist): |z (dx, dy, 1) - (0, 0, dist)|^2 = r^2
      const double a = dx * dx + dy * dy + 1, b = -2 * dist, c = dist * dist - r * r, disc = b * b - 4 * a * c;
      double z = wall;
      if (disc >= 0) z = (-b - std::sqrt(disc)) / (2 * a);
      pcl::PointXYZRGBA &p = (*cloud)(u, v);
      p.x = (float)(dx * z), p.y = (float)(dy * z), p.z = (float)z;
      p.r = (unsigned char)(u & 255), p.g = (unsigned char)(v & 255), p.b = (unsigned char)(int)(yaw * 40), p.a = 255;
    }
  return cloud;
}

int main(int argc, char **argv) {
  const int res = argc > 1 ? std::atoi(argv[1]) : 128, frames = argc > 2 ? std::atoi(argv[2]) : 4;
  std::vector<int> devices;
  if (argc > 3)
    for (const char *s = argv[3]; *s;) {
      devices.push_back(std::atoi(s));
      while (*s && *s != ',') ++s;
      if (*s == ',') ++s;
    }
  const int W = 640, H = 480;
  const double S = res / 256.0, f = 525.0;
  cpu_tsdf::TSDFVolumeOctree::Ptr tsdf(new cpu_tsdf::TSDFVolumeOctree);
  tsdf->setGridSize((float)S, (float)S, (float)S);
  tsdf->setResolution(res, res, res);
  tsdf->setImageSize(W, H);
  tsdf->setCameraIntrinsics(f, f, W / 2.0 - 0.5, H / 2.0 - 0.5);
  tsdf->setSensorDistanceBounds(0.f, (float)(3 * S));
  tsdf->setIntegrateColor(true);
  tsdf->setDevices(devices);  // <- the only line the reference does not have
  tsdf->reset();
  if (!tsdf->handle()) return 1;
  pcl::PointCloud<pcl::Normal> no_normals;
  Eigen::Affine3d last = Eigen::Affine3d::Identity();
  for (int i = 0; i < frames; ++i) {
    const double yaw = 0.15 * i;  // the camera swings around the y axis, always looking at the volume's centre
    Eigen::Matrix4d m = Eigen::Matrix4d::Identity();
    m(0, 0) = std::cos(yaw), m(0, 2) = std::sin(yaw), m(2, 0) = -std::sin(yaw), m(2, 2) = std::cos(yaw);
    m(0, 3) = -2.2 * S * std::sin(yaw), m(2, 3) = -2.2 * S * std::cos(yaw);
    Eigen::Affine3d pose;
    pose.matrix() = m;
    if (!tsdf->integrateCloud(*synth_cloud(W, H, f, S, yaw), no_normals, pose)) return 2;
    last = pose;
  }
  pcl::PointCloud<pcl::PointNormal>::Ptr view = tsdf->renderView(last, 2);
  size_t hits = 0;
  for (size_t i = 0; i < view->size(); ++i) hits += std::isfinite(view->points[i].z) ? 1 : 0;
  float d = std::numeric_limits<float>::quiet_NaN();
  const bool inside = tsdf->getFxn(pcl::PointXYZ(0.f, 0.f, (float)(-0.25 * S)), d);  // a point on the sphere's front
  cpu_tsdf::MarchingCubesTSDFOctree mc;
  mc.setInputTSDF(tsdf);
  mc.setMinWeight(2);
  mc.setColorByRGB(true);
  pcl::PolygonMesh mesh;
  mc.reconstruct(mesh);
  if (argc > 4) tsdf->save(argv[4]);
  std::printf("{\"devices\": %zu, \"res\": %d, \"frames\": %d, \"render_hits\": %zu, \"getFxn_ok\": %d, \"getFxn\": %.9g, "
              "\"triangles\": %zu, \"cloud_bytes\": %zu}\n",
              devices.size(), res, frames, hits, inside ? 1 : 0, (double)d, mesh.polygons.size(), mesh.cloud.data.size());
  return 0;
}
