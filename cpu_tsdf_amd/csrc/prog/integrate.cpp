// `integrate` -- the reference's sequence-to-mesh program (src/prog/integrate.cpp) on the MI355X path.
// Same command line, same file conventions, same outputs (out/mesh.ply, optional out/volume.tsdf); the
// per-frame work that the reference does in host loops -- unit scaling, zero -> NaN, world -> camera,
// z-buffer reprojection of the unorganised cloud, integrateCloud, marching cubes -- runs in HIP kernels
// behind cpu_tsdf::TSDFVolumeOctree / MarchingCubesTSDFOctree (include/cpu_tsdf/).
//
// What is reproduced from the reference, by line:
//   options and their defaults                                  integrate.cpp:257-362
//   intrinsics fx = 525*W/640, cx = W/2 - 0.5 (floats)          :350-353
//   *.pcd scraped and sorted; the pose of NAME.pcd is NAME with the pose prefix and extension   :377-438
//   pose files: 12 numbers read as float (text or raw), optional inverse, translation * pose_units  :442-474
//   resolution = smallest power of two >= int(volume_size / cell_size)                              :477-494
//   frame pose = poses[0]^-1 * poses[i]                                                            :650
//   --organized: the cloud must already have the image size                                        :585-592
//   mesh: min weight, colour by RGB when --color, --flatten, --cleanup, PLY ascii/binary           :685-716
// Not reproduced: --visualize (PCLVisualizer) and the reference's PCL_INFO chatter.  --cloud-only works but
// its VoxelGrid thinning needs real PCL.
#include <cpu_tsdf/marching_cubes_tsdf_octree.h>
#include <cpu_tsdf/tsdf_volume_octree.h>

#include <pcl/console/print.h>
#include <pcl/console/time.h>
#include <pcl/io/pcd_io.h>
#include <pcl/io/ply_io.h>

#include <boost/filesystem.hpp>
#include <boost/program_options.hpp>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <fstream>
#include <iostream>
#include <limits>
#include <string>
#include <vector>

#include "mesh_post.h"

namespace bpo = boost::program_options;
namespace bfs = boost::filesystem;

namespace {

struct Settings {
  std::string in_dir, out_dir;
  bool verbose = false, flatten = false, cleanup = false, invert = false, organized = false, world_frame = false,
       zero_nans = false, save_ascii = false, save_tsdf = false, cloud_only = false, color = false;
  float cloud_units = 1.f, pose_units = 1.f, max_sensor_dist = 3.0f, min_sensor_dist = 0.f, min_weight = 0.f,
        trunc_pos = 0.03f, trunc_neg = 0.03f, volume_size = 12.f, cell_size = 0.006f, max_cell_size = 0.5f;
  int width = 640, height = 480, num_random_splits = 1;
  float fx = 525.f, fy = 525.f, cx = 319.5f, cy = 239.5f;
  bool limit_frames = false;
  size_t num_frames = 0;
};

template <typename T>
void take(const bpo::variables_map &vm, const char *name, T &dst) {
  if (vm.count(name)) dst = vm[name].as<T>();
}

// 0 = run, 1 = usage was printed
int parseCommandLine(int argc, char **argv, Settings &s) {
  bpo::options_description d("Allowed options");
  d.add_options()("help,h", "produce help message")("in", bpo::value<std::string>()->required(), "Input dir")(
      "out", bpo::value<std::string>()->required(), "Output dir")("save-tsdf", "Save the full TSDF in the output directory")(
      "volume-size", bpo::value<float>(), "Volume size")("cell-size", bpo::value<float>(), "Size of the smallest voxel (default 0.006)")(
      "max-cell-size", bpo::value<float>(), "Accepted for compatibility: the dense grid has no coarse cells")(
      "num-frames", bpo::value<size_t>(), "Only the first N clouds are used")("verbose", "Verbose")(
      "color", "Store color in addition to depth in the TSDF")("flatten", "Flatten mesh vertices")("cleanup", "Clean up mesh")(
      "invert", "Transforms are inverted (world -> camera)")("world", "Clouds are given in the world frame")(
      "organized", "Clouds are already organized")("width", bpo::value<int>(), "Image width")(
      "height", bpo::value<int>(), "Image height")("zero-nans", "Nans are represented as (0,0,0)")(
      "num-random-splits", bpo::value<int>(), "Accepted for compatibility (octree pre-split sampling)")(
      "fx", bpo::value<float>(), "Focal length x")("fy", bpo::value<float>(), "Focal length y")(
      "cx", bpo::value<float>(), "Center pixel x")("cy", bpo::value<float>(), "Center pixel y")(
      "save-ascii", "Save ply file as ASCII rather than binary")("cloud-units", bpo::value<float>(), "Units of the data, in meters")(
      "pose-units", bpo::value<float>(), "Units of the poses, in meters")(
      "max-sensor-dist", bpo::value<float>(), "Maximum distance data can be from the sensor")(
      "min-sensor-dist", bpo::value<float>(), "Minimum distance data can be from the sensor")(
      "trunc-dist-pos", bpo::value<float>(), "Positive truncation distance")(
      "trunc-dist-neg", bpo::value<float>(), "Negative truncation distance")("min-weight", bpo::value<float>(), "Minimum weight to render")(
      "cloud-only", "Save aggregate cloud rather than actually running TSDF");
  bpo::variables_map vm;
  bpo::store(bpo::parse_command_line(argc, argv, d, bpo::command_line_style::unix_style ^ bpo::command_line_style::allow_short), vm);
  bool bad = false;
  try {
    bpo::notify(vm);
  } catch (...) {
    bad = true;
  }
  if (vm.count("help") || bad) {
    std::cout << "Usage: " << bfs::basename(argv[0]) << " --in [in_dir] --out [out_dir] [OPTS]\n"
              << "Integrates a sequence of PCD clouds (poses: NAME.txt ascii or NAME.transform binary float, camera in the "
                 "world frame) into a TSDF on the GPU and writes a mesh.\n\n"
              << d << std::endl;
    return 1;
  }
  s.in_dir = vm["in"].as<std::string>();
  s.out_dir = vm["out"].as<std::string>();
  s.verbose = vm.count("verbose");
  s.flatten = vm.count("flatten");
  s.cleanup = vm.count("cleanup");
  s.invert = vm.count("invert");
  s.organized = vm.count("organized");
  s.world_frame = vm.count("world");
  s.zero_nans = vm.count("zero-nans");
  s.save_ascii = vm.count("save-ascii");
  s.save_tsdf = vm.count("save-tsdf");
  s.cloud_only = vm.count("cloud-only");
  s.color = vm.count("color");
  take(vm, "cloud-units", s.cloud_units);
  take(vm, "pose-units", s.pose_units);
  take(vm, "num-random-splits", s.num_random_splits);
  take(vm, "max-sensor-dist", s.max_sensor_dist);
  take(vm, "min-sensor-dist", s.min_sensor_dist);
  take(vm, "min-weight", s.min_weight);
  take(vm, "trunc-dist-pos", s.trunc_pos);
  take(vm, "trunc-dist-neg", s.trunc_neg);
  take(vm, "width", s.width);
  take(vm, "height", s.height);
  // double expressions stored into floats, as the reference's globals are (:350-353)
  s.fx = 525. * s.width / 640.;
  s.fy = 525. * s.height / 480.;
  s.cx = static_cast<float>(s.width) / 2. - 0.5;
  s.cy = static_cast<float>(s.height) / 2. - 0.5;
  take(vm, "fx", s.fx);
  take(vm, "fy", s.fy);
  take(vm, "cx", s.cx);
  take(vm, "cy", s.cy);
  take(vm, "volume-size", s.volume_size);
  take(vm, "cell-size", s.cell_size);
  take(vm, "max-cell-size", s.max_cell_size);
  if (vm.count("num-frames")) {
    s.limit_frames = true;
    s.num_frames = vm["num-frames"].as<size_t>();
  }
  return 0;
}

// Longest common prefix of the first and last name that contains no digit (:209-230).
std::string sharedPrefix(const std::vector<std::string> &sorted) {
  if (sorted.empty()) return "";
  const std::string &a = sorted.front(), &b = sorted.back();
  size_t i = 0;
  while (i < a.size() && i < b.size() && a[i] == b[i] && !std::isdigit((unsigned char)a[i])) ++i;
  return a.substr(0, i);
}

struct Sequence {
  std::vector<std::string> clouds, pose_files;
  bool binary_poses = false;
};

bool findSequence(const std::string &dir, Sequence &seq) {
  std::vector<std::string> pose_candidates;
  std::string pose_ext;
  for (bfs::directory_iterator it(dir), end; it != end; ++it) {
    const std::string ext = bfs::extension(it->path()), name = it->path().string();
    if (ext == ".pcd" || ext == ".PCD") {
      seq.clouds.push_back(name);
    } else if (ext == ".transform" || ext == ".TRANSFORM" || ext == ".txt" || ext == ".TXT") {
      if (!pose_ext.empty() && ext != pose_ext) {
        PCL_ERROR("Files with extension %s and %s were found in this folder! Please choose a consistent extension.\n",
                  ext.c_str(), pose_ext.c_str());
        return false;
      }
      pose_ext = ext;
      seq.binary_poses = (ext == ".transform" || ext == ".TRANSFORM");
      pose_candidates.push_back(name);
    }
  }
  std::sort(seq.clouds.begin(), seq.clouds.end());
  std::sort(pose_candidates.begin(), pose_candidates.end());
  const std::string cloud_prefix = sharedPrefix(seq.clouds), pose_prefix = sharedPrefix(pose_candidates);
  for (const std::string &c : seq.clouds) {
    const std::string stem = bfs::basename(bfs::path(c.substr(cloud_prefix.size())));
    const std::string pose = pose_prefix + stem + pose_ext;
    if (!bfs::exists(pose)) {
      PCL_ERROR("Could not find matching transform file for %s\n", c.c_str());
      return false;
    }
    seq.pose_files.push_back(pose);
  }
  std::sort(seq.pose_files.begin(), seq.pose_files.end());
  return true;
}

// 3x4 row-major numbers, each read as a FLOAT and widened (:449-466).
Eigen::Affine3d readPose(const std::string &file, bool binary, bool invert, float pose_units) {
  std::ifstream f(file.c_str());
  Eigen::Matrix4d m;
  m(3, 0) = 0, m(3, 1) = 0, m(3, 2) = 0, m(3, 3) = 1;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      float v;
      if (binary)
        f.read(reinterpret_cast<char *>(&v), sizeof v);
      else
        f >> v;
      m(r, c) = static_cast<double>(v);
    }
  Eigen::Affine3d pose;
  pose = m;
  if (invert) pose = pose.inverse();
  pose.matrix().topRightCorner<3, 1>() *= pose_units;
  return pose;
}

int resolutionFor(float volume_size, float cell_size) {
  const int wanted = volume_size / cell_size;  // float quotient truncated, :478
  int n = 1;
  while (wanted > n) n *= 2;
  return n;
}

}  // namespace

int main(int argc, char **argv) {
  Settings s;
  if (parseCommandLine(argc, argv, s)) return 1;
  pcl::console::TicToc clock;
  clock.tic();
  Sequence seq;
  if (!findSequence(s.in_dir, seq)) return 1;
  std::vector<Eigen::Affine3d> poses;
  for (const std::string &pf : seq.pose_files) poses.push_back(readPose(pf, seq.binary_poses, s.invert, s.pose_units));

  const int res = resolutionFor(s.volume_size, s.cell_size);
  cpu_tsdf::TSDFVolumeOctree::Ptr tsdf;
  if (!s.cloud_only) {
    tsdf.reset(new cpu_tsdf::TSDFVolumeOctree);
    tsdf->setGridSize(s.volume_size, s.volume_size, s.volume_size);
    tsdf->setResolution(res, res, res);
    tsdf->setMaxVoxelSize(s.max_cell_size, s.max_cell_size, s.max_cell_size);
    tsdf->setImageSize(s.width, s.height);
    tsdf->setCameraIntrinsics(s.fx, s.fy, s.cx, s.cy);
    tsdf->setNumRandomSplts(s.num_random_splits);
    tsdf->setSensorDistanceBounds(s.min_sensor_dist, s.max_sensor_dist);
    tsdf->setIntegrateColor(s.color);
    tsdf->setDepthTruncationLimits(s.trunc_pos, s.trunc_neg);
    tsdf->reset();
    if (!tsdf->handle()) return 1;
  }
  size_t n_frames = seq.clouds.size();
  if (s.limit_frames) {
    if (s.num_frames <= n_frames)
      n_frames = s.num_frames;
    else
      PCL_WARN("Warning: Manually input --num-frames=%zu, but the sequence only has %zu clouds. Ignoring user specification.\n",
               s.num_frames, n_frames);
  }
  pcl::PointCloud<pcl::PointXYZRGBA> aggregate;
  for (size_t i = 0; i < n_frames; ++i) {
    if (poses.size() <= i) {
      PCL_WARN("Warning: no matching pose file found for cloud %s; defaulting to identity.\n", seq.clouds[i].c_str());
      poses.push_back(Eigen::Affine3d::Identity());
    }
    pcl::PointCloud<pcl::PointXYZRGBA> cloud;
    if (pcl::io::loadPCDFile(seq.clouds[i], cloud)) return 1;
    if (s.organized && (cloud.height != (unsigned)s.height || cloud.width != (unsigned)s.width)) {
      PCL_ERROR("Error: cloud %d has size %d x %d, but TSDF is initialized for %d x %d pointclouds\n", (int)i + 1,
                cloud.width, cloud.height, s.width, s.height);
      return 1;
    }
    const Eigen::Affine3d pose = poses[0].inverse() * poses[i];
    const Eigen::Affine3d to_camera = poses[i].inverse();
    if (s.cloud_only || s.organized) {
      // host path: an organised cloud needs no z-buffer; the aggregate dump never reaches the GPU
      if (s.cloud_units != 1)
        for (auto &pt : cloud.points) pt.x *= s.cloud_units, pt.y *= s.cloud_units, pt.z *= s.cloud_units;
      if (s.zero_nans)
        for (auto &pt : cloud.points)
          if (pt.x == 0 && pt.y == 0 && pt.z == 0) pt.x = pt.y = pt.z = std::numeric_limits<float>::quiet_NaN();
      if (s.world_frame) pcl::transformPointCloud(cloud, cloud, to_camera);
      if (!s.cloud_only) {
        tsdf->integrateCloud(cloud, pcl::PointCloud<pcl::Normal>(), pose);
        continue;
      }
      if (!s.organized) {
        PCL_ERROR("--cloud-only of unorganised clouds needs the z-buffered frame on the host: not supported here\n");
        return 1;
      }
      pcl::PointCloud<pcl::PointXYZRGBA> seen;
      for (const auto &pt : cloud.points)
        if (!std::isnan(pt.z)) seen.push_back(pt);
      pcl::transformPointCloud(seen, seen, pose);
      aggregate += seen;
      continue;
    }
    size_t filled = 0;
    if (!tsdf->integrateUnorganized(cloud, pose, s.cloud_units, s.zero_nans, s.world_frame ? &to_camera : nullptr,
                                    s.verbose ? &filled : nullptr))
      return 1;
    if (s.verbose) PCL_INFO("Frame %d / %d: %d of %d points kept by the reprojection\n", (int)i + 1, (int)n_frames, (int)filled,
                            (int)cloud.size());
  }
  bfs::create_directory(s.out_dir);
  if (s.cloud_only) {
    pcl::io::savePCDFileBinaryCompressed(s.out_dir + "/cloud.pcd", aggregate);
    return 0;
  }
  cpu_tsdf::MarchingCubesTSDFOctree mc;
  mc.setMinWeight(s.min_weight);
  mc.setInputTSDF(tsdf);
  if (s.color) mc.setColorByRGB(true);
  pcl::PolygonMesh mesh;
  mc.reconstruct(mesh);
  if (s.flatten) cpu_tsdf::mesh_post::flattenVertices(mesh);
  if (s.cleanup) cpu_tsdf::mesh_post::cleanupMesh(mesh);
  PCL_INFO("Entire pipeline took %f ms\n", clock.toc());
  if (s.save_ascii)
    pcl::io::savePLYFile(s.out_dir + "/mesh.ply", mesh);
  else
    pcl::io::savePLYFileBinary(s.out_dir + "/mesh.ply", mesh);
  if (s.save_tsdf) tsdf->save(s.out_dir + "/volume.tsdf");
  return 0;
}
