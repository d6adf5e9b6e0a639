// `dropin_rate` -- what a C++ user of the drop-in waits for: N calls of
// cpu_tsdf::TSDFVolumeOctree::integrateCloud (pcl::PointCloud<pcl::PointXYZRGBA>, the reference's signature,
// include/cpu_tsdf/tsdf_volume_octree.h:241-245) on organised clouds held in HOST memory, timed end to end: the AoS strip into
// the pinned slot, the upload, the kernel, and one final call that waits for the device.  No reference counterpart (the
// reference has no benchmark program); bench.py runs it for `host_path.cpp_dropin_*`, tools/cpp_path_timing.py for
// profiles/.
//
// usage: dropin_rate RES WIDTH HEIGHT COLOR PAIRING NFRAMES FRAMES.bin
//   FRAMES.bin: K records of { double pose[16] (camera-to-volume, row major); float depth[H*W] (camera z, NaN = no return);
//               uint8 bgra[H*W*4] }, written by the caller (bench.py: its own Scene-A frames); frame i uses record i % K.
//   Grid: RES^3 voxels of 2^-8 m; intrinsics as the reference's program derives them (src/prog/integrate.cpp:350-353).
// Prints one JSON line.
#include <cpu_tsdf/tsdf_volume_octree.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  if (argc < 8) {
    fprintf(stderr, "usage: %s RES WIDTH HEIGHT COLOR PAIRING NFRAMES FRAMES.bin\n", argv[0]);
    return 2;
  }
  const int res = atoi(argv[1]), W = atoi(argv[2]), H = atoi(argv[3]);
  const bool color = atoi(argv[4]) != 0, pairing = atoi(argv[5]) != 0;
  const int n_frames = atoi(argv[6]);
  FILE *f = fopen(argv[7], "rb");
  if (!f) {
    perror(argv[7]);
    return 2;
  }
  const size_t npx = (size_t)W * H, rec = 16 * sizeof(double) + npx * 4 + npx * 4;
  fseek(f, 0, SEEK_END);
  const size_t K = (size_t)ftell(f) / rec;
  fseek(f, 0, SEEK_SET);
  if (!K) {
    fprintf(stderr, "no frame record in %s\n", argv[7]);
    return 2;
  }
  const double size = res * std::ldexp(1.0, -8);
  const double fx = 525.0 * W / 640.0, fy = fx, cx = W / 2.0 - 0.5, cy = H / 2.0 - 0.5;  // integrate.cpp:350-353
  std::vector<pcl::PointCloud<pcl::PointXYZRGBA>> clouds(K);
  std::vector<Eigen::Affine3d> poses(K);
  std::vector<float> depth(npx);
  std::vector<unsigned char> bgra(npx * 4);
  for (size_t k = 0; k < K; ++k) {
    double m[16];
    if (fread(m, sizeof m, 1, f) != 1 || fread(depth.data(), 4, npx, f) != npx || fread(bgra.data(), 4, npx, f) != npx) return 2;
    Eigen::Matrix4d M;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) M(r, c) = m[4 * r + c];
    poses[k].matrix() = M;
    pcl::PointCloud<pcl::PointXYZRGBA> &cl = clouds[k];
    cl.width = W, cl.height = H, cl.is_dense = false;
    cl.points.resize(npx);
    for (int v = 0; v < H; ++v)
      for (int u = 0; u < W; ++u) {
        pcl::PointXYZRGBA &p = cl.points[(size_t)v * W + u];
        const float z = depth[(size_t)v * W + u];
        p.z = z;
        p.x = (float)((u - cx) * z / fx);
        p.y = (float)((v - cy) * z / fy);
        p.b = bgra[4 * ((size_t)v * W + u) + 0];
        p.g = bgra[4 * ((size_t)v * W + u) + 1];
        p.r = bgra[4 * ((size_t)v * W + u) + 2];
        p.a = 255;
      }
  }
  fclose(f);
  cpu_tsdf::TSDFVolumeOctree::Ptr tsdf(new cpu_tsdf::TSDFVolumeOctree);
  tsdf->setGridSize((float)size, (float)size, (float)size);
  tsdf->setResolution(res, res, res);
  tsdf->setImageSize(W, H);
  tsdf->setCameraIntrinsics(fx, fy, cx, cy);
  tsdf->setSensorDistanceBounds(0.f, (float)(3.0 * size));
  tsdf->setIntegrateColor(color);
  tsdf->setDepthTruncationLimits(0.03f, 0.03f);
  tsdf->reset();
  if (!tsdf->handle()) return 1;
  tsdf->setFramePairing(pairing);
  const pcl::PointCloud<pcl::Normal> no_normals;
  const pcl::PointXYZ probe(0.f, 0.f, 0.f);
  float val;
  for (size_t k = 0; k < std::min<size_t>(K, 4); ++k) tsdf->integrateCloud(clouds[k], no_normals, poses[k]);  // warm-up: pinned ring, first launches
  tsdf->getFxn(probe, val);  // waits for the device
  std::vector<double> in_call(n_frames);
  const double t0 = now_s();
  for (int i = 0; i < n_frames; ++i) {
    const double a = now_s();
    if (!tsdf->integrateCloud(clouds[i % K], no_normals, poses[i % K])) return 1;
    in_call[i] = now_s() - a;
  }
  tsdf->getFxn(probe, val);  // ordered after every queued frame: drains the pipeline
  const double wall = now_s() - t0;
  std::vector<double> s(in_call);
  std::sort(s.begin(), s.end());
  double mean = 0;
  for (double v : in_call) mean += v;
  printf("{\"res\": %d, \"image\": [%d, %d], \"color\": %s, \"frame_pairing\": %s, \"frames\": %d, \"distinct_clouds\": %zu, "
         "\"ms_in_integrateCloud_call_median\": %.4f, \"ms_in_integrateCloud_call_mean\": %.4f, \"sustained_ms_per_frame\": %.4f, "
         "\"sustained_frames_per_s\": %.2f}\n",
         res, W, H, color ? "true" : "false", pairing ? "true" : "false", n_frames, K, s[s.size() / 2] * 1e3, mean / n_frames * 1e3,
         wall / n_frames * 1e3, n_frames / wall);
  return 0;
}
