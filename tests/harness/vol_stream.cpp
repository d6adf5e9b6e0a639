// Test harness (CPU only) for cpu_tsdf_amd/csrc/host/vol_format.h: the streaming .vol writer / reader
// driven from host arrays instead of device memory.
//   vol_stream write <raw-in> <res> <size> <color 0|1> <chunk> <vol-out>
//   vol_stream read  <vol-in> <chunk> <raw-out>      (prints "res color blocks" on stdout)
// raw = d[res^3] float32 | w[res^3] float32 | rgb[3 res^3] uint8 (rgb only with colour), z-major.
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "vol_format.h"

using namespace cpu_tsdf;

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const std::string mode = argv[1];
  std::string err;
  if (mode == "write" && argc == 8) {
    const int n = std::atoi(argv[3]);
    const bool color = std::atoi(argv[5]) != 0;
    const size_t nv = (size_t)n * n * n;
    std::vector<float> d(nv), w(nv);
    std::vector<unsigned char> rgb(color ? 3 * nv : 0);
    FILE *f = std::fopen(argv[2], "rb");
    if (!f || std::fread(d.data(), 4, nv, f) != nv || std::fread(w.data(), 4, nv, f) != nv ||
        (color && std::fread(rgb.data(), 1, 3 * nv, f) != 3 * nv))
      return 3;
    std::fclose(f);
    VolHeader h;
    for (int k = 0; k < 3; ++k) {
      h.res[k] = n;
      h.size[k] = (float)std::atof(argv[4]);
      h.max_cell[k] = h.size[k] / n;
    }
    h.max_dist_pos = h.max_dist_neg = 0.03f;
    h.max_weight = 100.f;
    h.min_sensor_dist = 0.f;
    h.max_sensor_dist = 3.f;
    h.fx = h.fy = 525.;
    h.cx = 319.5;
    h.cy = 239.5;
    h.image_width = 640;
    h.image_height = 480;
    h.is_empty = false;
    h.weight_by_depth = h.weight_by_variance = false;
    for (int i = 0; i < 16; ++i) h.global_transform[i] = (i % 5 == 0) ? 1. : 0.;
    h.color = color;
    long fetched = 0;
    const bool ok = vol_write_stream(
        argv[7], h, std::atoi(argv[6]),
        [&](int x0, int y0, int z0, int c, float *bd, float *bw, unsigned char *brgb) {
          ++fetched;
          for (int z = 0; z < c; ++z)
            for (int y = 0; y < c; ++y)
              for (int x = 0; x < c; ++x) {
                const size_t s = ((size_t)(z0 + z) * n + y0 + y) * n + x0 + x, t = ((size_t)z * c + y) * c + x;
                bd[t] = d[s];
                bw[t] = w[s];
                if (brgb) std::memcpy(brgb + 3 * t, &rgb[3 * s], 3);
              }
          return true;
        },
        &err);
    if (!ok) {
      std::cerr << err << std::endl;
      return 1;
    }
    std::cout << fetched << std::endl;
    return 0;
  }
  if (mode == "read" && argc == 5) {
    VolHeader h;
    std::vector<float> d, w;
    std::vector<unsigned char> rgb, seen;
    int n = 0;
    long blocks = 0;
    bool twice = false;
    const bool ok = vol_read_stream(
        argv[2], h, std::atoi(argv[3]),
        [&](const VolHeader &hd) {
          n = hd.res[0];
          const size_t nv = (size_t)n * n * n;
          d.assign(nv, 12345.f);
          w.assign(nv, 12345.f);
          rgb.assign(hd.color ? 3 * nv : 0, 77);
          seen.assign(nv, 0);
          return true;
        },
        [&](int x0, int y0, int z0, int c, float *bd, float *bw, unsigned char *brgb) {
          ++blocks;
          for (int z = 0; z < c; ++z)
            for (int y = 0; y < c; ++y)
              for (int x = 0; x < c; ++x) {
                const size_t s = ((size_t)(z0 + z) * n + y0 + y) * n + x0 + x, t = ((size_t)z * c + y) * c + x;
                d[s] = bd[t];
                w[s] = bw[t];
                if (brgb) std::memcpy(&rgb[3 * s], brgb + 3 * t, 3);
                twice |= seen[s] != 0;
                seen[s] = 1;
              }
          return true;
        },
        &err);
    if (!ok) {
      std::cerr << err << std::endl;
      return 1;
    }
    for (unsigned char s : seen)
      if (!s) {
        std::cerr << "a voxel was never stored" << std::endl;
        return 4;
      }
    if (twice) {
      std::cerr << "a voxel was stored twice" << std::endl;
      return 5;
    }
    FILE *f = std::fopen(argv[4], "wb");
    std::fwrite(d.data(), 4, d.size(), f);
    std::fwrite(w.data(), 4, w.size(), f);
    if (h.color) std::fwrite(rgb.data(), 1, rgb.size(), f);
    std::fclose(f);
    std::cout << n << " " << (h.color ? 1 : 0) << " " << blocks << std::endl;
    return 0;
  }
  return 2;
}
