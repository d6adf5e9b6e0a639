// Test harness (CPU only) for cpu_tsdf_amd/csrc/host/vol_format.h: the streaming .vol writer / reader
// driven from host arrays instead of device memory.
//   vol_stream write <raw-in> <res> <size> <color 0|1> <chunk> <vol-out>
//   vol_stream read  <vol-in> <chunk> <raw-out>      (prints "res color blocks" on stdout)
// raw = d[res^3] float32 | w[res^3] float32 | rgb[3 res^3] uint8 (rgb only with colour), z-major.
#include <chrono>
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
  if (mode == "bench" && argc == 5) {
    // vol_stream bench <res> <chunk> <vol-out>: a TSDF-like grid (unseen octant, observed free space whose weights and
    // colours differ voxel by voxel, a noisy band) written and read back, seconds on stdout -- a CPU-only measure of the
    // format code itself (blocks are served by memcpy, where the product downloads them from the GPU)
    const int n = std::atoi(argv[2]), chunk = std::atoi(argv[3]);
    const size_t nv = (size_t)n * n * n;
    std::vector<float> d(nv, -1.f), w(nv, 0.f);
    std::vector<unsigned char> rgb(3 * nv, 0);
    uint32_t lcg = 12345u;
    for (int z = 0; z < n; ++z)
      for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
          if (x < n / 2 && y < n / 2) continue;  // never observed: a quarter of the grid collapses
          const size_t i = ((size_t)z * n + y) * n + x;
          lcg = lcg * 1664525u + 1013904223u;
          const bool band = (z > n / 2 - 6 && z < n / 2 + 6);
          d[i] = band ? (float)((int)(lcg >> 8 & 0xffff) - 32768) / 32768.f : 1.f;
          w[i] = (float)(1 + (lcg >> 28));
          rgb[3 * i] = (unsigned char)(lcg >> 3), rgb[3 * i + 1] = (unsigned char)(lcg >> 11), rgb[3 * i + 2] = (unsigned char)(lcg >> 19);
        }
    VolHeader h;
    for (int k = 0; k < 3; ++k) h.res[k] = n, h.size[k] = 1.f, h.max_cell[k] = 1.f / n;
    h.max_dist_pos = h.max_dist_neg = 0.03f, h.max_weight = 100.f, h.min_sensor_dist = 0.f, h.max_sensor_dist = 3.f;
    h.fx = h.fy = 525., h.cx = 319.5, h.cy = 239.5, h.image_width = 640, h.image_height = 480;
    h.is_empty = false, h.weight_by_depth = h.weight_by_variance = false, h.color = true;
    for (int i = 0; i < 16; ++i) h.global_transform[i] = (i % 5 == 0) ? 1. : 0.;
    auto copy = [&](bool out, int x0, int y0, int z0, int c, float *bd, float *bw, unsigned char *brgb) {
      for (int z = 0; z < c; ++z)
        for (int y = 0; y < c; ++y) {
          const size_t s = ((size_t)(z0 + z) * n + y0 + y) * n + x0, t = ((size_t)z * c + y) * c;
          if (out) {
            std::memcpy(bd + t, &d[s], 4 * (size_t)c), std::memcpy(bw + t, &w[s], 4 * (size_t)c);
            std::memcpy(brgb + 3 * t, &rgb[3 * s], 3 * (size_t)c);
          } else {
            std::memcpy(&d[s], bd + t, 4 * (size_t)c), std::memcpy(&w[s], bw + t, 4 * (size_t)c);
            std::memcpy(&rgb[3 * s], brgb + 3 * t, 3 * (size_t)c);
          }
        }
      return true;
    };
    auto digest = [&]() {  // FNV-1a over the three arrays: the read must put back exactly what was written
      uint64_t hsh = 1469598103934665603ull;
      auto eat = [&](const void *p, size_t bytes) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < bytes; i += 8) {  // (every 8th byte: a digest, not a proof -- tests/test_vol_stream.py compares everything)
          hsh ^= b[i];
          hsh *= 1099511628211ull;
        }
      };
      eat(d.data(), 4 * nv), eat(w.data(), 4 * nv), eat(rgb.data(), 3 * nv);
      return hsh;
    };
    const uint64_t before = digest();
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    if (!vol_write_stream(argv[4], h, chunk, [&](int x0, int y0, int z0, int c, float *bd, float *bw, unsigned char *brgb) {
          return copy(true, x0, y0, z0, c, bd, bw, brgb); }, &err)) {
      std::cerr << err << std::endl;
      return 1;
    }
    auto t1 = clk::now();
    const double write_s = std::chrono::duration<double>(t1 - t0).count();
    if (std::string(argv[4]) == "/dev/null") {  // the serialisation alone, no file system behind it
      std::printf("write %.3f s (to /dev/null)\n", std::chrono::duration<double>(t1 - t0).count());
      return 0;
    }
    std::fill(d.begin(), d.end(), 7.f), std::fill(w.begin(), w.end(), 7.f), std::fill(rgb.begin(), rgb.end(), (unsigned char)7);
    t1 = clk::now();
    VolHeader h2;
    if (!vol_read_stream(argv[4], h2, chunk, [&](const VolHeader &) { return true; },
                         [&](int x0, int y0, int z0, int c, float *bd, float *bw, unsigned char *brgb) {
                           return copy(false, x0, y0, z0, c, bd, bw, brgb); }, &err)) {
      std::cerr << err << std::endl;
      return 1;
    }
    auto t2 = clk::now();
    if (digest() != before) {
      std::cerr << "the read did not restore the grid" << std::endl;
      return 6;
    }
    FILE *f = std::fopen(argv[4], "rb");
    std::fseek(f, 0, SEEK_END);
    const double gb = (double)std::ftell(f) / 1e9;
    std::fclose(f);
    std::printf("write %.3f s  read %.3f s  file %.3f GB  (%.2f / %.2f GB/s of file)  restored\n", write_s,
                std::chrono::duration<double>(t2 - t1).count(), gb, gb / write_s, gb / std::chrono::duration<double>(t2 - t1).count());
    return 0;
  }
  return 2;
}
