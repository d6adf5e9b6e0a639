// Reader / writer for the reference's .vol checkpoint format, from and to a dense [z][y][x] grid.
//
// Format (src/lib/tsdf_volume_octree.cpp:222-245 + src/lib/octree.cpp:289-304,360-367,645-656 +
// include/eigen_extensions/eigen_extensions.h:249-257):
//   "# TSDFVolumeOctree Meta Information\n", then with precision 16 one group per line:
//   res(3) | size(3) | max_dist_pos | max_dist_neg | max_weight | min_sensor | max_sensor | max_cell(3) |
//   fx fy cx cy | image w h | is_empty | weight_by_depth | weight_by_variance
//   "% 4 4\n" + 4 rows of the global transform (columns right-aligned to the widest coefficient)
//   "<NOCOLOR|RGB>\n" "#OCTREEBINARY\n" res as 3 x size_t, size as 3 x float, then nodes in pre-order:
//   [uint8 r,g,b (RGB only)] float d,w,ctr_x,ctr_y,ctr_z,size,M ; int32 nsample ; size_t nchild (0|8)
//   child index = 4*(x>cx) + 2*(y>cy) + (z>cz), child centre = ctr -/+ size/4, child size = size/2.
//
// Writing synthesises an octree from the flat grid: a subtree whose voxels all hold the same (d, w, rgb)
// collapses into one leaf (lossless for every reader that looks voxels up by position, which is all the
// reference does); M_ and nsample_ -- only used by the reference's unreachable variance weighting -- are
// written as 0.  Reading rasterises every leaf over the voxels it covers.  Needs a cubic power-of-two
// grid (the only kind the reference's octree represents faithfully).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <vector>

namespace cpu_tsdf {

struct VolHeader {
  int res[3];
  float size[3];
  float max_dist_pos, max_dist_neg, max_weight, min_sensor_dist, max_sensor_dist;
  float max_cell[3];
  double fx, fy, cx, cy;
  int image_width, image_height;
  bool is_empty, weight_by_depth, weight_by_variance;
  double global_transform[16];  // row-major
  bool color;
};

namespace volfmt {

inline int log2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

struct Grid {
  int n, L;
  const float *d, *w;
  const unsigned char *rgb;
  // uniform[l][node] for levels 0..L-1 (level l has (2^l)^3 nodes); a node is uniform when all voxels
  // below it are bit-identical
  std::vector<std::vector<unsigned char> > uniform;
  size_t vox(int x, int y, int z) const { return ((size_t)z * n + y) * n + x; }
  bool same(size_t a, size_t b) const {
    return std::memcmp(d + a, d + b, 4) == 0 && std::memcmp(w + a, w + b, 4) == 0 &&
           (!rgb || std::memcmp(rgb + 3 * a, rgb + 3 * b, 3) == 0);
  }
};

inline void build_pyramid(Grid &g) {
  g.uniform.assign(g.L, std::vector<unsigned char>());
  for (int l = g.L - 1; l >= 0; --l) {
    const int m = 1 << l;            // nodes per axis at level l
    const int span = g.n >> l;       // voxels per node edge
    const int half = span / 2;
    g.uniform[l].assign((size_t)m * m * m, 0);
    const std::vector<unsigned char> *below = (l + 1 < g.L) ? &g.uniform[l + 1] : nullptr;
#pragma omp parallel for collapse(2)
    for (int kz = 0; kz < m; ++kz)
      for (int ky = 0; ky < m; ++ky)
        for (int kx = 0; kx < m; ++kx) {
          bool u = true;
          const size_t ref = g.vox(kx * span, ky * span, kz * span);
          for (int c = 0; c < 8 && u; ++c) {
            const int cx = 2 * kx + ((c >> 2) & 1), cy = 2 * ky + ((c >> 1) & 1), cz = 2 * kz + (c & 1);
            if (below) u = (*below)[((size_t)cz * 2 * m + cy) * 2 * m + cx] != 0;
            if (u) u = g.same(ref, g.vox(cx * half, cy * half, cz * half));
          }
          g.uniform[l][((size_t)kz * m + ky) * m + kx] = u ? 1 : 0;
        }
  }
}

template <typename T>
inline void put(std::ostream &f, const T &v) {
  f.write(reinterpret_cast<const char *>(&v), sizeof(T));
}

inline void write_node(std::ostream &f, const Grid &g, int level, int kx, int ky, int kz, float cx, float cy,
                       float cz, float size) {
  const int span = g.n >> level;
  const bool leaf = level == g.L || g.uniform[level][((size_t)kz * (1 << level) + ky) * (1 << level) + kx];
  const size_t v0 = g.vox(kx * span, ky * span, kz * span);
  if (g.rgb) {
    const unsigned char zero[3] = {0, 0, 0};
    f.write(reinterpret_cast<const char *>(leaf ? g.rgb + 3 * v0 : zero), 3);
  }
  const float d = leaf ? g.d[v0] : -1.f, w = leaf ? g.w[v0] : 0.f, M = 0.f;
  const int32_t nsample = 0;
  const size_t nchild = leaf ? 0 : 8;
  put(f, d);
  put(f, w);
  put(f, cx);
  put(f, cy);
  put(f, cz);
  put(f, size);
  put(f, M);
  put(f, nsample);
  put(f, nchild);
  if (leaf) return;
  const float off = size / 4, ns = size / 2;
  for (int c = 0; c < 8; ++c) {
    const int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
    write_node(f, g, level + 1, 2 * kx + bx, 2 * ky + by, 2 * kz + bz, bx ? cx + off : cx - off,
               by ? cy + off : cy - off, bz ? cz + off : cz - off, ns);
  }
}

// Eigen's operator<< for a 4x4 double matrix at the stream's precision [Eigen-recall: default IOFormat]
inline void write_matrix4(std::ostream &s, const double *m) {
  std::streamsize width = 0;
  for (int i = 0; i < 16; ++i) {
    std::stringstream ss;
    ss.copyfmt(s);
    ss << m[i];
    width = std::max<std::streamsize>(width, (std::streamsize)ss.str().length());
  }
  for (int r = 0; r < 4; ++r) {
    if (r) s << "\n";
    for (int c = 0; c < 4; ++c) {
      if (c) s << " ";
      s.width(width);
      s << m[4 * r + c];
    }
  }
}

template <typename T>
inline bool get(std::istream &f, T &v) {
  f.read(reinterpret_cast<char *>(&v), sizeof(T));
  return (bool)f;
}

struct ReadCtx {
  int n;
  float vs, half;
  bool color;
  float *d, *w;
  unsigned char *rgb;
  std::string err;
};

inline bool read_node(std::istream &f, ReadCtx &c, int depth) {
  unsigned char col[3] = {0, 0, 0};
  if (c.color) f.read(reinterpret_cast<char *>(col), 3);
  float d, w, cx, cy, cz, size, M;
  int32_t nsample;
  size_t nchild;
  if (!(get(f, d) && get(f, w) && get(f, cx) && get(f, cy) && get(f, cz) && get(f, size) && get(f, M) &&
        get(f, nsample) && get(f, nchild))) {
    c.err = "truncated octree";
    return false;
  }
  if (nchild == 0) {  // leaf: fill the voxels it covers
    const int span = std::max(1, (int)std::lround(size / c.vs));
    const int x0 = (int)std::lround((cx - size / 2 + c.half) / c.vs), y0 = (int)std::lround((cy - size / 2 + c.half) / c.vs),
              z0 = (int)std::lround((cz - size / 2 + c.half) / c.vs);
    if (x0 < 0 || y0 < 0 || z0 < 0 || x0 + span > c.n || y0 + span > c.n || z0 + span > c.n) {
      c.err = "leaf outside the grid";
      return false;
    }
    for (int z = z0; z < z0 + span; ++z)
      for (int y = y0; y < y0 + span; ++y)
        for (int x = x0; x < x0 + span; ++x) {
          const size_t v = ((size_t)z * c.n + y) * c.n + x;
          c.d[v] = d;
          c.w[v] = w;
          if (c.color) std::memcpy(c.rgb + 3 * v, col, 3);
        }
    return true;
  }
  if (nchild != 8 || depth > 24) {
    c.err = "malformed octree node";
    return false;
  }
  for (int k = 0; k < 8; ++k)
    if (!read_node(f, c, depth + 1)) return false;
  return true;
}

}  // namespace volfmt

inline bool vol_write(const std::string &filename, const VolHeader &h, const float *d, const float *w,
                      const unsigned char *rgb, std::string *err) {
  const int L = volfmt::log2_exact(h.res[0]);
  if (L < 0 || h.res[1] != h.res[0] || h.res[2] != h.res[0]) {
    if (err) *err = "the .vol octree format needs a cubic power-of-two resolution";
    return false;
  }
  std::ofstream f(filename.c_str(), std::ios::binary);
  if (!f) {
    if (err) *err = "cannot open " + filename;
    return false;
  }
  f << "# TSDFVolumeOctree Meta Information" << std::endl;
  f.precision(16);
  f << h.res[0] << " " << h.res[1] << " " << h.res[2] << std::endl;
  f << h.size[0] << " " << h.size[1] << " " << h.size[2] << std::endl;
  f << h.max_dist_pos << std::endl;
  f << h.max_dist_neg << std::endl;
  f << h.max_weight << std::endl;
  f << h.min_sensor_dist << std::endl;
  f << h.max_sensor_dist << std::endl;
  f << h.max_cell[0] << " " << h.max_cell[1] << " " << h.max_cell[2] << std::endl;
  f << h.fx << " " << h.fy << " " << h.cx << " " << h.cy << std::endl;
  f << h.image_width << " " << h.image_height << std::endl;
  f << h.is_empty << std::endl;
  f << h.weight_by_depth << std::endl;
  f << h.weight_by_variance << std::endl;
  {  // eigen_extensions::serializeASCII
    const std::streamsize old = f.precision();
    f.precision(16);
    f << "% " << 4 << " " << 4 << std::endl;
    volfmt::write_matrix4(f, h.global_transform);
    f << std::endl;
    f.precision(old);
  }
  f << (h.color ? "RGB" : "NOCOLOR") << std::endl;
  f << "#OCTREEBINARY" << std::endl;
  const size_t r3[3] = {(size_t)h.res[0], (size_t)h.res[1], (size_t)h.res[2]};
  for (int k = 0; k < 3; ++k) volfmt::put(f, r3[k]);
  for (int k = 0; k < 3; ++k) volfmt::put(f, h.size[k]);
  volfmt::Grid g;
  g.n = h.res[0];
  g.L = L;
  g.d = d;
  g.w = w;
  g.rgb = h.color ? rgb : nullptr;
  volfmt::build_pyramid(g);
  volfmt::write_node(f, g, 0, 0, 0, 0, 0.f, 0.f, 0.f, h.size[0]);
  f.close();
  if (!f) {
    if (err) *err = "write error on " + filename;
    return false;
  }
  return true;
}

inline bool vol_read(const std::string &filename, VolHeader &h, std::vector<float> &d, std::vector<float> &w,
                     std::vector<unsigned char> &rgb, std::string *err) {
  std::ifstream f(filename.c_str(), std::ios::binary);
  if (!f) {
    if (err) *err = "cannot open " + filename;
    return false;
  }
  char line[1024];
  f.getline(line, 1024);
  f >> h.res[0] >> h.res[1] >> h.res[2];
  f >> h.size[0] >> h.size[1] >> h.size[2];
  f >> h.max_dist_pos >> h.max_dist_neg >> h.max_weight >> h.min_sensor_dist >> h.max_sensor_dist;
  f >> h.max_cell[0] >> h.max_cell[1] >> h.max_cell[2];
  f >> h.fx >> h.fy >> h.cx >> h.cy;
  f >> h.image_width >> h.image_height;
  f >> h.is_empty >> h.weight_by_depth >> h.weight_by_variance;
  std::string s;
  while (s.empty() && std::getline(f, s)) {
  }
  if (s.empty() || s[0] != '%') {
    if (err) *err = "missing transform header in " + filename;
    return false;
  }
  for (int i = 0; i < 16; ++i) {
    std::string tok;
    f >> tok;
    h.global_transform[i] = tok[0] == 'n' ? std::nan("") : std::atof(tok.c_str());
  }
  std::string type;
  f >> type;
  if (type != "RGB" && type != "NOCOLOR") {
    if (err) *err = "unsupported voxel type '" + type + "' (only NOCOLOR and RGB)";
    return false;
  }
  h.color = type == "RGB";
  do {
    f.getline(line, 1024);
  } while (f && !(line[0] == '#' && line[1] == 'O'));
  size_t r3[3];
  float s3[3];
  for (int k = 0; k < 3; ++k) volfmt::get(f, r3[k]);
  for (int k = 0; k < 3; ++k) volfmt::get(f, s3[k]);
  if (!f || (int)r3[0] != h.res[0] || volfmt::log2_exact(h.res[0]) < 0 || h.res[1] != h.res[0] || h.res[2] != h.res[0]) {
    if (err) *err = "bad octree header (cubic power-of-two grids only)";
    return false;
  }
  const size_t n = (size_t)h.res[0] * h.res[1] * h.res[2];
  d.assign(n, -1.f);
  w.assign(n, 0.f);
  rgb.assign(h.color ? 3 * n : 0, 0);
  volfmt::ReadCtx c;
  c.n = h.res[0];
  c.vs = s3[0] / (float)h.res[0];
  c.half = s3[0] / 2;
  c.color = h.color;
  c.d = d.data();
  c.w = w.data();
  c.rgb = h.color ? rgb.data() : nullptr;
  if (!volfmt::read_node(f, c, 0)) {
    if (err) *err = c.err;
    return false;
  }
  return true;
}

}  // namespace cpu_tsdf
