// Stand-ins for the PCL I/O, search and segmentation pieces the `integrate` / `tsdf2mesh` programs call
// (no PCL on this machine): PCD reader (ascii / binary / binary_compressed), PCD binary writer, PLY
// writer for PolygonMesh, radius search, Euclidean clustering, a pass-through VoxelGrid.
// [PCL-recall] throughout: file grammars and algorithm contracts as PCL 1.10-1.13 document them; where
// PCL leaves an order unspecified (radius-search ties) this file picks index order.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "mini_pcl.h"

namespace pcl {

struct PointIndices {
  PCLHeader header;
  std::vector<int> indices;
};

namespace io {
namespace detail {

// LibLZF decompression (the format PCL's binary_compressed PCD body uses).
inline bool lzf_decompress(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {  // literal run
      ctrl++;
      if (op + ctrl > out_len || ip + ctrl > in_len) return false;
      std::memcpy(out + op, in + ip, ctrl);
      op += ctrl;
      ip += ctrl;
    } else {  // back reference
      size_t len = ctrl >> 5;
      if (ip >= in_len) return false;
      size_t ref_off = ((ctrl & 0x1f) << 8) + 1;
      if (len == 7) {
        len += in[ip++];
        if (ip >= in_len) return false;
      }
      ref_off += in[ip++];
      len += 2;
      if (ref_off > op || op + len > out_len) return false;
      size_t ref = op - ref_off;
      for (size_t k = 0; k < len; ++k) out[op++] = out[ref++];
    }
  }
  return op == out_len;
}

struct PcdField {
  std::string name;
  int size = 4, count = 1;
  char type = 'F';
  size_t offset = 0;
};

inline double read_scalar(const unsigned char *p, const PcdField &f) {
  switch (f.type) {
    case 'F':
      if (f.size == 4) { float v; std::memcpy(&v, p, 4); return v; }
      { double v; std::memcpy(&v, p, 8); return v; }
    case 'U':
      if (f.size == 1) return *p;
      if (f.size == 2) { std::uint16_t v; std::memcpy(&v, p, 2); return v; }
      if (f.size == 4) { std::uint32_t v; std::memcpy(&v, p, 4); return v; }
      { std::uint64_t v; std::memcpy(&v, p, 8); return (double)v; }
    default:
      if (f.size == 1) return *(const signed char *)p;
      if (f.size == 2) { std::int16_t v; std::memcpy(&v, p, 2); return v; }
      if (f.size == 4) { std::int32_t v; std::memcpy(&v, p, 4); return v; }
      { std::int64_t v; std::memcpy(&v, p, 8); return (double)v; }
  }
}

inline std::uint32_t parse_rgb_token(const std::string &tok) {
  // PCL writes the packed colour of a FLOAT32 "rgb" field as an unsigned integer in ASCII files; older
  // files carry the float's decimal text instead
  if (tok.find_first_of(".eEnN") == std::string::npos) return (std::uint32_t)std::strtoul(tok.c_str(), nullptr, 10);
  const float f = std::strtof(tok.c_str(), nullptr);
  std::uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}

}  // namespace detail

// pcl::io::loadPCDFile<PointXYZRGBA>: fields x, y, z and rgb / rgba are mapped by name, everything else is
// skipped; is_dense is cleared if a non-finite coordinate was read.  0 on success, -1 on failure.
inline int loadPCDFile(const std::string &file, PointCloud<PointXYZRGBA> &cloud) {
  std::ifstream f(file.c_str(), std::ios::binary);
  if (!f) {
    std::fprintf(stderr, "[pcl::PCDReader::read] could not open %s\n", file.c_str());
    return -1;
  }
  std::vector<detail::PcdField> fields;
  size_t width = 0, height = 1, points = 0;
  bool have_points = false;
  std::string line, data_kind;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream ls(line);
    std::string key;
    if (!(ls >> key) || key[0] == '#') continue;
    if (key == "FIELDS" || key == "COLUMNS") {
      std::string n;
      while (ls >> n) {
        detail::PcdField pf;
        pf.name = n;
        fields.push_back(pf);
      }
    } else if (key == "SIZE") {
      for (auto &pf : fields) ls >> pf.size;
    } else if (key == "TYPE") {
      for (auto &pf : fields) ls >> pf.type;
    } else if (key == "COUNT") {
      for (auto &pf : fields) ls >> pf.count;
    } else if (key == "WIDTH") {
      ls >> width;
    } else if (key == "HEIGHT") {
      ls >> height;
    } else if (key == "POINTS") {
      ls >> points;
      have_points = true;
    } else if (key == "DATA") {
      ls >> data_kind;
      break;
    }
  }
  if (data_kind.empty() || fields.empty()) {
    std::fprintf(stderr, "[pcl::PCDReader::read] %s: no DATA / FIELDS line\n", file.c_str());
    return -1;
  }
  if (!have_points) points = width * height;
  size_t point_size = 0;
  for (auto &pf : fields) {
    pf.offset = point_size;
    point_size += (size_t)pf.size * pf.count;
  }
  cloud.points.assign(points, PointXYZRGBA());
  cloud.width = (std::uint32_t)(width ? width : points);
  cloud.height = (std::uint32_t)(width ? height : 1);
  cloud.is_dense = true;
  auto assign = [&](PointXYZRGBA &pt, const detail::PcdField &pf, double v, std::uint32_t raw) {
    if (pf.name == "x") pt.x = (float)v;
    else if (pf.name == "y") pt.y = (float)v;
    else if (pf.name == "z") pt.z = (float)v;
    else if (pf.name == "rgb" || pf.name == "rgba") pt.rgba = raw;
  };
  if (data_kind == "ascii") {
    for (size_t i = 0; i < points; ++i) {
      if (!std::getline(f, line)) return -1;
      std::istringstream ls(line);
      for (const auto &pf : fields)
        for (int c = 0; c < pf.count; ++c) {
          std::string tok;
          if (!(ls >> tok)) return -1;
          if (c) continue;
          if (pf.name == "rgb" || pf.name == "rgba")
            assign(cloud.points[i], pf, 0, pf.type == 'F' ? detail::parse_rgb_token(tok) : (std::uint32_t)std::strtoul(tok.c_str(), nullptr, 10));
          else
            assign(cloud.points[i], pf, std::strtod(tok.c_str(), nullptr), 0);
        }
    }
  } else {
    std::vector<unsigned char> body(points * point_size);
    if (data_kind == "binary") {
      f.read((char *)body.data(), (std::streamsize)body.size());
      if ((size_t)f.gcount() != body.size()) return -1;
    } else if (data_kind == "binary_compressed") {
      std::uint32_t csize = 0, usize = 0;
      f.read((char *)&csize, 4);
      f.read((char *)&usize, 4);
      std::vector<unsigned char> comp(csize), soa(usize);
      f.read((char *)comp.data(), csize);
      if ((size_t)f.gcount() != csize || usize != body.size() ||
          !detail::lzf_decompress(comp.data(), csize, soa.data(), usize))
        return -1;
      size_t off = 0;  // the decompressed body is field-major: all x, then all y, ...
      for (const auto &pf : fields) {
        const size_t fs = (size_t)pf.size * pf.count;
        for (size_t i = 0; i < points; ++i) std::memcpy(&body[i * point_size + pf.offset], &soa[off + i * fs], fs);
        off += fs * points;
      }
    } else {
      return -1;
    }
    for (size_t i = 0; i < points; ++i)
      for (const auto &pf : fields) {
        const unsigned char *p = &body[i * point_size + pf.offset];
        std::uint32_t raw = 0;
        if (pf.size == 4) std::memcpy(&raw, p, 4);
        assign(cloud.points[i], pf, detail::read_scalar(p, pf), raw);
      }
  }
  for (const auto &pt : cloud.points)
    if (!std::isfinite(pt.x) || !std::isfinite(pt.y) || !std::isfinite(pt.z)) {
      cloud.is_dense = false;
      break;
    }
  return 0;
}

// Writes a plain binary PCD (this stand-in does not compress); only the --cloud-only mode calls it.
inline int savePCDFileBinaryCompressed(const std::string &file, const PointCloud<PointXYZRGBA> &cloud) {
  std::ofstream f(file.c_str(), std::ios::binary);
  if (!f) return -1;
  const size_t n = cloud.points.size();
  f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgba\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\n"
    << "WIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA binary\n";
  for (const auto &pt : cloud.points) {
    f.write((const char *)&pt.x, 12);
    f.write((const char *)&pt.rgba, 4);
  }
  return f ? 0 : -1;
}

namespace detail {
struct MeshLayout {
  int x = -1, y = -1, z = -1, rgb = -1;
  bool alpha = false;
};
inline MeshLayout mesh_layout(const PolygonMesh &m) {
  MeshLayout l;
  for (const auto &fd : m.cloud.fields) {
    if (fd.name == "x") l.x = (int)fd.offset;
    if (fd.name == "y") l.y = (int)fd.offset;
    if (fd.name == "z") l.z = (int)fd.offset;
    if (fd.name == "rgb" || fd.name == "rgba") {
      l.rgb = (int)fd.offset;
      l.alpha = fd.name == "rgba";
    }
  }
  return l;
}
inline int save_ply(const std::string &file, const PolygonMesh &mesh, bool binary, unsigned precision) {
  std::ofstream f(file.c_str(), std::ios::binary);
  if (!f) return -1;
  const MeshLayout l = mesh_layout(mesh);
  const size_t nv = (size_t)mesh.cloud.width * mesh.cloud.height, step = mesh.cloud.point_step;
  f << "ply\nformat " << (binary ? "binary_little_endian" : "ascii") << " 1.0\ncomment PCL generated\n"
    << "element vertex " << nv << "\nproperty float x\nproperty float y\nproperty float z\n";
  if (l.rgb >= 0) {
    f << "property uchar red\nproperty uchar green\nproperty uchar blue\n";
    if (l.alpha) f << "property uchar alpha\n";
  }
  f << "element face " << mesh.polygons.size() << "\nproperty list uchar int vertex_indices\nend_header\n";
  if (!binary) f << std::setprecision((int)precision);
  for (size_t i = 0; i < nv; ++i) {
    const unsigned char *p = mesh.cloud.data.data() + i * step;
    float xyz[3] = {0, 0, 0};
    if (l.x >= 0) std::memcpy(&xyz[0], p + l.x, 4);
    if (l.y >= 0) std::memcpy(&xyz[1], p + l.y, 4);
    if (l.z >= 0) std::memcpy(&xyz[2], p + l.z, 4);
    unsigned char c[4] = {0, 0, 0, 0};  // memory order b, g, r, a
    if (l.rgb >= 0) std::memcpy(c, p + l.rgb, 4);
    if (binary) {
      f.write((const char *)xyz, 12);
      if (l.rgb >= 0) {
        const unsigned char rgb[4] = {c[2], c[1], c[0], c[3]};
        f.write((const char *)rgb, l.alpha ? 4 : 3);
      }
    } else {
      f << xyz[0] << " " << xyz[1] << " " << xyz[2];
      if (l.rgb >= 0) {
        f << " " << (int)c[2] << " " << (int)c[1] << " " << (int)c[0];
        if (l.alpha) f << " " << (int)c[3];
      }
      f << "\n";
    }
  }
  for (const auto &poly : mesh.polygons) {
    if (binary) {
      const unsigned char n = (unsigned char)poly.vertices.size();
      f.write((const char *)&n, 1);
      for (std::uint32_t v : poly.vertices) {
        const std::int32_t iv = (std::int32_t)v;
        f.write((const char *)&iv, 4);
      }
    } else {
      f << poly.vertices.size();
      for (std::uint32_t v : poly.vertices) f << " " << v;
      f << "\n";
    }
  }
  return f ? 0 : -1;
}
}  // namespace detail

inline int savePLYFile(const std::string &file, const PolygonMesh &mesh, unsigned precision = 5) {
  return detail::save_ply(file, mesh, false, precision);
}
inline int savePLYFileBinary(const std::string &file, const PolygonMesh &mesh) {
  return detail::save_ply(file, mesh, true, 5);
}

}  // namespace io

// ---- pcl/search/kdtree.h: only the radius search by point index, on x, y, z ------------------------------
namespace search {
template <typename PointT>
class KdTree {
 public:
  typedef boost::shared_ptr<KdTree<PointT> > Ptr;
  typedef typename PointCloud<PointT>::ConstPtr CloudConstPtr;
  explicit KdTree(bool sorted = true) : sorted_(sorted), cell_(0) {}
  void setInputCloud(const CloudConstPtr &cloud) {
    cloud_ = cloud;
    cell_ = 0;
  }
  bool getSortedResults() const { return sorted_; }
  // Points strictly closer than `radius` to point `index`, nearest first (ties by index); squared distances.
  int radiusSearch(int index, double radius, std::vector<int> &k_indices, std::vector<float> &k_sqr_distances,
                   unsigned max_nn = 0) const {
    (void)max_nn;
    k_indices.clear();
    k_sqr_distances.clear();
    if (!cloud_ || radius <= 0) return 0;
    if (cell_ != radius) build(radius);
    const PointT &q = cloud_->points[index];
    if (!std::isfinite(q.x) || !std::isfinite(q.y) || !std::isfinite(q.z)) return 0;
    const long cx = coord(q.x), cy = coord(q.y), cz = coord(q.z);
    std::vector<std::pair<float, int> > found;
    const float r2 = (float)(radius * radius);
    for (long dz = -1; dz <= 1; ++dz)
      for (long dy = -1; dy <= 1; ++dy)
        for (long dx = -1; dx <= 1; ++dx) {
          const auto it = grid_.find(key(cx + dx, cy + dy, cz + dz));
          if (it == grid_.end()) continue;
          for (int j : it->second) {
            const PointT &p = cloud_->points[j];
            const float ex = p.x - q.x, ey = p.y - q.y, ez = p.z - q.z;
            const float d2 = ex * ex + ey * ey + ez * ez;
            if (d2 < r2) found.push_back({d2, j});
          }
        }
    std::sort(found.begin(), found.end());
    for (const auto &fj : found) {
      k_sqr_distances.push_back(fj.first);
      k_indices.push_back(fj.second);
    }
    return (int)found.size();
  }

 private:
  long coord(float v) const { return (long)std::floor((double)v / cell_); }
  static std::uint64_t key(long x, long y, long z) {
    return ((std::uint64_t)(x & 0x1fffff) << 42) | ((std::uint64_t)(y & 0x1fffff) << 21) | (std::uint64_t)(z & 0x1fffff);
  }
  void build(double radius) const {
    cell_ = radius;
    grid_.clear();
    for (size_t i = 0; i < cloud_->points.size(); ++i) {
      const PointT &p = cloud_->points[i];
      if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
      grid_[key(coord(p.x), coord(p.y), coord(p.z))].push_back((int)i);
    }
  }
  bool sorted_;
  CloudConstPtr cloud_;
  mutable double cell_;
  mutable std::unordered_map<std::uint64_t, std::vector<int> > grid_;
};
}  // namespace search

// ---- pcl/segmentation/extract_clusters.h: region growing over the radius graph -----------------------------
template <typename PointT>
class EuclideanClusterExtraction {
 public:
  typedef typename PointCloud<PointT>::ConstPtr CloudConstPtr;
  typedef typename search::KdTree<PointT>::Ptr KdTreePtr;
  EuclideanClusterExtraction() : tolerance_(0), min_size_(1), max_size_(2147483647) {}
  void setInputCloud(const CloudConstPtr &c) { cloud_ = c; }
  template <typename P>
  void setInputCloud(const boost::shared_ptr<P> &c) { cloud_ = c; }
  void setSearchMethod(const KdTreePtr &t) { tree_ = t; }
  void setClusterTolerance(double t) { tolerance_ = t; }
  void setMinClusterSize(int n) { min_size_ = n; }
  void setMaxClusterSize(int n) { max_size_ = n; }
  void extract(std::vector<PointIndices> &clusters) {
    clusters.clear();
    if (!cloud_ || !tree_) return;
    const size_t n = cloud_->points.size();
    std::vector<bool> processed(n, false);
    std::vector<int> nn;
    std::vector<float> d2;
    for (size_t i = 0; i < n; ++i) {
      if (processed[i]) continue;
      std::vector<int> queue(1, (int)i);
      processed[i] = true;
      for (size_t q = 0; q < queue.size(); ++q) {
        if (!tree_->radiusSearch(queue[q], tolerance_, nn, d2)) continue;
        for (int j : nn)
          if (!processed[j]) {
            queue.push_back(j);
            processed[j] = true;
          }
      }
      if ((int)queue.size() >= min_size_ && (int)queue.size() <= max_size_) {
        PointIndices r;
        r.indices = queue;
        std::sort(r.indices.begin(), r.indices.end());
        r.header = cloud_->header;
        clusters.push_back(r);
      }
    }
    std::stable_sort(clusters.begin(), clusters.end(),
                     [](const PointIndices &a, const PointIndices &b) { return a.indices.size() > b.indices.size(); });
  }

 private:
  CloudConstPtr cloud_;
  KdTreePtr tree_;
  double tolerance_;
  int min_size_, max_size_;
};

// ---- pcl/filters/voxel_grid.h: pass-through (only the --cloud-only aggregate uses it) ----------------------
template <typename PointT>
class VoxelGrid {
 public:
  void setLeafSize(float, float, float) {}
  template <typename P>
  void setInputCloud(const boost::shared_ptr<P> &c) { cloud_ = c; }
  void filter(PointCloud<PointT> &out) {
    if (cloud_ && cloud_.get() != &out) out = *cloud_;
  }

 private:
  typename PointCloud<PointT>::ConstPtr cloud_;
};

}  // namespace pcl
