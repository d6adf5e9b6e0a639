// Mesh post-processing of the `integrate` program (--flatten / --cleanup), re-implemented without a
// kd-tree library: both passes only ever ask "which points lie strictly within r of this one", which a
// uniform hash grid of cell r answers from the 27 surrounding cells.
//
// Contracts reproduced (src/prog/integrate.cpp):
//   flattenVertices (:103-158): vertices are visited in index order; an unassigned vertex opens a new output
//     vertex and EVERY vertex strictly within min_dist of it -- assigned or not -- is (re)mapped to it; faces
//     are re-indexed, a face with two equal corners is dropped, face order is kept.  Colours are lost (the
//     reference converts the cloud to PointXYZ).
//   cleanupMesh (:160-237): faces whose centroids form a connected group (links: centroid distance strictly
//     below face_dist) of at most min_neighbors faces are removed; vertices no face uses are removed; vertex
//     and face order are kept.  Colours are lost likewise.
#pragma once

#include <pcl/PolygonMesh.h>
#include <pcl/conversions.h>
#include <pcl/point_types.h>

#include <cmath>
#include <cstdint>
#include <unordered_map>
#include <vector>

namespace cpu_tsdf {
namespace mesh_post {

class PointGrid {
 public:
  PointGrid(const std::vector<float> &xyz, double cell) : xyz_(xyz), cell_(cell) {
    for (size_t i = 0; i < xyz.size() / 3; ++i)
      if (finite(i)) cells_[key(c(xyz[3 * i]), c(xyz[3 * i + 1]), c(xyz[3 * i + 2]))].push_back((int)i);
  }
  // calls f(j) for every j (including i) with |p_j - p_i|^2 < r2, computed in float like a kd-tree would
  template <typename F>
  void forNeighbours(size_t i, float r2, F f) const {
    if (!finite(i)) return;
    const float x = xyz_[3 * i], y = xyz_[3 * i + 1], z = xyz_[3 * i + 2];
    const long cx = c(x), cy = c(y), cz = c(z);
    for (long dz = -1; dz <= 1; ++dz)
      for (long dy = -1; dy <= 1; ++dy)
        for (long dx = -1; dx <= 1; ++dx) {
          const auto it = cells_.find(key(cx + dx, cy + dy, cz + dz));
          if (it == cells_.end()) continue;
          for (int j : it->second) {
            const float ex = xyz_[3 * j] - x, ey = xyz_[3 * j + 1] - y, ez = xyz_[3 * j + 2] - z;
            if (ex * ex + ey * ey + ez * ez < r2) f(j);
          }
        }
  }

 private:
  bool finite(size_t i) const {
    return std::isfinite(xyz_[3 * i]) && std::isfinite(xyz_[3 * i + 1]) && std::isfinite(xyz_[3 * i + 2]);
  }
  long c(float v) const { return (long)std::floor((double)v / cell_); }
  static std::uint64_t key(long x, long y, long z) {
    return ((std::uint64_t)(x & 0x1fffff) << 42) | ((std::uint64_t)(y & 0x1fffff) << 21) | (std::uint64_t)(z & 0x1fffff);
  }
  const std::vector<float> &xyz_;
  double cell_;
  std::unordered_map<std::uint64_t, std::vector<int> > cells_;
};

inline std::vector<float> vertexPositions(const pcl::PolygonMesh &mesh) {
  pcl::PointCloud<pcl::PointXYZ> v;
  pcl::fromPCLPointCloud2(mesh.cloud, v);
  std::vector<float> xyz(3 * v.points.size());
  for (size_t i = 0; i < v.points.size(); ++i) {
    xyz[3 * i] = v.points[i].x;
    xyz[3 * i + 1] = v.points[i].y;
    xyz[3 * i + 2] = v.points[i].z;
  }
  return xyz;
}

inline void storeVertices(const std::vector<float> &xyz, const std::vector<int> &keep, pcl::PolygonMesh &mesh) {
  pcl::PointCloud<pcl::PointXYZ> out;
  for (int i : keep) {
    pcl::PointXYZ p;
    p.x = xyz[3 * i];
    p.y = xyz[3 * i + 1];
    p.z = xyz[3 * i + 2];
    out.push_back(p);
  }
  pcl::toPCLPointCloud2(out, mesh.cloud);
}

inline void flattenVertices(pcl::PolygonMesh &mesh, float min_dist = 0.0001f) {
  const std::vector<float> xyz = vertexPositions(mesh);
  const size_t n = xyz.size() / 3;
  const PointGrid grid(xyz, min_dist);
  const float r2 = (float)((double)min_dist * (double)min_dist);
  std::vector<int> remap(n, -1), seeds;
  for (size_t i = 0; i < n; ++i) {
    if (remap[i] >= 0) continue;
    const int idx = (int)seeds.size();
    remap[i] = idx;
    // the reference additionally requires the SQUARED distance to be below min_dist itself (:128)
    grid.forNeighbours(i, r2 < min_dist ? r2 : min_dist, [&](int j) { if ((size_t)j != i) remap[j] = idx; });
    seeds.push_back((int)i);
  }
  size_t kept = 0;
  for (size_t f = 0; f < mesh.polygons.size(); ++f) {
    pcl::Vertices &v = mesh.polygons[f];
    for (auto &k : v.vertices) k = (std::uint32_t)remap[k];
    if (v.vertices.size() >= 3 &&
        (v.vertices[0] == v.vertices[1] || v.vertices[1] == v.vertices[2] || v.vertices[2] == v.vertices[0]))
      continue;
    if (kept != f) mesh.polygons[kept] = mesh.polygons[f];
    ++kept;
  }
  mesh.polygons.resize(kept);
  storeVertices(xyz, seeds, mesh);
}

inline void cleanupMesh(pcl::PolygonMesh &mesh, float face_dist = 0.02f, int min_neighbors = 5) {
  const std::vector<float> xyz = vertexPositions(mesh);
  // centroids of the triangles (:78-92: (v0 + v1 + v2) / 3 in float)
  std::vector<float> cen;
  std::vector<size_t> tri;  // polygon index of each centroid
  for (size_t f = 0; f < mesh.polygons.size(); ++f) {
    const auto &v = mesh.polygons[f].vertices;
    if (v.size() != 3) continue;
    for (int k = 0; k < 3; ++k) cen.push_back(((xyz[3 * v[0] + k] + xyz[3 * v[1] + k]) + xyz[3 * v[2] + k]) / 3.f);
    tri.push_back(f);
  }
  const size_t nf = tri.size();
  const PointGrid grid(cen, face_dist);
  const float r2 = (float)((double)face_dist * (double)face_dist);
  std::vector<char> seen(nf, 0), drop(mesh.polygons.size(), 0);
  std::vector<int> group;
  for (size_t i = 0; i < nf; ++i) {
    if (seen[i]) continue;
    group.assign(1, (int)i);
    seen[i] = 1;
    for (size_t q = 0; q < group.size(); ++q)
      grid.forNeighbours((size_t)group[q], r2, [&](int j) {
        if (!seen[j]) {
          seen[j] = 1;
          group.push_back(j);
        }
      });
    if ((int)group.size() <= min_neighbors)
      for (int g : group) drop[tri[g]] = 1;
  }
  size_t kept = 0;
  for (size_t f = 0; f < mesh.polygons.size(); ++f) {
    if (drop[f]) continue;
    if (kept != f) mesh.polygons[kept] = mesh.polygons[f];
    ++kept;
  }
  mesh.polygons.resize(kept);
  const size_t n = xyz.size() / 3;
  std::vector<int> new_index(n, -1), keep;
  std::vector<char> used(n, 0);
  for (const auto &p : mesh.polygons)
    for (int k = 0; k < 3; ++k) used[p.vertices[k]] = 1;
  for (size_t i = 0; i < n; ++i)
    if (used[i]) {
      new_index[i] = (int)keep.size();
      keep.push_back((int)i);
    }
  for (auto &p : mesh.polygons)
    for (int k = 0; k < 3; ++k) p.vertices[k] = (std::uint32_t)new_index[p.vertices[k]];
  storeVertices(xyz, keep, mesh);
}

}  // namespace mesh_post
}  // namespace cpu_tsdf
