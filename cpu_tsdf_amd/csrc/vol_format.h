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
// reference does); M_ and nsample_ -- only used by the reference's variance weighting, which only a loaded file can
// switch on -- are written as 0 unless the caller supplies them through the optional VarFn side channel (volumes whose
// header says weight_by_variance: tsdf_hip_save / tsdf_hip_load do), in which case they also take part in "same".  Reading rasterises every leaf over the voxels it covers.  Needs a cubic power-of-two
// grid (the only kind the reference's octree represents faithfully).
//
// Both directions stream: the grid is visited in cubic chunks (edge `chunk`, a power of two) through a
// fetch / store callback, so a 2048^3 volume (94 GB of voxels) is saved and loaded with a few hundred MB
// of host memory.  The writer makes two passes -- one that classifies every chunk as uniform or not (the
// top of the tree is built from that table), one that emits nodes in pre-order and re-fetches only the
// non-uniform chunks -- and produces byte for byte the file the whole-grid writer would.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <iomanip>
#include <atomic>
#include <sstream>
#include <string>
#include <thread>
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

// f(i) for i in [0, n) on up to 16 threads (plain std::thread: this header is compiled into libtsdf_hip.so,
// which must not drag a second OpenMP runtime into the caller's process)
template <class F>
inline void par_for(int n, const F &f) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int t = (int)std::min<unsigned>(hw ? hw : 1u, 16u);
  if (n < 4 || t < 2) {
    for (int i = 0; i < n; ++i) f(i);
    return;
  }
  std::atomic<int> next(0);
  std::vector<std::thread> th;
  for (int k = 0; k < std::min(t, n); ++k)
    th.emplace_back([&]() {
      for (int i; (i = next.fetch_add(1)) < n;) f(i);
    });
  for (auto &x : th) x.join();
}

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
  const float *M = nullptr;     // optional: OctreeNode::M_ / nsample_ per voxel (variance weighting)
  const int32_t *ns = nullptr;
  // uniform[l][node] for levels 0..L-1 (level l has (2^l)^3 nodes); a node is uniform when all voxels
  // below it are bit-identical
  std::vector<std::vector<unsigned char> > uniform;
  // optional, one flag per voxel of this grid: "the finer data this voxel stands for is uniform" (the
  // streaming writer's chunk table); a node over a voxel with flag 0 is never uniform
  const std::vector<unsigned char> *leaf_ok = nullptr;
  size_t vox(int x, int y, int z) const { return ((size_t)z * n + y) * n + x; }
  bool same(size_t a, size_t b) const {
    return std::memcmp(d + a, d + b, 4) == 0 && std::memcmp(w + a, w + b, 4) == 0 &&
           (!rgb || std::memcmp(rgb + 3 * a, rgb + 3 * b, 3) == 0) &&
           (!M || (std::memcmp(M + a, M + b, 4) == 0 && ns[a] == ns[b]));
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
    auto plane = [&](int kz) {
      for (int ky = 0; ky < m; ++ky)
        for (int kx = 0; kx < m; ++kx) {
          bool u = true;
          const size_t ref = g.vox(kx * span, ky * span, kz * span);
          for (int c = 0; c < 8 && u; ++c) {
            const int cx = 2 * kx + ((c >> 2) & 1), cy = 2 * ky + ((c >> 1) & 1), cz = 2 * kz + (c & 1);
            if (below) u = (*below)[((size_t)cz * 2 * m + cy) * 2 * m + cx] != 0;
            else if (g.leaf_ok) u = (*g.leaf_ok)[g.vox(cx, cy, cz)] != 0;
            if (u) u = g.same(ref, g.vox(cx * half, cy * half, cz * half));
          }
          g.uniform[l][((size_t)kz * m + ky) * m + kx] = u ? 1 : 0;
        }
    };
    if (m >= 32)
      par_for(m, plane);
    else
      for (int kz = 0; kz < m; ++kz) plane(kz);
  }
}

template <typename T>
inline void put(std::ostream &f, const T &v) {
  f.write(reinterpret_cast<const char *>(&v), sizeof(T));
}

// Node records are 40 / 43 bytes and a big volume has hundreds of millions of them: they are gathered
// into a 4 MB block before they reach the stream.
struct NodeSink {
  std::ostream *f;  // null: a memory sink -- the records stay in `mem`, which grows (a subtree serialised by a worker
                    // thread; plain malloc / realloc: no zero fill, and the capacity survives reset() for the next block)
  std::vector<char> buf;
  char *mem = nullptr;
  size_t cap = 0, n = 0;
  explicit NodeSink(std::ostream &s) : f(&s), buf(4u << 20) {}
  NodeSink() : f(nullptr) {}
  NodeSink(const NodeSink &) = delete;
  NodeSink &operator=(const NodeSink &) = delete;
  ~NodeSink() { std::free(mem); }
  void reset() { n = 0; }
  void flush() {
    if (!f) return;
    if (n) f->write(buf.data(), (std::streamsize)n);
    n = 0;
  }
  char *room(size_t bytes) {
    if (f) {
      if (n + bytes > buf.size()) flush();
      char *p = buf.data() + n;
      n += bytes;
      return p;
    }
    if (n + bytes > cap) {
      const size_t want = std::max<size_t>((size_t)1 << 20, cap + cap / 2 + bytes);
      char *m2 = static_cast<char *>(std::realloc(mem, want));
      if (!m2) throw std::bad_alloc();
      mem = m2, cap = want;
    }
    char *p = mem + n;
    n += bytes;
    return p;
  }
  void append(const NodeSink &part) {  // (a stream sink only) the records of a memory sink, in one write
    flush();
    if (part.n) f->write(part.mem, (std::streamsize)part.n);
  }
};

inline void put_node(NodeSink &f, bool color, const unsigned char *rgb, float d, float w, float cx, float cy,
                     float cz, float size, bool leaf, float M = 0.f, int32_t ns = 0) {
  char *p = f.room(color ? 43 : 40);
  if (color) {
    const unsigned char zero[3] = {0, 0, 0};
    std::memcpy(p, leaf ? rgb : zero, 3);
    p += 3;
  }
  if (!leaf) {
    d = -1.f;
    w = 0.f;
    M = 0.f;
    ns = 0;
  }
  const float rec[7] = {d, w, cx, cy, cz, size, M};
  const int32_t nsample = ns;
  const size_t nchild = leaf ? 0 : 8;
  std::memcpy(p, rec, 28);
  std::memcpy(p + 28, &nsample, 4);
  std::memcpy(p + 32, &nchild, 8);
}

inline void write_node(NodeSink &f, const Grid &g, int level, int kx, int ky, int kz, float cx, float cy,
                       float cz, float size) {
  const int span = g.n >> level;
  const bool leaf = level == g.L || g.uniform[level][((size_t)kz * (1 << level) + ky) * (1 << level) + kx];
  const size_t v0 = g.vox(kx * span, ky * span, kz * span);
  put_node(f, g.rgb != nullptr, g.rgb ? g.rgb + 3 * v0 : nullptr, g.d[v0], g.w[v0], cx, cy, cz, size, leaf, g.M ? g.M[v0] : 0.f,
           g.ns ? g.ns[v0] : 0);
  if (leaf) return;
  const float off = size / 4, ns = size / 2;
  for (int c = 0; c < 8; ++c) {
    const int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
    write_node(f, g, level + 1, 2 * kx + bx, 2 * ky + by, 2 * kz + bz, bx ? cx + off : cx - off,
               by ? cy + off : cy - off, bz ? cz + off : cz - off, ns);
  }
}

// write_node(f, g, 0, ...) for a big block, with the serialisation spread over threads: pre-order puts the records of
// the 64 subtrees below level 2 one after another, so the workers fill one memory sink per subtree (a subtree whose
// parent or grandparent is a uniform leaf has no records) and this thread writes the level-0 / level-1 records and
// the parts in order.  Byte for byte what write_node writes; the centres descend by the same float operations.
inline void write_block(NodeSink &f, const Grid &g, float cx, float cy, float cz, float size, std::vector<NodeSink> &parts) {
  if (g.L < 5) return write_node(f, g, 0, 0, 0, 0, cx, cy, cz, size);
  const bool color = g.rgb != nullptr;
  auto child = [](int k, int &kx, int &ky, int &kz, float &x, float &y, float &z, float &sz) {
    const int bx = (k >> 2) & 1, by = (k >> 1) & 1, bz = k & 1;
    const float off = sz / 4;
    kx = 2 * kx + bx, ky = 2 * ky + by, kz = 2 * kz + bz;
    x = bx ? x + off : x - off, y = by ? y + off : y - off, z = bz ? z + off : z - off;
    sz = sz / 2;
  };
  auto node = [&](NodeSink &s, int level, int kx, int ky, int kz, float x, float y, float z, float sz) -> bool {  // true: a leaf
    const int span = g.n >> level;
    const bool leaf = g.uniform[level][((size_t)kz * (1 << level) + ky) * (1 << level) + kx] != 0;
    const size_t v0 = g.vox(kx * span, ky * span, kz * span);
    put_node(s, color, color ? g.rgb + 3 * v0 : nullptr, g.d[v0], g.w[v0], x, y, z, sz, leaf, g.M ? g.M[v0] : 0.f, g.ns ? g.ns[v0] : 0);
    return leaf;
  };
  if (parts.size() != 64) parts = std::vector<NodeSink>(64);
  for (NodeSink &p : parts) p.reset();
  const bool root_leaf = g.uniform[0][0] != 0;
  if (!root_leaf)
    par_for(64, [&](int i) {
      int kx = 0, ky = 0, kz = 0;
      float x = cx, y = cy, z = cz, sz = size;
      child(i >> 3, kx, ky, kz, x, y, z, sz);
      if (g.uniform[1][((size_t)kz * 2 + ky) * 2 + kx]) return;
      child(i & 7, kx, ky, kz, x, y, z, sz);
      write_node(parts[i], g, 2, kx, ky, kz, x, y, z, sz);
    });
  if (node(f, 0, 0, 0, 0, cx, cy, cz, size)) return;
  for (int k1 = 0; k1 < 8; ++k1) {
    int kx = 0, ky = 0, kz = 0;
    float x = cx, y = cy, z = cz, sz = size;
    child(k1, kx, ky, kz, x, y, z, sz);
    if (node(f, 1, kx, ky, kz, x, y, z, sz)) continue;
    for (int k2 = 0; k2 < 8; ++k2) f.append(parts[k1 * 8 + k2]);
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

struct NodeSource {  // the reading counterpart of NodeSink
  std::istream &f;
  std::vector<char> buf;
  size_t pos, end;
  explicit NodeSource(std::istream &s) : f(s), buf(4u << 20), pos(0), end(0) {}
  const char *take(size_t bytes) {  // null at end of file
    if (end - pos < bytes) {
      std::memmove(buf.data(), buf.data() + pos, end - pos);
      end -= pos;
      pos = 0;
      f.read(buf.data() + end, (std::streamsize)(buf.size() - end));
      end += (size_t)f.gcount();
      if (end < bytes) return nullptr;
    }
    const char *p = buf.data() + pos;
    pos += bytes;
    return p;
  }
};

struct MemSource {  // records already in memory (one subtree of an open chunk, parsed by a worker thread)
  const char *p, *end;
  const char *take(size_t bytes) {
    if ((size_t)(end - p) < bytes) return nullptr;
    const char *r = p;
    p += bytes;
    return r;
  }
};

// One cubic block of voxels, edge c, origin (x0,y0,z0): d, w [c^3] and rgb [3 c^3] (null without colour),
// x fastest.  fetch fills the buffers from the volume, store writes them into it; both return false on error.
typedef std::function<bool(int x0, int y0, int z0, int c, float *d, float *w, unsigned char *rgb)> BlockFn;
// Optional side channel next to a BlockFn: OctreeNode::M_ / nsample_ of the same block ([z][y][x] like d), fetched right
// after the writer fetched the block, stored right after the reader stored it.  Empty = not carried (written as 0).
typedef std::function<bool(int x0, int y0, int z0, int c, float *M, int32_t *nsample)> VarFn;

struct ReadCtx {
  int n, C, Lc;  // grid edge, chunk edge, tree level whose nodes are chunks
  float vs, half;
  bool color;
  bool has_var = false;
  std::vector<float> d, w;  // the chunk being assembled (owner only; the parsers write through the views below)
  std::vector<unsigned char> rgb;
  std::vector<float> M;       // only with `var`
  std::vector<int32_t> ns;
  float *pd = nullptr, *pw = nullptr, *pM = nullptr;
  unsigned char *prgb = nullptr;
  int32_t *pns = nullptr;
  int ox, oy, oz;  // its origin; ox < 0: none open
  BlockFn store;
  VarFn var;
  std::string err;
  std::vector<char> sub;  // the records of the open chunk's subtree while worker threads parse them
  void bind() {
    pd = d.data(), pw = w.data(), prgb = rgb.data(), pM = M.data(), pns = ns.data();
    has_var = (bool)var;
  }
  // what a worker thread needs: the geometry and the views, none of the vectors
  void view_of(const ReadCtx &o) {
    n = o.n, C = o.C, Lc = o.Lc, vs = o.vs, half = o.half, color = o.color, has_var = o.has_var;
    pd = o.pd, pw = o.pw, pM = o.pM, prgb = o.prgb, pns = o.pns;
    ox = o.ox, oy = o.oy, oz = o.oz;
  }
  bool store_block(int x, int y, int z) {
    if (!store(x, y, z, C, d.data(), w.data(), color ? rgb.data() : nullptr)) return false;
    return !var || var(x, y, z, C, M.data(), ns.data());
  }
};

inline bool node_box(const ReadCtx &c, float cx, float cy, float cz, float size, int &x0, int &y0, int &z0, int &span) {
  span = std::max(1, (int)std::lround(size / c.vs));
  x0 = (int)std::lround((cx - size / 2 + c.half) / c.vs);
  y0 = (int)std::lround((cy - size / 2 + c.half) / c.vs);
  z0 = (int)std::lround((cz - size / 2 + c.half) / c.vs);
  return x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + span <= c.n && y0 + span <= c.n && z0 + span <= c.n;
}

inline void fill_chunk(ReadCtx &c, int x0, int y0, int z0, int span, float d, float w, const unsigned char *col, float M = 0.f,
                       int32_t ns = 0) {
  for (int z = z0; z < z0 + span; ++z)
    for (int y = y0; y < y0 + span; ++y) {
      const size_t row = ((size_t)z * c.C + y) * c.C + x0;
      for (int x = 0; x < span; ++x) {
        c.pd[row + x] = d;
        c.pw[row + x] = w;
      }
      if (c.has_var)
        for (int x = 0; x < span; ++x) {
          c.pM[row + x] = M;
          c.pns[row + x] = ns;
        }
      if (c.color)
        for (int x = 0; x < span; ++x) std::memcpy(c.prgb + 3 * (row + x), col, 3);
    }
}

// Appends the records of ONE subtree (pre-order, `rec` bytes each) from f to c.sub without interpreting them beyond the
// child count; `budget` = records the open chunk may still hold (bounds memory on a malformed file).
template <class S>
inline bool scan_subtree(S &f, ReadCtx &c, size_t rec, size_t &used, size_t &budget) {
  size_t pending = 1;
  while (pending) {
    const char *p = f.take(rec);
    if (!p) {
      c.err = "truncated octree";
      return false;
    }
    if (!budget) {
      c.err = "octree deeper than the grid";
      return false;
    }
    --budget;
    if (used + rec > c.sub.size()) c.sub.resize(std::max<size_t>((size_t)4 << 20, 2 * c.sub.size()));
    std::memcpy(c.sub.data() + used, p, rec);
    used += rec;
    size_t nchild;
    std::memcpy(&nchild, p + rec - 8, 8);
    if (nchild == 8) pending += 8;
    else if (nchild != 0) {
      c.err = "malformed octree node";
      return false;
    }
    --pending;
  }
  return true;
}

template <class S>
inline bool read_node(S &f, ReadCtx &c, int depth);

// The eight children of a chunk node that has just been opened, parsed on worker threads: the records of the chunk's
// subtree are copied out of the stream (structure only: child counts), split at the grandchildren -- in pre-order the
// 64 subtrees below level 2 follow one another -- and every piece fills its own part of the chunk.  Same checks, same
// voxels as the recursion; the first error in pre-order is the one reported.
template <class S>
inline bool read_open_chunk_parallel(S &f, ReadCtx &c, int depth) {
  const size_t rec = c.color ? 43 : 40;
  size_t used = 0, budget = 0;
  for (int l = 1; (c.C >> l) >= 1; ++l) budget += (size_t)1 << (3 * l);  // 8 + 64 + ... + C^3: every node a chunk's subtree can hold
  struct Piece {
    size_t a, b;
    int depth;
  };
  std::vector<Piece> pieces;
  for (int k1 = 0; k1 < 8; ++k1) {
    const size_t a1 = used;
    const char *p = f.take(rec);
    if (!p) {
      c.err = "truncated octree";
      return false;
    }
    if (!budget) {
      c.err = "octree deeper than the grid";
      return false;
    }
    --budget;
    if (used + rec > c.sub.size()) c.sub.resize(std::max<size_t>((size_t)4 << 20, 2 * c.sub.size()));
    std::memcpy(c.sub.data() + used, p, rec);
    used += rec;
    size_t nchild;
    std::memcpy(&nchild, p + rec - 8, 8);
    if (nchild == 0) {
      pieces.push_back({a1, used, depth + 1});  // a leaf child: one record
    } else if (nchild == 8) {
      if (depth + 1 > 24) {  // the one check read_node makes on an inner record that is not itself handed to it here
        c.err = "malformed octree node";
        return false;
      }
      for (int k2 = 0; k2 < 8; ++k2) {
        const size_t a2 = used;
        if (!scan_subtree(f, c, rec, used, budget)) return false;
        pieces.push_back({a2, used, depth + 2});
      }
    } else {
      c.err = "malformed octree node";
      return false;
    }
  }
  std::vector<std::string> errs(pieces.size());
  std::vector<unsigned char> ok(pieces.size(), 0);
  const char *base = c.sub.data();
  par_for((int)pieces.size(), [&](int i) {
    ReadCtx t;
    t.view_of(c);
    MemSource ms{base + pieces[i].a, base + pieces[i].b};
    ok[i] = read_node(ms, t, pieces[i].depth) ? 1 : 0;
    if (!ok[i]) errs[i] = t.err;
  });
  for (size_t i = 0; i < pieces.size(); ++i)
    if (!ok[i]) {
      c.err = errs[i];
      return false;
    }
  return true;
}

template <class S>
inline bool read_node(S &f, ReadCtx &c, int depth) {
  unsigned char col[3] = {0, 0, 0};
  const char *p = f.take(c.color ? 43 : 40);
  if (!p) {
    c.err = "truncated octree";
    return false;
  }
  if (c.color) {
    std::memcpy(col, p, 3);
    p += 3;
  }
  float rec[7];  // d, w, centre, size, M; then nsample
  int32_t nsample;
  size_t nchild;
  std::memcpy(rec, p, 28);
  std::memcpy(&nsample, p + 28, 4);
  std::memcpy(&nchild, p + 32, 8);
  const float d = rec[0], w = rec[1], cx = rec[2], cy = rec[3], cz = rec[4], size = rec[5], Mv = rec[6];
  int x0, y0, z0, span;
  if (nchild == 0) {  // leaf: fill the voxels it covers
    if (!node_box(c, cx, cy, cz, size, x0, y0, z0, span)) {
      c.err = "leaf outside the grid";
      return false;
    }
    if (c.ox >= 0) {  // inside the open chunk
      if (x0 < c.ox || y0 < c.oy || z0 < c.oz || x0 + span > c.ox + c.C || y0 + span > c.oy + c.C ||
          z0 + span > c.oz + c.C) {
        c.err = "leaf outside its parent node";
        return false;
      }
      fill_chunk(c, x0 - c.ox, y0 - c.oy, z0 - c.oz, span, d, w, col, Mv, nsample);
      return true;
    }
    if (span < c.C || x0 % c.C || y0 % c.C || z0 % c.C || span % c.C) {
      c.err = "octree nodes are not aligned with their depth";
      return false;
    }
    fill_chunk(c, 0, 0, 0, c.C, d, w, col, Mv, nsample);  // one constant chunk, stored over every chunk the leaf covers
    for (int z = z0; z < z0 + span; z += c.C)
      for (int y = y0; y < y0 + span; y += c.C)
        for (int x = x0; x < x0 + span; x += c.C)
          if (!c.store_block(x, y, z)) {
            c.err = "storing a block failed";
            return false;
          }
    return true;
  }
  if (nchild != 8 || depth > 24) {
    c.err = "malformed octree node";
    return false;
  }
  const bool opens = c.ox < 0 && depth == c.Lc;
  if (opens) {
    if (!node_box(c, cx, cy, cz, size, x0, y0, z0, span) || span != c.C || x0 % c.C || y0 % c.C || z0 % c.C) {
      c.err = "octree nodes are not aligned with their depth";
      return false;
    }
    c.ox = x0;
    c.oy = y0;
    c.oz = z0;
    fill_chunk(c, 0, 0, 0, c.C, -1.f, 0.f, col);  // (the eight children overwrite all of it)
  } else if (c.ox < 0 && depth > c.Lc) {
    c.err = "octree deeper than the grid";
    return false;
  }
  if (opens && c.C >= 32) {
    if (!read_open_chunk_parallel(f, c, depth)) return false;
  } else {
    for (int k = 0; k < 8; ++k)
      if (!read_node(f, c, depth + 1)) return false;
  }
  if (opens) {
    const bool ok = c.store_block(c.ox, c.oy, c.oz);
    c.ox = -1;
    if (!ok) {
      c.err = "storing a block failed";
      return false;
    }
  }
  return true;
}

// largest power of two <= want that divides n (n a power of two)
inline int chunk_edge(int n, int want) {
  int c = 1;
  while (2 * c <= want && 2 * c <= n) c *= 2;
  return c;
}

struct WriteCtx {
  NodeSink *f;
  bool color;
  int C, Lc, LC;  // chunk edge, chunk level, log2(C)
  Grid top;       // one voxel per chunk (its first voxel) + the chunk-uniform table
  std::vector<float> d, w;
  std::vector<unsigned char> rgb;
  std::vector<float> M;  // only with `var`
  std::vector<int32_t> ns;
  BlockFn fetch;
  VarFn var;
  std::string err;
  std::vector<NodeSink> parts;  // write_block's per-subtree memory sinks, kept from block to block
  bool fetch_block(int x, int y, int z) {
    if (!fetch(x, y, z, C, d.data(), w.data(), color ? rgb.data() : nullptr)) return false;
    return !var || var(x, y, z, C, M.data(), ns.data());
  }
};

inline bool write_top(WriteCtx &c, int level, int kx, int ky, int kz, float cx, float cy, float cz, float size) {
  const int m = 1 << level;
  const int span = c.top.n >> level;  // chunks per node edge
  const size_t v0 = c.top.vox(kx * span, ky * span, kz * span);
  const bool leaf = level == c.Lc ? (*c.top.leaf_ok)[v0] != 0 : c.top.uniform[level][((size_t)kz * m + ky) * m + kx] != 0;
  if (leaf || level < c.Lc) {
    put_node(*c.f, c.color, c.color ? c.top.rgb + 3 * v0 : nullptr, c.top.d[v0], c.top.w[v0], cx, cy, cz, size, leaf,
             c.top.M ? c.top.M[v0] : 0.f, c.top.ns ? c.top.ns[v0] : 0);
    if (leaf) return true;
    const float off = size / 4, ns = size / 2;
    for (int k = 0; k < 8; ++k) {
      const int bx = (k >> 2) & 1, by = (k >> 1) & 1, bz = k & 1;
      if (!write_top(c, level + 1, 2 * kx + bx, 2 * ky + by, 2 * kz + bz, bx ? cx + off : cx - off,
                     by ? cy + off : cy - off, bz ? cz + off : cz - off, ns))
        return false;
    }
    return true;
  }
  // a non-uniform chunk: fetch it again and write its subtree
  if (!c.fetch_block(kx * c.C, ky * c.C, kz * c.C)) {
    c.err = "fetching a block failed";
    return false;
  }
  Grid g;
  g.n = c.C;
  g.L = c.LC;
  g.d = c.d.data();
  g.w = c.w.data();
  g.rgb = c.color ? c.rgb.data() : nullptr;
  g.M = c.var ? c.M.data() : nullptr;
  g.ns = c.var ? c.ns.data() : nullptr;
  build_pyramid(g);
  write_block(*c.f, g, cx, cy, cz, size, c.parts);
  return true;
}

inline bool all_same(const Grid &g) {
  std::atomic<int> differ(0);
  auto plane = [&](int z) {
    if (differ.load(std::memory_order_relaxed)) return;
    const size_t a = (size_t)z * g.n * g.n, b = a + (size_t)g.n * g.n;
    for (size_t i = a; i < b; ++i)
      if (!g.same(0, i)) {
        differ.store(1, std::memory_order_relaxed);
        return;
      }
  };
  if (g.n >= 32)
    par_for(g.n, plane);
  else
    for (int z = 0; z < g.n; ++z) plane(z);
  return !differ.load();
}

}  // namespace volfmt

// Writes the volume `fetch` serves into `filename`, visiting it in blocks of edge <= chunk.
inline bool vol_write_stream(const std::string &filename, const VolHeader &h, int chunk, const volfmt::BlockFn &fetch,
                             std::string *err, const volfmt::VarFn &var = volfmt::VarFn()) {
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

  volfmt::NodeSink sink(f);
  volfmt::WriteCtx c;
  c.f = &sink;
  c.color = h.color;
  c.C = volfmt::chunk_edge(h.res[0], chunk);
  c.LC = volfmt::log2_exact(c.C);
  c.Lc = L - c.LC;
  c.fetch = fetch;
  c.var = var;
  const size_t cv = (size_t)c.C * c.C * c.C;
  c.d.resize(cv);
  c.w.resize(cv);
  c.rgb.resize(h.color ? 3 * cv : 0);
  c.M.resize(var ? cv : 0);
  c.ns.resize(var ? cv : 0);
  // pass 1: one voxel and one "uniform" flag per chunk
  const int m = 1 << c.Lc;
  std::vector<float> td((size_t)m * m * m), tw((size_t)m * m * m), tM(var ? td.size() : 0);
  std::vector<int32_t> tns(var ? td.size() : 0);
  std::vector<unsigned char> trgb(h.color ? 3 * td.size() : 0), tok(td.size());
  volfmt::Grid blk;
  blk.n = c.C;
  blk.L = c.LC;
  blk.d = c.d.data();
  blk.w = c.w.data();
  blk.rgb = h.color ? c.rgb.data() : nullptr;
  blk.M = var ? c.M.data() : nullptr;
  blk.ns = var ? c.ns.data() : nullptr;
  for (int kz = 0; kz < m; ++kz)
    for (int ky = 0; ky < m; ++ky)
      for (int kx = 0; kx < m; ++kx) {
        if (!c.fetch_block(kx * c.C, ky * c.C, kz * c.C)) {
          if (err) *err = "fetching a block failed";
          return false;
        }
        const size_t t = ((size_t)kz * m + ky) * m + kx;
        td[t] = c.d[0];
        tw[t] = c.w[0];
        if (h.color) std::memcpy(&trgb[3 * t], c.rgb.data(), 3);
        if (var) tM[t] = c.M[0], tns[t] = c.ns[0];
        tok[t] = volfmt::all_same(blk) ? 1 : 0;
      }
  c.top.n = m;
  c.top.L = c.Lc;
  c.top.d = td.data();
  c.top.w = tw.data();
  c.top.rgb = h.color ? trgb.data() : nullptr;
  c.top.M = var ? tM.data() : nullptr;
  c.top.ns = var ? tns.data() : nullptr;
  c.top.leaf_ok = &tok;
  volfmt::build_pyramid(c.top);
  // pass 2: nodes in pre-order
  if (!volfmt::write_top(c, 0, 0, 0, 0, 0.f, 0.f, 0.f, h.size[0])) {
    if (err) *err = c.err;
    return false;
  }
  sink.flush();
  f.close();
  if (!f) {
    if (err) *err = "write error on " + filename;
    return false;
  }
  return true;
}

// Reads `filename`: the header first (on_header gets it and prepares the volume; false aborts), then the
// voxels, handed to `store` in blocks of edge min(chunk, res).  Every voxel is stored exactly once.
inline bool vol_read_stream(const std::string &filename, VolHeader &h, int chunk,
                            const std::function<bool(const VolHeader &)> &on_header, const volfmt::BlockFn &store,
                            std::string *err, const volfmt::VarFn &var = volfmt::VarFn()) {
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
  if (!on_header(h)) {
    if (err) *err = "the volume could not be prepared";
    return false;
  }
  volfmt::ReadCtx c;
  c.n = h.res[0];
  c.C = volfmt::chunk_edge(c.n, chunk);
  c.Lc = volfmt::log2_exact(c.n) - volfmt::log2_exact(c.C);
  c.vs = s3[0] / (float)h.res[0];
  c.half = s3[0] / 2;
  c.color = h.color;
  const size_t cv = (size_t)c.C * c.C * c.C;
  c.d.resize(cv);
  c.w.resize(cv);
  c.rgb.resize(h.color ? 3 * cv : 0);
  c.ox = c.oy = c.oz = -1;
  c.store = store;
  c.var = (var && h.weight_by_variance) ? var : volfmt::VarFn();  // only a file that weights by variance carries meaningful values
  c.M.resize(c.var ? cv : 0);
  c.ns.resize(c.var ? cv : 0);
  c.bind();
  volfmt::NodeSource src(f);
  if (!volfmt::read_node(src, c, 0)) {
    if (err) *err = c.err;
    return false;
  }
  return true;
}

}  // namespace cpu_tsdf
