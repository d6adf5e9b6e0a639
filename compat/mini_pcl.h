// mini_pcl.h -- the slice of the Point Cloud Library API that cpu_tsdf's library code (and this repo's
// C++ host shell) touches, for machines without PCL.  NOT PCL.
//
// Layouts (sizes, field order, b,g,r,a byte order) follow PCL's point_types; arithmetic that is
// observable in results is restated from upstream PCL >= 1.10 and marked [PCL-recall] -- PCL is not
// vendored in the reference and not installed here, so none of it can be checked offline:
//   * pcl::transformPoint / transformPointCloud / ...WithNormals  (common/impl/transforms.hpp,
//     detail::Transformer, SSE2 build): se3: x*c0 + (y*c1 + (z*c2 + c3)), so3: x*c0 + (y*c1 + z*c2);
//     clouds with is_dense == false skip non-finite points
//   * pcl::FrustumCulling::applyFilter  (filters/impl/frustum_culling.hpp): six plane tests <= 0
//   * pcl::MarchingCubes::getBoundingBox / createSurface / interpolateEdge  (surface/impl/
//     marching_cubes.hpp) with Bourke's edge/triangle tables
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "mini_eigen.h"

namespace boost {
using std::const_pointer_cast;
using std::dynamic_pointer_cast;
using std::make_shared;
using std::shared_ptr;
using std::static_pointer_cast;
}  // namespace boost

// ---- pcl/console/print.h ------------------------------------------------------------------------
namespace pcl {
namespace console {
inline bool verbose() {
  static const bool v = std::getenv("MINI_PCL_VERBOSE") != nullptr;
  return v;
}
}  // namespace console
}  // namespace pcl
#define PCL_INFO(...)                                           \
  do {                                                          \
    if (pcl::console::verbose()) std::fprintf(stdout, __VA_ARGS__); \
  } while (0)
#define PCL_WARN(...) std::fprintf(stderr, __VA_ARGS__)
#define PCL_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
#define PCL_DEBUG(...) \
  do {                 \
  } while (0)
#ifndef PCL_EXPORTS
#define PCL_EXPORTS
#endif
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace pcl {

// ---- pcl/console/time.h ---------------------------------------------------------------------------
namespace console {
class TicToc {
 public:
  void tic() { t0_ = std::chrono::steady_clock::now(); }
  double toc() const {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count();
  }
  void toc_print() const { std::fprintf(stdout, "[done, %g ms]\n", toc()); }

 private:
  std::chrono::steady_clock::time_point t0_;
};
}  // namespace console

// ---- pcl/point_types.h ------------------------------------------------------------------------------
#define MINI_PCL_XYZ_MAPS                                                                         \
  Eigen::Map3<float> getVector3fMap() { return Eigen::Map3<float>(data); }                        \
  Eigen::Map3<const float> getVector3fMap() const { return Eigen::Map3<const float>(data); }      \
  Eigen::Map3<float, true> getArray3fMap() { return Eigen::Map3<float, true>(data); }             \
  Eigen::Map3<const float, true> getArray3fMap() const { return Eigen::Map3<const float, true>(data); }
#define MINI_PCL_NORMAL_MAPS                                                                      \
  Eigen::Map3<float> getNormalVector3fMap() { return Eigen::Map3<float>(data_n); }                \
  Eigen::Map3<const float> getNormalVector3fMap() const { return Eigen::Map3<const float>(data_n); }

struct alignas(16) PointXYZ {
  union {
    float data[4];
    struct {
      float x, y, z;
    };
  };
  PointXYZ() : PointXYZ(0.f, 0.f, 0.f) {}
  PointXYZ(float _x, float _y, float _z) {
    x = _x;
    y = _y;
    z = _z;
    data[3] = 1.0f;
  }
  MINI_PCL_XYZ_MAPS
};

struct alignas(16) PointXYZRGBA {
  union {
    float data[4];
    struct {
      float x, y, z;
    };
  };
  union {
    union {
      struct {
        std::uint8_t b, g, r, a;
      };
      float rgb;
    };
    std::uint32_t rgba;
  };
  std::uint32_t pad_[3];
  PointXYZRGBA() {
    x = y = z = 0.f;
    data[3] = 1.0f;
    r = g = b = 0;
    a = 255;
    pad_[0] = pad_[1] = pad_[2] = 0;
  }
  MINI_PCL_XYZ_MAPS
};

struct alignas(16) PointXYZRGB {
  union {
    float data[4];
    struct {
      float x, y, z;
    };
  };
  union {
    union {
      struct {
        std::uint8_t b, g, r, a;
      };
      float rgb;
    };
    std::uint32_t rgba;
  };
  std::uint32_t pad_[3];
  PointXYZRGB() {
    x = y = z = 0.f;
    data[3] = 1.0f;
    r = g = b = 0;
    a = 255;
    pad_[0] = pad_[1] = pad_[2] = 0;
  }
  MINI_PCL_XYZ_MAPS
};

struct alignas(16) Normal {
  union {
    float data_n[4];
    float normal[3];
    struct {
      float normal_x, normal_y, normal_z;
    };
  };
  union {
    struct {
      float curvature;
    };
    float data_c[4];
  };
  Normal() {
    normal_x = normal_y = normal_z = data_n[3] = 0.f;
    curvature = 0.f;
    data_c[1] = data_c[2] = data_c[3] = 0.f;
  }
  MINI_PCL_NORMAL_MAPS
};

struct alignas(16) PointNormal {
  union {
    float data[4];
    struct {
      float x, y, z;
    };
  };
  union {
    float data_n[4];
    float normal[3];
    struct {
      float normal_x, normal_y, normal_z;
    };
  };
  union {
    struct {
      float curvature;
    };
    float data_c[4];
  };
  PointNormal() {
    x = y = z = 0.f;
    data[3] = 1.0f;
    normal_x = normal_y = normal_z = data_n[3] = 0.f;
    curvature = 0.f;
    data_c[1] = data_c[2] = data_c[3] = 0.f;
  }
  MINI_PCL_XYZ_MAPS
  MINI_PCL_NORMAL_MAPS
};

struct alignas(16) PointXYZRGBNormal {
  union {
    float data[4];
    struct {
      float x, y, z;
    };
  };
  union {
    float data_n[4];
    float normal[3];
    struct {
      float normal_x, normal_y, normal_z;
    };
  };
  union {
    struct {
      union {
        union {
          struct {
            std::uint8_t b, g, r, a;
          };
          float rgb;
        };
        std::uint32_t rgba;
      };
      float curvature;
    };
    float data_c[4];
  };
  PointXYZRGBNormal() {
    x = y = z = 0.f;
    data[3] = 1.0f;
    normal_x = normal_y = normal_z = data_n[3] = 0.f;
    data_c[0] = data_c[1] = data_c[2] = data_c[3] = 0.f;
    r = g = b = 0;
    a = 255;
    curvature = 0.f;
  }
  MINI_PCL_XYZ_MAPS
  MINI_PCL_NORMAL_MAPS
};

struct Intensity {
  float intensity;
  Intensity() : intensity(0.f) {}
};

static_assert(sizeof(PointXYZ) == 16 && sizeof(PointXYZRGBA) == 32 && sizeof(PointXYZRGB) == 32 &&
                  sizeof(PointNormal) == 48 && sizeof(PointXYZRGBNormal) == 48 && sizeof(Normal) == 32,
              "PCL point layouts");

// ---- pcl/point_cloud.h ----------------------------------------------------------------------------
struct PCLHeader {
  std::uint32_t seq = 0;
  std::uint64_t stamp = 0;
  std::string frame_id;
};

template <typename PointT>
class PointCloud {
 public:
  typedef boost::shared_ptr<PointCloud<PointT> > Ptr;
  typedef boost::shared_ptr<const PointCloud<PointT> > ConstPtr;
  typedef PointT PointType;
  typedef std::vector<PointT, Eigen::aligned_allocator<PointT> > VectorType;

  PointCloud() : width(0), height(0), is_dense(true) {}
  PointCloud(std::uint32_t width_, std::uint32_t height_, const PointT &value_ = PointT())
      : points(width_ * height_, value_), width(width_), height(height_), is_dense(true) {}

  const PointT &operator()(std::size_t column, std::size_t row) const { return points[row * width + column]; }
  PointT &operator()(std::size_t column, std::size_t row) { return points[row * width + column]; }
  const PointT &at(int column, int row) const { return points.at(row * width + column); }
  PointT &at(int column, int row) { return points.at(row * width + column); }
  const PointT &at(std::size_t n) const { return points.at(n); }
  PointT &at(std::size_t n) { return points.at(n); }
  const PointT &operator[](std::size_t n) const { return points[n]; }
  PointT &operator[](std::size_t n) { return points[n]; }
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  bool isOrganized() const { return height > 1; }
  void resize(std::size_t n) {
    points.resize(n);
    if (width * height != n) {
      width = static_cast<std::uint32_t>(n);
      height = 1;
    }
  }
  void push_back(const PointT &pt) {
    points.push_back(pt);
    width = static_cast<std::uint32_t>(points.size());
    height = 1;
  }
  void clear() {
    points.clear();
    width = height = 0;
  }
  typename VectorType::iterator begin() { return points.begin(); }
  typename VectorType::iterator end() { return points.end(); }
  typename VectorType::const_iterator begin() const { return points.begin(); }
  typename VectorType::const_iterator end() const { return points.end(); }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
  PointCloud &operator+=(const PointCloud &rhs) {  // concatenation: the result is unorganised
    points.insert(points.end(), rhs.points.begin(), rhs.points.end());
    width = static_cast<std::uint32_t>(points.size());
    height = 1;
    is_dense = is_dense && rhs.is_dense;
    return *this;
  }

  PCLHeader header;
  VectorType points;
  std::uint32_t width;
  std::uint32_t height;
  bool is_dense;
};

// ---- pcl/PCLPointCloud2.h, pcl/PolygonMesh.h, pcl/Vertices.h ------------------------------------------
struct PCLPointField {
  std::string name;
  std::uint32_t offset = 0;
  std::uint8_t datatype = 0;
  std::uint32_t count = 0;
  enum { INT8 = 1, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 };
};

struct PCLPointCloud2 {
  PCLHeader header;
  std::uint32_t height = 0, width = 0;
  std::vector<PCLPointField> fields;
  std::uint8_t is_bigendian = 0;
  std::uint32_t point_step = 0, row_step = 0;
  std::vector<std::uint8_t> data;
  std::uint8_t is_dense = 0;
};

struct Vertices {
  std::vector<std::uint32_t> vertices;
};

struct PolygonMesh {
  typedef boost::shared_ptr<PolygonMesh> Ptr;
  typedef boost::shared_ptr<const PolygonMesh> ConstPtr;
  PCLHeader header;
  PCLPointCloud2 cloud;
  std::vector<Vertices> polygons;
};

namespace detail {
inline void add_field(std::vector<PCLPointField> &f, const char *name, std::uint32_t off, std::uint8_t type) {
  PCLPointField p;
  p.name = name;
  p.offset = off;
  p.datatype = type;
  p.count = 1;
  f.push_back(p);
}
inline void fields_of(const PointXYZ *, std::vector<PCLPointField> &f) {
  add_field(f, "x", 0, PCLPointField::FLOAT32);
  add_field(f, "y", 4, PCLPointField::FLOAT32);
  add_field(f, "z", 8, PCLPointField::FLOAT32);
}
inline void fields_of(const PointXYZRGB *, std::vector<PCLPointField> &f) {
  fields_of((const PointXYZ *)nullptr, f);
  add_field(f, "rgb", 16, PCLPointField::FLOAT32);
}
inline void fields_of(const PointXYZRGBA *, std::vector<PCLPointField> &f) {
  fields_of((const PointXYZ *)nullptr, f);
  add_field(f, "rgba", 16, PCLPointField::UINT32);
}
}  // namespace detail

// pcl/conversions.h: the blob is the raw point array, point_step = sizeof(PointT)
template <typename PointT>
void toPCLPointCloud2(const PointCloud<PointT> &cloud, PCLPointCloud2 &msg) {
  if (cloud.width == 0 && cloud.height == 0) {
    msg.width = static_cast<std::uint32_t>(cloud.points.size());
    msg.height = 1;
  } else {
    msg.height = cloud.height;
    msg.width = cloud.width;
  }
  const std::size_t bytes = sizeof(PointT) * cloud.points.size();
  msg.data.resize(bytes);
  if (bytes) std::memcpy(msg.data.data(), cloud.points.data(), bytes);
  msg.fields.clear();
  detail::fields_of((const PointT *)nullptr, msg.fields);
  msg.header = cloud.header;
  msg.point_step = sizeof(PointT);
  msg.row_step = static_cast<std::uint32_t>(sizeof(PointT) * msg.width);
  msg.is_dense = cloud.is_dense;
}

template <typename PointT>
void fromPCLPointCloud2(const PCLPointCloud2 &msg, PointCloud<PointT> &cloud) {
  cloud.header = msg.header;
  cloud.width = msg.width;
  cloud.height = msg.height;
  cloud.is_dense = msg.is_dense != 0;
  const std::size_t n = (std::size_t)msg.width * msg.height;
  cloud.points.assign(n, PointT());
  if (msg.data.empty()) return;
  std::vector<PCLPointField> want;
  detail::fields_of((const PointT *)nullptr, want);
  // fields are matched by name ("rgb" and "rgba" are the same 4 bytes); anything the target lacks is dropped
  for (const PCLPointField &w : want)
    for (const PCLPointField &f : msg.fields) {
      const bool colour = (w.name == "rgb" || w.name == "rgba") && (f.name == "rgb" || f.name == "rgba");
      if (f.name != w.name && !colour) continue;
      for (std::size_t i = 0; i < n; ++i)
        std::memcpy(reinterpret_cast<unsigned char *>(&cloud.points[i]) + w.offset, msg.data.data() + i * msg.point_step + f.offset, 4);
    }
}

template <typename PointT>
void copyPointCloud(const PointCloud<PointT> &in, PointCloud<PointT> &out) {
  out = in;
}

// ---- pcl/common/transforms.h  [PCL-recall: detail::Transformer, SSE2 nesting] ---------------------------
namespace detail {
template <typename Scalar>
struct Transformer {
  const Eigen::Matrix<Scalar, 4, 4> &tf;
  explicit Transformer(const Eigen::Matrix<Scalar, 4, 4> &t) : tf(t) {}
  void se3(const float *src, float *tgt) const {
    const Scalar x = src[0], y = src[1], z = src[2];
    for (int r = 0; r < 3; ++r) tgt[r] = static_cast<float>(x * tf(r, 0) + (y * tf(r, 1) + (z * tf(r, 2) + tf(r, 3))));
    tgt[3] = 1.0f;
  }
  void so3(const float *src, float *tgt) const {
    const Scalar x = src[0], y = src[1], z = src[2];
    for (int r = 0; r < 3; ++r) tgt[r] = static_cast<float>(x * tf(r, 0) + (y * tf(r, 1) + z * tf(r, 2)));
    tgt[3] = 0.0f;
  }
};
inline bool finite3(const float *p) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }
}  // namespace detail

template <typename PointT, typename Scalar>
inline PointT transformPoint(const PointT &point, const Eigen::Affine3<Scalar> &transform) {
  PointT ret = point;
  float out[4];
  detail::Transformer<Scalar>(transform.matrix()).se3(point.data, out);
  ret.x = out[0];
  ret.y = out[1];
  ret.z = out[2];
  return ret;
}

template <typename PointT, typename Scalar>
void transformPointCloud(const PointCloud<PointT> &cloud_in, PointCloud<PointT> &cloud_out,
                         const Eigen::Affine3<Scalar> &transform, bool copy_all_fields = true) {
  (void)copy_all_fields;
  if (&cloud_in != &cloud_out) cloud_out = cloud_in;
  detail::Transformer<Scalar> tf(transform.matrix());
  for (std::size_t i = 0; i < cloud_out.points.size(); ++i) {
    if (!cloud_in.is_dense && !detail::finite3(cloud_in.points[i].data)) continue;
    float out[4];
    tf.se3(cloud_in.points[i].data, out);
    cloud_out.points[i].x = out[0];
    cloud_out.points[i].y = out[1];
    cloud_out.points[i].z = out[2];
  }
}

template <typename PointT, typename Scalar>
void transformPointCloudWithNormals(const PointCloud<PointT> &cloud_in, PointCloud<PointT> &cloud_out,
                                    const Eigen::Affine3<Scalar> &transform, bool copy_all_fields = true) {
  (void)copy_all_fields;
  if (&cloud_in != &cloud_out) cloud_out = cloud_in;
  detail::Transformer<Scalar> tf(transform.matrix());
  for (std::size_t i = 0; i < cloud_out.points.size(); ++i) {
    if (!cloud_in.is_dense && !detail::finite3(cloud_in.points[i].data)) continue;
    float out[4], nout[4];
    tf.se3(cloud_in.points[i].data, out);
    tf.so3(cloud_in.points[i].data_n, nout);
    cloud_out.points[i].x = out[0];
    cloud_out.points[i].y = out[1];
    cloud_out.points[i].z = out[2];
    cloud_out.points[i].normal_x = nout[0];
    cloud_out.points[i].normal_y = nout[1];
    cloud_out.points[i].normal_z = nout[2];
  }
}

template <typename PointT>
void getMinMax3D(const PointCloud<PointT> &cloud, PointT &min_pt, PointT &max_pt) {
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
                 std::numeric_limits<float>::max()};
  float mx[3] = {-mn[0], -mn[1], -mn[2]};
  for (const PointT &p : cloud.points) {
    if (!cloud.is_dense && !detail::finite3(p.data)) continue;
    for (int k = 0; k < 3; ++k) {
      mn[k] = p.data[k] < mn[k] ? p.data[k] : mn[k];
      mx[k] = p.data[k] > mx[k] ? p.data[k] : mx[k];
    }
  }
  min_pt.x = mn[0];
  min_pt.y = mn[1];
  min_pt.z = mn[2];
  max_pt.x = mx[0];
  max_pt.y = mx[1];
  max_pt.z = mx[2];
}

// ---- pcl/filters/frustum_culling.h  [PCL-recall: FrustumCulling::applyFilter] ----------------------------
template <typename PointT>
class FrustumCulling {
 public:
  typedef typename PointCloud<PointT>::ConstPtr PointCloudConstPtr;
  explicit FrustumCulling(bool extract_removed_indices = false)
      : camera_pose_(Eigen::Matrix4f::Identity()), hfov_(60.f), vfov_(60.f), np_dist_(0.1f), fp_dist_(5.f) {
    (void)extract_removed_indices;
  }
  void setCameraPose(const Eigen::Matrix4f &camera_pose) { camera_pose_ = camera_pose; }
  void setHorizontalFOV(float hfov) { hfov_ = hfov; }
  void setVerticalFOV(float vfov) { vfov_ = vfov; }
  void setNearPlaneDistance(float np_dist) { np_dist_ = np_dist; }
  void setFarPlaneDistance(float fp_dist) { fp_dist_ = fp_dist; }
  void setInputCloud(const PointCloudConstPtr &cloud) { input_ = cloud; }

  void filter(std::vector<int> &indices) {
    using Eigen::Vector3f;
    using Eigen::Vector4f;
    indices.clear();
    const Vector3f view = camera_pose_.block<3, 1>(0, 0);   // view vector: first column
    const Vector3f up = camera_pose_.block<3, 1>(0, 1);     // up vector: second column
    const Vector3f right = camera_pose_.block<3, 1>(0, 2);  // right vector: third column
    const Vector3f T = camera_pose_.block<3, 1>(0, 3);      // camera position
    const float vfov_rad = float(vfov_ * M_PI / 180);
    const float hfov_rad = float(hfov_ * M_PI / 180);
    const float np_h = float(2 * tan(vfov_rad / 2) * np_dist_);
    const float np_w = float(2 * tan(hfov_rad / 2) * np_dist_);
    const float fp_h = float(2 * tan(vfov_rad / 2) * fp_dist_);
    const float fp_w = float(2 * tan(hfov_rad / 2) * fp_dist_);
    const Vector3f fp_c(T + view * fp_dist_);
    const Vector3f fp_tl(fp_c + (up * fp_h / 2) - (right * fp_w / 2));
    const Vector3f fp_tr(fp_c + (up * fp_h / 2) + (right * fp_w / 2));
    const Vector3f fp_bl(fp_c - (up * fp_h / 2) - (right * fp_w / 2));
    const Vector3f fp_br(fp_c - (up * fp_h / 2) + (right * fp_w / 2));
    const Vector3f np_c(T + view * np_dist_);
    const Vector3f np_tr(np_c + (up * np_h / 2) + (right * np_w / 2));
    const Vector3f np_bl(np_c - (up * np_h / 2) - (right * np_w / 2));
    const Vector3f np_br(np_c - (up * np_h / 2) + (right * np_w / 2));
    auto plane = [](const Vector3f &n, const Vector3f &through) { return Vector4f(n[0], n[1], n[2], -through.dot(n)); };
    const Vector4f pl_f = plane((fp_bl - fp_br).cross(fp_tr - fp_br), fp_c);
    const Vector4f pl_n = plane((np_tr - np_br).cross(np_bl - np_br), np_c);
    const Vector3f a(fp_bl - T), b(fp_br - T), c(fp_tr - T), d(fp_tl - T);
    const Vector4f pl_r = plane(b.cross(c), T);
    const Vector4f pl_l = plane(d.cross(a), T);
    const Vector4f pl_t = plane(c.cross(d), T);
    const Vector4f pl_b = plane(a.cross(b), T);
    for (std::size_t i = 0; i < input_->points.size(); ++i) {
      const PointT &p = input_->points[i];
      const Vector4f pt(p.x, p.y, p.z, 1.0f);
      const bool is_in_fov = (pt.dot(pl_l) <= 0) && (pt.dot(pl_r) <= 0) && (pt.dot(pl_t) <= 0) &&
                             (pt.dot(pl_b) <= 0) && (pt.dot(pl_f) <= 0) && (pt.dot(pl_n) <= 0);
      if (is_in_fov) indices.push_back(static_cast<int>(i));
    }
  }

 private:
  PointCloudConstPtr input_;
  Eigen::Matrix4f camera_pose_;
  float hfov_, vfov_, np_dist_, fp_dist_;
};

// ---- pcl/surface/marching_cubes.h  [PCL-recall] -----------------------------------------------------------
#include "mini_pcl_mc_tables.inc"

template <typename PointNT>
class MarchingCubes {
 public:
  typedef typename PointCloud<PointNT>::Ptr PointCloudPtr;
  typedef typename PointCloud<PointNT>::ConstPtr PointCloudConstPtr;

  MarchingCubes(const float percentage_extend_grid = 0.0f, const float iso_level = 0.0f)
      : res_x_(32), res_y_(32), res_z_(32), percentage_extend_grid_(percentage_extend_grid), iso_level_(iso_level) {}
  virtual ~MarchingCubes() {}

  void setIsoLevel(float iso_level) { iso_level_ = iso_level; }
  float getIsoLevel() { return iso_level_; }
  void setGridResolution(int res_x, int res_y, int res_z) {
    res_x_ = res_x;
    res_y_ = res_y;
    res_z_ = res_z;
  }
  void getGridResolution(int &res_x, int &res_y, int &res_z) {
    res_x = res_x_;
    res_y = res_y_;
    res_z = res_z_;
  }
  void setPercentageExtendGrid(float percentage) { percentage_extend_grid_ = percentage; }
  float getPercentageExtendGrid() { return percentage_extend_grid_; }
  void setInputCloud(const PointCloudConstPtr &cloud) { input_ = cloud; }

  // pcl::SurfaceReconstruction::reconstruct(PolygonMesh&): initCompute; polygons.clear();
  // performReconstruction; deinitCompute
  void reconstruct(PolygonMesh &output) {
    output.header = input_ ? input_->header : PCLHeader();
    output.polygons.clear();
    performReconstruction(output);
  }
  // pcl::SurfaceReconstruction::reconstruct(PointCloud&, std::vector<Vertices>&) [PCL-recall]: the same with
  // the second performReconstruction overload
  void reconstruct(PointCloud<PointNT> &points, std::vector<Vertices> &polygons) {
    points.header = input_ ? input_->header : PCLHeader();
    polygons.clear();
    performReconstruction(points, polygons);
  }

 protected:
  virtual void voxelizeData() = 0;
  virtual void performReconstruction(PolygonMesh &output) = 0;
  // upstream pcl::MarchingCubes implements this one itself (voxelizeData + createSurface over grid_); the stand-in
  // has no grid, and nothing in the reference calls it
  virtual void performReconstruction(PointCloud<PointNT> &points, std::vector<Vertices> &polygons) {
    points.clear();
    polygons.clear();
  }

  void getBoundingBox() {
    PointNT max_pt, min_pt;
    getMinMax3D(*input_, min_pt, max_pt);
    lower_boundary_ = min_pt.getArray3fMap();
    upper_boundary_ = max_pt.getArray3fMap();
    const Eigen::Array3f size3_extend = 0.5f * percentage_extend_grid_ * (upper_boundary_ - lower_boundary_);
    lower_boundary_ -= size3_extend;
    upper_boundary_ += size3_extend;
  }

  void interpolateEdge(Eigen::Vector3f &p1, Eigen::Vector3f &p2, float val_p1, float val_p2,
                       Eigen::Vector3f &output) {
    const float mu = (iso_level_ - val_p1) / (val_p2 - val_p1);
    output = p1 + mu * (p2 - p1);
  }

  void createSurface(const std::vector<float> &leaf_node, const Eigen::Vector3i &index_3d,
                     PointCloud<PointNT> &cloud) {
    int cubeindex = 0;
    for (int k = 0; k < 8; ++k)
      if (leaf_node[k] < iso_level_) cubeindex |= 1 << k;
    if (mc_detail::edgeTable[cubeindex] == 0) return;
    const Eigen::Vector3f center = lower_boundary_.matrix() + (size_voxel_ * index_3d.cast<float>().array()).matrix();
    std::vector<Eigen::Vector3f, Eigen::aligned_allocator<Eigen::Vector3f> > p;
    p.resize(8);
    for (int i = 0; i < 8; ++i) {
      Eigen::Vector3f point = center;
      if (i & 0x4) point[1] = static_cast<float>(center[1] + size_voxel_[1]);
      if (i & 0x2) point[2] = static_cast<float>(center[2] + size_voxel_[2]);
      if ((i & 0x1) ^ ((i >> 1) & 0x1)) point[0] = static_cast<float>(center[0] + size_voxel_[0]);
      p[i] = point;
    }
    static const int ev[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                  {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    std::vector<Eigen::Vector3f, Eigen::aligned_allocator<Eigen::Vector3f> > vertex_list;
    vertex_list.resize(12);
    for (int e = 0; e < 12; ++e)
      if (mc_detail::edgeTable[cubeindex] & (1 << e))
        interpolateEdge(p[ev[e][0]], p[ev[e][1]], leaf_node[ev[e][0]], leaf_node[ev[e][1]], vertex_list[e]);
    for (int i = 0; mc_detail::triTable[cubeindex][i] != -1; i += 3) {
      PointNT p1, p2, p3;
      p1.getVector3fMap() = vertex_list[mc_detail::triTable[cubeindex][i]];
      cloud.push_back(p1);
      p2.getVector3fMap() = vertex_list[mc_detail::triTable[cubeindex][i + 1]];
      cloud.push_back(p2);
      p3.getVector3fMap() = vertex_list[mc_detail::triTable[cubeindex][i + 2]];
      cloud.push_back(p3);
    }
  }

  PointCloudConstPtr input_;
  std::vector<float> grid_;
  int res_x_, res_y_, res_z_;
  Eigen::Array3f upper_boundary_, lower_boundary_, size_voxel_;
  float percentage_extend_grid_;
  float iso_level_;
};

}  // namespace pcl
