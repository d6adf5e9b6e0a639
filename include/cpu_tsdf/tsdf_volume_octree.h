// cpu_tsdf::TSDFVolumeOctree -- MI355X drop-in for the reference class of the same name
// (include/cpu_tsdf/tsdf_volume_octree.h:49-377).  Public signatures are kept; the octree of
// shared_ptr voxels is gone: voxels live in a flat SoA grid in HBM behind the C ABI of tsdf_hip.h, and
// every heavy method forwards to a HIP kernel.  What is intentionally absent: the public `octree_`
// member and getFrustumCulledVoxels (they expose OctreeNode pointers).  setColorMode takes "RGB", "RGBNormalized"
// and "LAB" like the reference (LAB: L, A, B state bit-identical, displayed bytes within 1 -- see tsdf_hip.h).
//
// Host-side arithmetic that decides results is done here with the caller's own Eigen/PCL, exactly where
// the reference does it: trans.inverse().cast<float>() (hpp:54), trans.rotation().cast<float>() /
// translation (:303-304), transformPointCloudWithNormals (:422), transformPointCloud in the mesher.
#pragma once

#include <cpu_tsdf/tsdf_interface.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <Eigen/Geometry>
#include <tsdf_hip.h>

#include <string>
#include <vector>

namespace cpu_tsdf {

class TSDFVolumeOctree : public TSDFInterface {
 public:
  typedef boost::shared_ptr<TSDFVolumeOctree> Ptr;
  typedef boost::shared_ptr<const TSDFVolumeOctree> ConstPtr;

  TSDFVolumeOctree();
  ~TSDFVolumeOctree();
  TSDFVolumeOctree(const TSDFVolumeOctree &) = delete;
  TSDFVolumeOctree &operator=(const TSDFVolumeOctree &) = delete;

  void setResolution(int xres, int yres, int zres);
  void getResolution(int &xres, int &yres, int &zres) const;
  void setGridSize(float xsize, float ysize, float zsize);
  void getGridSize(float &xsize, float &ysize, float &zsize) const;
  void setImageSize(int width, int height);
  void getImageSize(int &width, int &height) const;
  void setDepthTruncationLimits(float max_dist_pos, float max_dist_neg);
  void getDepthTruncationLimits(float &max_dist_pos, float &max_dist_neg) const;
  void setWeightTruncationLimit(float max_weight);
  float getWeightTruncationLimit() const;
  void setGlobalTransform(const Eigen::Affine3d &trans) { global_transform_ = trans; }
  Eigen::Affine3d getGlobalTransform() const { return global_transform_; }
  void setCameraIntrinsics(const double focal_length_x, const double focal_length_y,
                           const double principal_point_x, const double principal_point_y);
  void getCameraIntrinsics(double &focal_length_x, double &focal_length_y, double &principal_point_x,
                           double &principal_point_y) const;
  // Accepted and remembered (save() writes them), but a dense grid has no coarse cells / pre-split pass.
  void setMaxVoxelSize(float x, float y, float z);
  void setNumRandomSplts(int n) { num_random_splits_ = n; }
  int getNumRandomSplits() { return num_random_splits_; }
  void setIntegrateColor(bool integrate_color);
  void setColorMode(const std::string &color_mode);
  void setSensorDistanceBounds(float min_sensor_dist, float max_sensor_dist);
  void getSensorDistanceBounds(float &min_sensor_dist, float &max_sensor_dist) const;

  // (Re)allocates the grid on the GPU; every voxel (d = -1, w = 0).
  void reset();
  void save(const std::string &filename) const;
  void load(const std::string &filename);

  bool getFxn(const pcl::PointXYZ &pt, float &val) const;
  bool getGradient(const pcl::PointXYZ &pt, Eigen::Vector3f &grad) const;
  bool getHessian(const pcl::PointXYZ &pt, Eigen::Matrix3f &hessian) const;
  bool getFxnAndGradient(const pcl::PointXYZ &pt, float &val, Eigen::Vector3f &grad) const;
  bool getFxnGradientAndHessian(const pcl::PointXYZ &pt, float &val, Eigen::Vector3f &grad,
                                Eigen::Matrix3f &hessian) const;

  // `cloud` must be organised with the configured image size; only pt.z (and r,g,b when colour is on) is
  // read, `normals` is unused (as in the reference's default settings).  trans: camera -> volume.
  template <typename PointT, typename NormalT>
  bool integrateCloud(const pcl::PointCloud<PointT> &cloud, const pcl::PointCloud<NormalT> &normals,
                      const Eigen::Affine3d &trans = Eigen::Affine3d::Identity());

  pcl::PointCloud<pcl::PointNormal>::Ptr renderView(const Eigen::Affine3d &trans = Eigen::Affine3d::Identity(),
                                                    int downsampleBy = 1) const;
  pcl::PointCloud<pcl::PointXYZRGBNormal>::Ptr renderColoredView(
      const Eigen::Affine3d &trans = Eigen::Affine3d::Identity(), int downsampleBy = 1) const;
  pcl::PointCloud<pcl::Intensity>::Ptr getIntensityCloud(const Eigen::Affine3d &trans = Eigen::Affine3d::Identity()) const;

  pcl::PointXYZ getVoxelCenter(size_t x, size_t y, size_t z) const;
  bool getVoxelIndex(float x, float y, float z, int &x_i, int &y_i, int &z_i) const;
  pcl::PointCloud<pcl::PointXYZ>::ConstPtr getVoxelCenters(int nlevels = 4) const;
  void getOccupiedVoxelIndices(std::vector<Eigen::Vector3i> &indices) const;
  bool isEmpty() const { return is_empty_; }

  // ---- extensions (not in the reference) -----------------------------------------------------------------
  // Planar entry point used by the integrateCloud template: depth H x W floats (NaN = no return), bgra
  // 4 bytes per pixel in PointXYZRGBA byte order or NULL.
  bool integratePlanar(const float *depth, const unsigned char *bgra, int width, int height,
                       const Eigen::Affine3d &trans);
  // The two halves the integrateCloud template is made of: beginFrame hands out the pinned staging slot of the next
  // frame (*depth: height x width floats; *bgra: 4 bytes per pixel, or NULL when colour is off), commitFrame queues its
  // upload and the integrate launch and returns (pipelined: see impl/tsdf_volume_octree.hpp).
  bool beginFrame(int width, int height, float **depth, unsigned char **bgra);
  bool commitFrame(const Eigen::Affine3d &trans);
  // The `integrate` program's per-cloud preparation + integrateCloud in one call (src/prog/integrate.cpp:
  // 559-618, 650, 673): an UNORGANISED cloud in sensor units is scaled by cloud_units, (0,0,0) becomes NaN if
  // zero_nans, it is moved by *world_to_cam (= poses[i].inverse()) when given, z-buffered into the configured
  // image on the GPU (tsdf_hip_organize) and integrated with pose `trans`.
  bool integrateUnorganized(const pcl::PointCloud<pcl::PointXYZRGBA> &cloud, const Eigen::Affine3d &trans,
                            float cloud_units = 1.f, bool zero_nans = false,
                            const Eigen::Affine3d *world_to_cam = nullptr, size_t *n_valid_pixels = nullptr);
  // Raw voxel block readback ([z][y][x]); any pointer may be NULL; rgb is 3 bytes per voxel.
  bool downloadBlock(int x0, int y0, int z0, int nx, int ny, int nz, float *d, float *w, unsigned char *rgb) const;
  // The C-ABI handle (NULL before reset()); used by MarchingCubesTSDFOctree.
  tsdf_handle handle() const { return h_; }
  void setTransformOrder(int order) { p_.xform_order = order; }
  void setDevice(int device) { p_.device = device; }
  // Spread the volume over several GPUs of this node: contiguous Z-slabs, one per entry (tsdf_hip_create_multi);
  // takes effect at the next reset() / load().  Every method of this class and MarchingCubesTSDFOctree then drives
  // all of them from this one process: the frame of integrateCloud fans out to every GPU, each integrates its own
  // planes, reconstruct / renderView / getFxn exchange halo planes and ray records between them.  Results do not
  // depend on the partition.  An empty list = one GPU (setDevice).
  void setDevices(const std::vector<int> &devices) { devices_ = devices; }
  const std::vector<int> &getDevices() const { return devices_; }
  // The reference's integrateCloud only visits the voxels pcl::FrustumCulling keeps (getFrustumCulledVoxels,
  // src/lib/tsdf_volume_octree.cpp:619-652): 1.1 x the field of view around the optical AXIS between the sensor-range
  // planes.  integrateCloud here does the same: the six planes are built with the caller's Eigen, operation by operation as
  // pcl::FrustumCulling::applyFilter builds them, and handed to the library per frame (tsdf_hip_set_reference_cull), which
  // applies them wherever they can decide a voxel (for an ordinary camera looking at a volume inside its sensor range:
  // nowhere, at no cost).  setReferenceCull(false) opts out: every voxel updateVoxel itself accepts is integrated.
  void setReferenceCull(bool flag) { reference_cull_ = flag; }
  // Not in the reference: integrateCloud calls are integrated two per sweep of the volume where both camera poses see
  // the whole grid (tsdf_hip_set_frame_pairing: a cloud's kernel waits for the next integrateCloud -- or for any other
  // method of this class, which launches it on its own first).  The same voxels, bit for bit; ~1.15x the frames per
  // second on a stream of clouds.  Takes effect at once and survives reset(); on a setDevices set every slab pairs by itself.
  void setFramePairing(bool flag) {
    frame_pairing_ = flag;
    if (h_) (void)tsdf_hip_set_frame_pairing(h_, flag ? 1 : 0);
  }
  bool getFramePairing() const { return frame_pairing_; }
  // integrateCloud returns as soon as the cloud is staged and its upload + kernel are queued (the reference returns after
  // the work, always `true`; here `false` means the call could not be queued).  A device error of the queued work surfaces
  // at the next call that waits for the device (renderView, getFxn, save, reconstruct, ...).  setSynchronous(true) makes
  // every integrateCloud wait for its own kernel and report its status itself -- the reference's timing, at the price of
  // the overlap (measured: 59.8 against 60.5 frames/s at 2048^3, where the kernel dwarfs the upload; 2x at 512^3).
  void setSynchronous(bool flag) { synchronous_ = flag; }
  bool getSynchronous() const { return synchronous_; }
  bool getReferenceCull() const { return reference_cull_; }
  // TSDF_LAYOUT_* (include/tsdf_hip.h): how the weight is stored in HBM; default AUTO
  void setLayout(int layout) { p_.layout = layout; }

  const float UNOBSERVED_VOXEL;

 private:
  bool ready(const char *who) const;
 public:
  // false (and a PCL_ERROR) if setGridSize was not a cube: renderView / getFxn... / MarchingCubesTSDFOctree refuse then
  // (the reference mixes two geometries there; see the definition)
  bool cubicForQueries(const char *who) const;
 private:
  tsdf_params p_;
  tsdf_handle h_;
  float max_cell_size_[3];
  int num_random_splits_;
  bool is_empty_, weight_by_depth_, weight_by_variance_;
  std::string color_mode_;
  std::vector<int> devices_;
  bool reference_cull_ = true;
  bool frame_pairing_ = false;
  bool synchronous_ = false;
  mutable bool cull_planes_set_ = false;
  bool applyReferenceCull(const Eigen::Affine3d &trans) const;
  // pinned staging for renderView's readback (tsdf_hip_host_alloc): the GPU writes it by DMA, the conversion into the
  // returned PointCloud reads it -- no intermediate copy
  mutable float *view_buf_;
  mutable size_t view_cap_;
  Eigen::Affine3d global_transform_;

 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
};

}  // namespace cpu_tsdf

#include <cpu_tsdf/impl/tsdf_volume_octree.hpp>
