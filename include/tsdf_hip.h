/*
 * tsdf_hip.h -- C ABI of the MI355X (gfx950) TSDF fusion library (libtsdf_hip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes only, no C++ / torch types.
 * The C++ host shell (the headers under include/cpu_tsdf/, same class names and signatures as the
 * reference) and the Python binding (cpu_tsdf_amd/capi.py) both sit on top of it.
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * sdmiller/cpu_tsdf tree).  All functions return 0 on success or a TSDF_HIP_E_* code;
 * the reference itself has no error reporting, the host shell maps
 * failures to `false` / empty results.
 *
 * Volume storage: flat SoA planes in HBM, x fastest:
 *     d  [nz_local][ny][pitch]  float   truncation-normalised distance, init -1
 *     w  [nz_local][ny][pitch]  float   weight, init 0                     (F32W layout)
 *     rgb[nz_local][ny][pitch]  uint32  r | g<<8 | b<<16 (only when integrate_color); in the PACKED layout
 *                                       byte 3 holds the observation count k, w = min(k, max_weight)
 *     k  [nz_local][ny][pitch]  uint8   the same count when PACKED without colour
 * pitch = nx rounded up to a multiple of 4.  A handle may own only a Z-slab
 * [z_begin, z_end) of the full grid (multi-GPU partitioning) plus `halo` extra planes on
 * each side that integrate never touches but raycast / marching cubes may read.
 */
#ifndef TSDF_HIP_H
#define TSDF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tsdf_hip_volume *tsdf_handle;

enum {
  TSDF_HIP_OK = 0,
  TSDF_HIP_E_INVALID = 1,     /* bad argument / params */
  TSDF_HIP_E_NOMEM = 2,       /* hipMalloc failed */
  TSDF_HIP_E_HIP = 3,         /* any other HIP runtime error (see tsdf_hip_last_error) */
  TSDF_HIP_E_NODEVICE = 4,    /* no gfx950 device visible */
  TSDF_HIP_E_UNSUPPORTED = 5,
  TSDF_HIP_E_IO = 6           /* file missing, unreadable, unwritable or not a .vol (see tsdf_hip_last_error) */
};

/* Summation order of the rigid transform g = T * (c,1), which decides the last ulp of the
 * camera-frame voxel centre (include/cpu_tsdf/impl/tsdf_volume_octree.hpp:145 calls
 * pcl::transformPoint, whose arithmetic lives in PCL, not in the reference tree). */
enum {
  TSDF_XFORM_PCL_SSE = 0,     /* x*c0 + (y*c1 + (z*c2 + c3))   PCL >= 1.10 Transformer<float>::se3, SSE2 build */
  TSDF_XFORM_LEFT_TO_RIGHT = 1 /* ((m0*x + m1*y) + m2*z) + m3    PCL scalar build / Eigen Affine3f * Vector3f */
};

/* How the weight is stored.  Every observation adds w_new = 1 and clamps at max_weight
 * (src/lib/octree.cpp:156-158; the depth/variance weightings of hpp:200-204 have no setter), so
 * w == min(k, max_weight) where k counts observations.  PACKED stores k in one byte -- byte 3 of the
 * colour word, or a uint8 plane without colour -- and saves a third of the HBM traffic of integrateCloud;
 * every entry point still speaks float weights.  It needs 0 <= max_weight <= 255 and weights of that form:
 * tsdf_hip_upload refuses anything else with TSDF_HIP_E_UNSUPPORTED (load such a volume with F32W).
 * AUTO = PACKED when max_weight allows it, else F32W. */
enum { TSDF_LAYOUT_AUTO = 0, TSDF_LAYOUT_F32W = 1, TSDF_LAYOUT_PACKED = 2 };

/* Which voxel class carries the colour (setColorMode, include/cpu_tsdf/tsdf_volume_octree.h:290 ->
 * OctreeNode::instantiateByTypeString, src/lib/octree.cpp:193-206).  Only read when integrate_color is set.
 *   RGB             RGBNode (octree.cpp:328-337): three truncated uint8 running means.  The default.
 *   RGB_NORMALIZED  RGBNormalized (octree.cpp:380-402): running means of r/i, g/i, b/i and of the intensity
 *                   i = sqrt(r^2+g^2+b^2) as four floats; the colour read back is (uint8)(r_n * i) etc.
 *                   Four more float planes per voxel, float weights (no PACKED layout), a plain per-voxel
 *                   kernel; tsdf_hip_upload of rgb and save/load are refused (the reference's own
 *                   serialisation of this class writes one byte of each float, octree.cpp:417-433).
 *   LAB             LABNode (octree.cpp:531-551): float running means of the CIE L*a*b* image of each pixel
 *                   (RGB2LAB, octree.cpp:436-481); the colour read back is LAB2RGB of the means
 *                   (octree.cpp:483-527).  Three more float planes, float weights, a plain kernel, no rgb
 *                   upload and no save/load, like RGB_NORMALIZED.  Both conversions go through std::pow in
 *                   the reference.  Here the sRGB curve (a function of one byte) is tabulated by the host's
 *                   own libm at create, and the cube roots run on the device in fp64 before the same rounding
 *                   to float: the L, A, B state equals the reference's bit for bit for every one of the 2^24
 *                   pixel colours (the GPU tests sweep them all); the bytes LAB2RGB returns may differ by 1
 *                   where (float) * 255 lands within an ulp of an integer (tolerance stated in
 *                   tests/test_lab_gpu.py: +-1, > 99.9 % identical). */
enum {
  TSDF_COLOR_RGB = 0,
  TSDF_COLOR_RGB_NORMALIZED = 1,
  TSDF_COLOR_LAB = 2
};

/* Everything TSDFVolumeOctree's setters configure before reset()
 * (src/lib/tsdf_volume_octree.cpp:54-85 defaults, :92-199 setters). */
typedef struct tsdf_params {
  int32_t res[3];             /* setResolution            tsdf_volume_octree.cpp:92-98   */
  float size[3];              /* setGridSize              :110-116 (metres)              */
  float max_dist_pos;         /* setDepthTruncationLimits :146-151                       */
  float max_dist_neg;
  float max_weight;           /* setWeightTruncationLimit :163-167                       */
  float min_sensor_dist;      /* setSensorDistanceBounds  tsdf_volume_octree.h:185-190   */
  float max_sensor_dist;
  double fx, fy, cx, cy;      /* setCameraIntrinsics      :176-186 (double, as stored)   */
  int32_t image_width;        /* setImageSize             :128-133                       */
  int32_t image_height;
  int32_t integrate_color;    /* setIntegrateColor        tsdf_volume_octree.h:172-173   */
  int32_t xform_order;        /* TSDF_XFORM_*                                            */
  int32_t z_begin, z_end;     /* Z-slab owned by this handle; 0,0 => whole grid          */
  int32_t halo;               /* extra planes kept below z_begin and above z_end         */
  int32_t device;             /* HIP ordinal, -1 => current device                       */
  int32_t layout;             /* TSDF_LAYOUT_*                                           */
  int32_t color_mode;         /* TSDF_COLOR_*: setColorMode, tsdf_volume_octree.h:290    */
} tsdf_params;

/* Fill *p with the reference constructor defaults (tsdf_volume_octree.cpp:54-85). */
void tsdf_hip_default_params(tsdf_params *p);

/* reset() -- tsdf_volume_octree.cpp:201-219: allocate the grid, every voxel (d=-1, w=0). */
int tsdf_hip_create(const tsdf_params *p, tsdf_handle *out);
int tsdf_hip_reset(tsdf_handle h);
int tsdf_hip_destroy(tsdf_handle h);

/* One volume over several GPUs of ONE node, in one process, behind the same handle type -- the drop-in form of the
 * Z-slab partition: the handle owns n_devices Z-slab handles (contiguous, balanced ranges of z planes, one per entry
 * of `devices`, which may repeat an ordinal), each with tsdf_hip_render_halo(p) halo planes, and EVERY entry point
 * below that takes a handle works on it:
 *   integrate*      the frame fans out to every slab (from pinned host memory over each GPU's own PCIe link, or from
 *                   the GPU that holds it by peer copy over xGMI) and every slab integrates its own planes; no voxel
 *                   crosses a link.  n_observed is the sum over the slabs.
 *   march / fetch   one plane of halo per slab from its upper neighbour (peer copy), the slabs mesh concurrently, the
 *                   triangle lists are merged by the reference's order; the merged mesh lives on the host.
 *   raycast*        ray hand-off between the slabs on COMPACT lists: each slab keeps only the rays it is responsible
 *                   for and advances them on its own stream; a ray that needs another slab's voxel travels there
 *                   point-to-point (96 B), a finished ray goes to the first slab (36 B), which assembles the image
 *                   and applies the camera transform; tsdf_hip_multi_render_stats reports what moved.
 *   sample, lookup_rgb, download, upload, save, reset, synchronize, set_weighting, centers, layout: as for one handle.
 * Not available on such a handle (TSDF_HIP_E_UNSUPPORTED): set_stream, device_planes, get/set_planes_device,
 * raycast_begin/advance*, march_fetch_device -- they expose ONE device's memory.
 * p->z_begin / z_end must be 0 (the whole grid); p->device is ignored.  Every slab works on its own non-blocking
 * stream (also when ordinals repeat), every cross-slab dependency is an event.  Results are bit-identical to one
 * handle holding the whole grid -- tested with repeated ordinals on one GPU and, where more than one GPU is visible,
 * with distinct ordinals (tests/test_multi_gpu.py); the latter has not run on this project's one-GPU test boxes.  cpu_tsdf::TSDFVolumeOctree::setDevices() is the C++ face of it;
 * cpu_tsdf_amd/zslab.py is the multi-PROCESS form of the same partition (one rank per GPU, RCCL). */
int tsdf_hip_create_multi(const tsdf_params *p, const int32_t *devices, int n_devices, tsdf_handle *out);
/* Number of Z-slab handles behind h (1 for an ordinary handle) and the device / owned planes / halo of slab k. */
int tsdf_hip_slab_count(tsdf_handle h);
/* Report-only, multi-GPU handles: of the last tsdf_hip_raycast* call -- out[0] rounds, out[1] ray records handed from one
 * slab to another, out[2] bytes that moved between slabs (hand-offs + finished rays to the first slab), out[3] host
 * waits (one per slab and round: the slabs of a round run concurrently). */
int tsdf_hip_multi_render_stats(tsdf_handle h, uint64_t out[4]);
/* Report-only, multi-GPU handles: how the slabs reach each other.  Copies between two slabs' GPUs (the frame fan-out of
 * the device entry points, halo planes, ray records) are hipMemcpyPeerAsync where the driver grants peer access; a pair
 * it refuses -- said once on stderr at create -- goes through a pinned relay buffer on the host instead (device -> host
 * on the source GPU, host -> device on the receiver's stream, every step ordered by events: slower, never a hang).
 * TSDF_HIP_NO_PEER=1 in the environment at create routes EVERY cross-slab copy that way (how a one-GPU box tests it).
 * out[0] = refused device pairs, out[1] = 1 if everything is relayed, out[2] = bytes relayed since create. */
int tsdf_hip_multi_link_stats(tsdf_handle h, uint64_t out[3]);
/* Report-only, multi-GPU handles: per-slab k_integrate time.  While enabled, every slab's integrate launch is bracketed
 * by HIP events on that slab's stream; tsdf_hip_multi_kernel_ms synchronises slab k and returns the summed milliseconds
 * and the number of launches since timing was enabled (or last read). */
int tsdf_hip_multi_timing(tsdf_handle h, int enable);
int tsdf_hip_multi_kernel_ms(tsdf_handle h, int k, float *ms_sum, int32_t *launches);
int tsdf_hip_slab_info(tsdf_handle h, int k, int32_t *device, int32_t *z_begin, int32_t *z_end, int32_t *halo);

/* Host memory and the boundary.  Every entry point that takes or returns HOST arrays moves them through a pinned
 * two-slot bounce buffer owned by the handle (one extra host memcpy, the same cost whatever memory the caller hands
 * over) -- unless the caller's buffer is itself pinned or registered (hipHostMalloc, hipHostRegister, or
 * tsdf_hip_host_alloc below): that is detected (hipPointerGetAttributes) and the DMA engine reads / writes it
 * directly.  tsdf_hip_host_alloc gives callers without HIP of their own (the C++ shell, ctypes) such memory. */
int tsdf_hip_host_alloc(size_t bytes, void **out);
int tsdf_hip_host_free(void *p);

/* Work is queued on this hipStream_t (default: the null stream). */
int tsdf_hip_set_stream(tsdf_handle h, void *hip_stream);
int tsdf_hip_synchronize(tsdf_handle h);

/* integrateCloud -- include/cpu_tsdf/impl/tsdf_volume_octree.hpp:48-103 (+ updateVoxel :113-218).
 *   depth        image_height x image_width floats, row-major: pt.z of cloud(u,v); NaN = no return.
 *   bgra         same shape, 4 bytes/pixel in PCL PointXYZRGBA memory order (b,g,r,a); NULL unless
 *                integrate_color.
 *   cam_from_vol 3x4 row-major float: trans.inverse().cast<float>() exactly as hpp:54 computes it.
 *   n_observed   optional: number of voxels that reached addObservation this frame.
 * The host variant copies the frame to the device and synchronises; the device variant takes
 * device pointers, is asynchronous on the handle's stream and only synchronises when
 * n_observed != NULL.  The caller's device buffers must be complete when the call is made and stay untouched until
 * the work has finished (tsdf_hip_synchronize or any synchronising call): on a multi-GPU handle the other GPUs copy
 * the frame from them on their own streams. */
int tsdf_hip_integrate(tsdf_handle h, const float *depth, const uint8_t *bgra,
                       const float cam_from_vol[12], uint64_t *n_observed);
int tsdf_hip_integrate_device(tsdf_handle h, const float *d_depth, const uint32_t *d_bgra,
                              const float cam_from_vol[12], uint64_t *n_observed);
/* Pipelined form of tsdf_hip_integrate for a stream of host frames: the frame is copied into a pinned
 * two-slot ring and the call returns; the upload overlaps the previous frame's kernel, so frames flow at the
 * kernel's rate.  The caller's buffers may be reused as soon as the call returns.  Same results; ordered
 * before every later call on this handle; asynchronous errors surface at the next synchronising call. */
int tsdf_hip_integrate_async(tsdf_handle h, const float *depth, const uint8_t *bgra,
                             const float cam_from_vol[12]);
/* The two halves of tsdf_hip_integrate_async, for a caller that can WRITE its frame straight into the pinned slot
 * (the C++ integrateCloud template strips the z / b,g,r fields of a pcl::PointCloud into it, in parallel) instead of
 * building planar images first: _begin returns the slot's host pointers (*depth: image_height x image_width floats,
 * *bgra: as many 4-byte pixels, NULL without integrate_color), waiting only for the kernel that read the slot two frames
 * ago; _commit uploads and queues the integrate launch, as tsdf_hip_integrate_async does.  One begin per commit. */
int tsdf_hip_frame_begin(tsdf_handle h, float **depth, uint8_t **bgra);
int tsdf_hip_frame_commit(tsdf_handle h, const float cam_from_vol[12]);

/* The `integrate` program's per-cloud preparation -- src/prog/integrate.cpp:559-618 and reprojectPoint
 * :201-207: scale by cloud_units, optionally turn (0,0,0) into NaN, optionally move the cloud by
 * poses[i].inverse() (world_to_cam: the first three rows of that matrix, row-major doubles, computed by
 * the caller's Eigen; NULL = clouds are already in the camera frame), then z-buffer the points into an
 * image_height x image_width organised frame (smallest z wins, the earlier point among equals) using the
 * volume's intrinsics AS FLOATS, as the program does.
 *   xyz / bgra   n points: 3 floats every xyz_stride floats, 4 bytes (b,g,r,a) every bgra_stride bytes
 *                (a PCL PointXYZRGBA array is xyz_stride 8, bgra at byte 16 with bgra_stride 32); bgra may be NULL.
 *   depth_out / bgra_out / n_valid   optional host copies of the frame (NaN = empty pixel) and the number
 *                of filled pixels; with all three NULL the call is asynchronous.
 * The frame stays in the handle; tsdf_hip_integrate_staged integrates it (same as tsdf_hip_integrate_device
 * on it). */
int tsdf_hip_organize(tsdf_handle h, const float *xyz, size_t xyz_stride, const uint8_t *bgra,
                      size_t bgra_stride, size_t n, float cloud_units, int zero_nans,
                      const double world_to_cam[12], float *depth_out, uint8_t *bgra_out, uint64_t *n_valid);
int tsdf_hip_integrate_staged(tsdf_handle h, const float cam_from_vol[12], uint64_t *n_observed);

/* Of the last tsdf_hip_integrate* call on this handle that asked for n_observed: out[0] = that count, out[1] = bytes
 * of voxel words whose VALUE changed (4 per distance / weight / colour word, 1 per count byte) -- with the bytes read
 * per observed voxel this is the algorithmic traffic of the chosen HBM layout (bench.py's roofline). */
int tsdf_hip_last_count_detail(tsdf_handle h, uint64_t out[2]);
/* Of the same call: out[0] = observed voxels whose DISTANCE word was not read, because the kernel could tell it from the
 * voxel's observation count (PACKED layout: in a cell of 64 x 4 x 1 voxels that no frame since the reset has observed
 * inside the truncation band, an observed voxel sits at max_dist_pos / max_dist_neg and an unobserved one at the reset
 * value) -- 4 bytes each that the launch did not move; out[1] = 1 if the launch was allowed to do so;
 * out[2] = bytes of the voxel planes the launch REQUESTED (it asks for a quad's words together with the frame pixels, so
 * also for quads none of whose voxels turns out to be observed; 0 for the plain kernels, which do not count). */
int tsdf_hip_last_read_detail(tsdf_handle h, uint64_t out[3]);

/* The two observation weightings of updateVoxel -- include/cpu_tsdf/impl/tsdf_volume_octree.hpp:200-204.  The
 * reference has no setter for them: weight_by_depth_ / weight_by_variance_ only become true through load()
 * (src/lib/tsdf_volume_octree.cpp:265-266); tsdf_hip_load applies what the file says.
 *   weight_by_depth     w_new *= (1 - std::min(pt.z / 10., 1.)) (:201-202).  Weights are then no longer counts:
 *                       needs the F32W layout (E_UNSUPPORTED on a PACKED handle; tsdf_hip_load with layout AUTO
 *                       picks F32W by itself) and TSDF_COLOR_RGB.  Integrated by a plain per-voxel kernel (exact
 *                       fp64 projection, IEEE divisions), every operation in the reference's order.
 *   weight_by_variance  w_new *= exp(logNormal(d_new, d, variance)) once a voxel has more than 5 samples (:203-204),
 *                       from OctreeNode::M_ / nsample_ (src/lib/octree.cpp:160-161,281-287): two more planes per voxel
 *                       (float M, int32 nsample), allocated when the flag is set, zero like a fresh octree's, updated by
 *                       every observation from then on, carried by tsdf_hip_save / tsdf_hip_load in the node records the
 *                       reference keeps them in, and readable / writable through tsdf_hip_*_variance_state.  Needs
 *                       the F32W layout and TSDF_COLOR_RGB like weight_by_depth (tsdf_hip_load with AUTO picks F32W);
 *                       integrated by the same plain kernel.  std::exp(float) is the host libm's expf restated on the
 *                       device (glibc's table algorithm, in the FMA or the plain build's form, whichever this host
 *                       runs): equal on every float in +-(2^-26 .. 104), which tests/test_wvar_gpu.py sweeps. */
int tsdf_hip_set_weighting(tsdf_handle h, int weight_by_depth, int weight_by_variance);

/* OctreeNode::M_ / nsample_ of a block of voxels ([z][y][x]; either pointer may be NULL) -- include/cpu_tsdf/octree.h:
 * 164-165, the state weight_by_variance integrates with.  Only volumes that weight by variance keep it (E_INVALID
 * otherwise).  The reference exposes the members directly; tsdf_hip_save / tsdf_hip_load move them with the file. */
int tsdf_hip_download_variance_state(tsdf_handle h, int x0, int y0, int z0, int nx, int ny, int nz, float *M, int32_t *nsample);
int tsdf_hip_upload_variance_state(tsdf_handle h, int x0, int y0, int z0, int nx, int ny, int nz, const float *M,
                                   const int32_t *nsample);

/* renderView -- tsdf_volume_octree.cpp:278-421 (everything except the last line).
 *   rot        3x3 row-major float:  trans.rotation().cast<float>()      (:303)
 *   origin     3 floats:             trans.translation().cast<float>()   (:304)
 *              Both are computed by the caller with its own Eigen so the kernel sees exactly the
 *              numbers the reference would (Affine3d::rotation() runs an SVD inside Eigen).
 *   downsample downsampleBy
 *   out        (image_height/ds) x (image_width/ds) x 8 floats per pixel:
 *              x,y,z, nx,ny,nz, t_star, iterations -- in the VOLUME frame; the reference's final
 *              transformPointCloudWithNormals by trans^-1 (:422) is applied by the host shell.
 *              A miss has NaN x,y,z and a zero normal (PointNormal's default constructor). */
int tsdf_hip_raycast(tsdf_handle h, const float rot[9], const float origin[3], int downsample,
                     float *out);
/* The same with the reference's last line (:422) done in the kernel: cam_from_vol = the first three rows of
 * trans.inverse().matrix() (row-major doubles, from the caller's Eigen); points and normals come back in the
 * CAMERA frame, as renderView returns them.  For callers without PCL (the Python binding). */
int tsdf_hip_raycast_camera(tsdf_handle h, const float rot[9], const float origin[3], int downsample,
                            const double cam_from_vol[12], float *out);

/* renderView across Z-slab handles (one handle per GPU): ray hand-off.  The reference's ray loop
 * (tsdf_volume_octree.cpp:313-369) chooses each step from the voxel it last visited, so the loop state of a
 * ray travels between the slabs it crosses instead of the voxels.  One record of
 * TSDF_HIP_RAY_RECORD_INTS 32-bit words per ray, in DEVICE memory, row-major like the image:
 *   [0] status (1 suspended, 2 finished; 0 = "not touched" in a delta buffer)   [1] global z plane of the
 *   voxel the ray needs next, -1 before it needed any   [2] iterations   [3] hit_voxel   [4] t   [5..7] pt
 *   [8] last_d   [9] last_w   [10] step   [11] the ray's pixel index   [12] finish flag (the march is done, [4]
 *   holds t_star and only the normal is left: a hit whose extrapolated point lies in another slab)   [13..15] zero
 *   [16..23] the 8 output floats of tsdf_hip_raycast.
 * tsdf_hip_raycast_begin   writes the start record of every ray (identical on every rank).
 * tsdf_hip_raycast_advance zero-fills d_delta, then resumes every suspended ray of d_state that this
 *   handle is responsible for (needed plane inside its OWNED slab; or ray index % world == rank while the
 *   ray has not needed a voxel yet) until it finishes or needs another slab's voxel, and writes the new
 *   record to d_delta.  Exactly one rank touches a ray per round, so the caller merges with an integer
 *   SUM all-reduce of the deltas (RCCL) and overwrites the records whose status word is non-zero; it
 *   repeats until no record is suspended (at most world + 2 rounds).  The refinement walk and the trilinear samples of a hit read up
 *   to tsdf_hip_render_halo(params) planes beyond the owned slab: create the handles with that halo and
 *   refresh it (tsdf_hip_get/set_planes_device) before rendering.  Synchronises the stream. */
#define TSDF_HIP_RAY_RECORD_INTS 24
int tsdf_hip_raycast_begin(tsdf_handle h, const float rot[9], const float origin[3], int downsample,
                           int32_t *d_state);
int tsdf_hip_raycast_advance(tsdf_handle h, const float rot[9], const float origin[3], int downsample,
                             int rank, int world, const int32_t *d_state, int32_t *d_delta);
/* Scalable form: the caller keeps, per rank, a COMPACT list of the records it is responsible for (word 11 of a
 * record is the ray's pixel index, set by tsdf_hip_raycast_begin) and advances it in place; suspended records then
 * travel point-to-point to the owner of their next voxel, finished ones stay.  Traffic is proportional to the
 * rays that cross a slab boundary instead of to the image. */
int tsdf_hip_raycast_advance_list(tsdf_handle h, const float rot[9], const float origin[3], int downsample,
                                  int rank, int world, int32_t *d_records, size_t count);
int tsdf_hip_render_halo(const tsdf_params *p);

/* Placement of the voxel planes (not in the reference).  On MI355X the physical pages behind an allocation decide how
 * fast a streaming read-modify-write of it runs (a 2048^3 volume integrates in 17.9 ms or in 18.4-19.0 ms depending on
 * the allocation), so tsdf_hip_create allocates the planes of a volume of 4 GiB or more up to
 * `alloc_tries` times (TSDF_HIP_ALLOC_TRIES in the environment, default 3, 1 = off; a second candidate is only
 * tried while it fits next to the first), sweeps each candidate once and keeps the fastest.  The search ends early
 * at a candidate that streams at >= 5.55 TB/s (the fastest class) and goes on for up to alloc_tries more (8 at most)
 * while none has reached 5.15 TB/s.  This reports what happened: ms[i] = probe sweep of candidate i (negative = not
 * probed), *chosen = the one kept; returns the number of candidates tried. */
int tsdf_hip_alloc_probe(tsdf_handle h, float ms[8], int32_t *chosen);

/* The reference's integrateCloud visits only the voxels pcl::FrustumCulling keeps (getFrustumCulledVoxels,
 * tsdf_volume_octree.cpp:619-652, called at impl/tsdf_volume_octree.hpp:93-94): a pyramid of 1.1 x the field of view
 * AROUND THE OPTICAL AXIS between near = min_sensor_dist and far = max_sensor_dist, tested per voxel centre as
 * `pt.dot(plane) <= 0` for six planes (pt = (x, y, z, 1), the dot reduced as (p0 + p1) + (p2 + p3), fp32, no FMA).
 * For an ordinary camera that pyramid contains every ray of the image and the cull only removes voxels updateVoxel
 * rejects anyway; with a principal point more than ~10 % off centre, or a range plane cutting the volume, it decides
 * voxels.  To integrate exactly the reference's voxels in EVERY regime, hand the six planes of the frame's pose to
 * tsdf_hip_set_reference_cull before each integrate call -- cpu_tsdf::TSDFVolumeOctree::integrateCloud and the Python
 * binding do so by default (setReferenceCull(false) opts out).
 *   planes  l, r, t, b, far, near (the order of PCL's test), 4 floats each, in the VOLUME frame.  They are PCL / Eigen
 *           arithmetic on the forward pose `trans` (pcl::FrustumCulling::applyFilter with camera pose
 *           trans.cast<float>() * cam2robot, tsdf_volume_octree.cpp:633-646), so the caller computes them: the C++
 *           shell with the caller's Eigen, anyone else with tsdf_hip_reference_cull_planes.  NULL = no cull (the
 *           conservative superset: every voxel updateVoxel itself accepts).
 * Cost: none when the planes provably keep every voxel of the slab (eight corner voxels, evaluated per launch on the
 * host: the usual case -- the launch is then the ordinary one); otherwise the frame's ROW INTERVALS carry the cull
 * (k_rows: per voxel row the exact x range the six planes keep, found by bisection on the very float expression of the
 * per-voxel test) and k_integrate masks by interval -- a few per cent.  Non-finite planes and the RGB_NORMALIZED / LAB
 * kernels test the six planes per voxel.  The planes stay in force for every later integrate call on the handle: set
 * them per frame.  tests/test_oracle_golden.py pins the restatement against the compiled reference,
 * tests/test_index_box.py the intervals and tests/test_integrate_gpu.py the kernels against the restatement. */
int tsdf_hip_set_reference_cull(tsdf_handle h, const float planes[24]);
/* Two frames in one sweep (not in the reference; its integrateCloud takes one cloud).  Integrates frame A, then frame B
 * -- the same voxels, bit for bit, as two tsdf_hip_integrate_device calls in that order -- but where the handle and both
 * poses allow it (PACKED layout, nx a multiple of 4, every voxel of the slab inside the sensor range and the image for
 * BOTH poses: the camera-outside-the-volume case; the cull planes keeping the whole slab) one kernel reads each voxel's
 * words once, applies updateVoxel for A and then for B in registers and writes them back once: half the HBM bytes per
 * frame (k_integrate2).  Otherwise it IS two launches.  planes_a / planes_b: as
 * tsdf_hip_set_reference_cull (NULL = none); the handle keeps frame B's.  n_observed (nullable, 2 values): per frame, as
 * the single call reports; tsdf_hip_last_count_detail then describes the pair.  *fused (nullable): 1 if one sweep did
 * it.  Each frame's colour image must lie behind its depth image in one allocation (as tsdf_hip_integrate_device
 * prefers); device pointers, asynchronous unless n_observed.  On a multi-GPU set both frames go to every slab's ring
 * and every slab decides for itself; *fused = 1 when every slab swept once, n_observed = the slabs' sums. */
int tsdf_hip_integrate_device2(tsdf_handle h, const float *d_depth_a, const uint32_t *d_bgra_a, const float cam_from_vol_a[12],
                               const float *planes_a, const float *d_depth_b, const uint32_t *d_bgra_b,
                               const float cam_from_vol_b[12], const float *planes_b, uint64_t *n_observed, int32_t *fused);

/* Frame pairing for the pipelined host entry points (tsdf_hip_frame_begin / _commit, tsdf_hip_integrate_async); not in the
 * reference.  While on, a committed frame is uploaded at once but its kernel launch waits: when the NEXT frame is
 * committed the two are integrated by one tsdf_hip_integrate_device2-style call (one sweep of the volume for both where
 * the handle and both poses allow it, else two launches, frame order kept either way); any other entry point that reads
 * or writes the volume (synchronize, integrate, download, march, raycast, sample, save, reset, ...) first launches a
 * waiting frame on its own.  Results are identical to pairing off; what changes is when the kernels run (a stream of
 * frames: 11.2 instead of 12.7 ms per frame at 2048^3 + colour; without colour a pair is two launches of
 * the pipelined single-frame kernel, which is the faster way there since round 6).  Each frame keeps the cull planes that
 * were in force (tsdf_hip_set_reference_cull) when IT was committed.  Off by default.  On a multi-GPU set
 * (tsdf_hip_create_multi, round 6) every slab pairs the frames of its own ring: one sweep of a slab per pair where both
 * poses see all of THAT slab. */
int tsdf_hip_set_frame_pairing(tsdf_handle h, int on);

/* Host only: those six planes from the forward pose `trans` (row-major 4x4 doubles, camera -> volume, what
 * integrateCloud is called with) and the camera of `p` [PCL-recall: filters/impl/frustum_culling.hpp]. */
int tsdf_hip_reference_cull_planes(const tsdf_params *p, const double trans[16], float planes[24]);
/* 1 if that cull cannot change results for ANY pose with these parameters as far as the field of view goes (the
 * culling pyramid contains every ray of the image and max_sensor_dist is finite and moderate); 0 for a principal point
 * so far off centre, or a range so large (>= 1e15, inf), that the reference drops voxels which project into the image.
 * Report-only since the planes are applied per launch. */
int tsdf_hip_reference_cull_is_noop(const tsdf_params *p);

/* getFxn / getGradient / getHessian -- tsdf_volume_octree.cpp:655-828, batched.
 *   xyz n x 3 floats; val n floats (nullable); grad n x 3 (nullable); hess n x 9 row-major (nullable);
 *   ok n bytes: 1 where the reference returns true.  A Z-slab handle answers only for points whose lower-corner
 *   plane it owns (exactly one handle of a partition does; it needs plane z_end fresh in its halo). */
int tsdf_hip_sample(tsdf_handle h, const float *xyz, size_t n, float *val, float *grad,
                    float *hess, uint8_t *ok);

/* renderColoredView's colour lookup -- tsdf_volume_octree.cpp:443-448: for n points (volume frame) the voxel
 * Octree::getContainingVoxel returns (src/lib/octree.cpp:112-133,628-643) and its RGBNode colour (r,g,b; zeros
 * if the volume stores no colour).  found[i] = 0 where the reference gets NULL. */
int tsdf_hip_lookup_rgb(tsdf_handle h, const float *xyz, size_t n, uint8_t *rgb, uint8_t *found);

/* MarchingCubesTSDFOctree::reconstruct -- src/lib/marching_cubes_tsdf_octree.cpp:108-236
 * (+ pcl::MarchingCubes::createSurface).  color_mode: 0 none, 1 setColorByRGB, 2 setColorByConfidence.
 * tsdf_hip_march runs the kernels and reports the triangle count; tsdf_hip_march_fetch copies
 * n_tri*9 floats (3 vertices, volume frame) and, if rgb != NULL, n_tri*9 bytes (r,g,b per vertex) and,
 * if cell != NULL, n_tri uint64 cell keys ((x<<42)|(y<<21)|z of the base voxel).  Triangles come
 * out in the reference's order (octree pre-order = Morton order with x as the high bit). */
int tsdf_hip_march(tsdf_handle h, float w_min, int color_mode, uint64_t *n_tri);
int tsdf_hip_march_fetch(tsdf_handle h, float *verts, uint8_t *rgb, uint64_t *cell);
/* Report-only: device milliseconds of the last tsdf_hip_march by phase -- ms[0] classify (k_mc_classify), ms[1] count
 * read-back + sort + scan, ms[2] emit (k_mc_emit) -- and the number of active cells. */
int tsdf_hip_march_timing(tsdf_handle h, float ms[3], uint64_t *n_cells);
/* Report-only: out[0] active cells, out[1] triangles, out[2] bytes of the distance plane the classify pass of the last
 * tsdf_hip_march REQUESTED (with the band flags of integrateCloud deciding, it reads only the quads an in-band
 * observation is near: src/lib/marching_cubes_tsdf_octree.cpp:179-236 visits every leaf), out[3] bit 0: the flags were
 * in use (0: every plane was read -- after an upload / load the flags say nothing until reset); out[3] bit 1: the
 * weight test (marching_cubes_tsdf_octree.cpp:98, w < w_min) was not evaluated because it could not fail -- PACKED counts,
 * planes written only by integrateCloud since the reset, no halo plane, w_min <= min(1, max_weight): a corner with
 * |d| < 1 has been observed, and an observation counts. */
int tsdf_hip_march_stats(tsdf_handle h, uint64_t out[4]);
/* The same copies into DEVICE buffers of the caller, asynchronous on the handle's stream (multi-GPU mesh merge:
 * the buffers go straight to RCCL). */
int tsdf_hip_march_fetch_device(tsdf_handle h, float *d_verts, uint8_t *d_rgb, uint64_t *d_cell);

/* Block transfer of raw voxels (parity tests, save/load, halo exchange).  Coordinates are global
 * grid indices; the block must lie inside the handle's slab + halo.  Any pointer may be NULL.
 * rgb is 3 bytes per voxel (r,g,b).  *_device variants take device pointers (rgb then 4 bytes). */
int tsdf_hip_download(tsdf_handle h, int x0, int y0, int z0, int nx, int ny, int nz, float *d,
                      float *w, uint8_t *rgb);
int tsdf_hip_upload(tsdf_handle h, int x0, int y0, int z0, int nx, int ny, int nz, const float *d,
                    const float *w, const uint8_t *rgb);

/* The float colour state of the voxels of planes z0 .. z0+nz (nz * ny * nx floats): plane_index 0..3 = r_n_, g_n_,
 * b_n_, i_ of RGBNormalized (octree.h:217-222) or 0..2 = L_, A_, B_ of LABNode (octree.h:296-298).  The reference has
 * no accessor for it (the members are public); the parity tests read it.  Single-device handles only. */
int tsdf_hip_download_color_state(tsdf_handle h, int plane_index, int z0, int nz, float *out);

/* save / load -- src/lib/tsdf_volume_octree.cpp:222-275 (+ Octree::serialize / deserialize,
 * src/lib/octree.cpp:289-304,360-367,645-678): the reference's .vol checkpoint, readable and writable by both
 * sides.  The dense grid becomes an octree whose uniform subtrees are single leaves (lossless for every
 * reader that looks voxels up by position); both directions stream the grid through host memory in cubic
 * blocks (tuning "vol_chunk", 256), so host memory stays at one block whatever the resolution.  Needs a
 * cubic power-of-two grid and a handle that owns all of it.
 *   meta      what the file carries that a tsdf_params does not; NULL on save = max_cell_size of one voxel,
 *             not empty, no depth/variance weighting, identity transform.
 * tsdf_hip_load makes a NEW handle from the file's header; `defaults` supplies what the file does not hold
 * (device, layout, xform_order; NULL = tsdf_hip_default_params).  With layout AUTO a file whose weights
 * are not min(k, max_weight) is read again into the F32W layout; PACKED fails with E_UNSUPPORTED. */
typedef struct tsdf_vol_meta {
  float max_cell_size[3];      /* setMaxVoxelSize, tsdf_volume_octree.h:166 (unused by a dense grid) */
  int32_t is_empty;            /* tsdf_volume_octree.h:286,356 */
  int32_t weight_by_depth;     /* :358 */
  int32_t weight_by_variance;
  double global_transform[16]; /* setGlobalTransform :132, row-major */
} tsdf_vol_meta;
int tsdf_hip_save(tsdf_handle h, const char *filename, const tsdf_vol_meta *meta);
int tsdf_hip_load(const char *filename, const tsdf_params *defaults, tsdf_handle *out, tsdf_params *params_out,
                  tsdf_vol_meta *meta_out);
/* The same into a multi-GPU set (tsdf_hip_create_multi). */
int tsdf_hip_load_multi(const char *filename, const tsdf_params *defaults, const int32_t *devices, int n_devices,
                        tsdf_handle *out, tsdf_params *params_out, tsdf_vol_meta *meta_out);

/* The same writer and reader for a volume that is not one handle (Z-slabs on several GPUs): the voxels
 * move through callbacks, one cubic block of `edge`^3 voxels at a time (d, w: edge^3 floats; rgb: 3 edge^3
 * bytes, NULL without colour; x fastest).  A callback returns 0 to go on; anything else aborts the call,
 * which then returns that value.  No device is touched by these two functions themselves.
 *   tsdf_hip_save_blocks  p describes the whole grid (res, size, ... ; slab fields ignored).  `fetch` is
 *                         called once per block and once more for each block that is not uniform.
 *   tsdf_hip_load_blocks  `on_header` receives the file's parameters (fields the file does not carry are
 *                         taken from `defaults`, NULL = tsdf_hip_default_params) before the first block;
 *                         then `store` is called exactly once per block.  store == NULL reads the header only. */
typedef int (*tsdf_block_fn)(void *user, int x0, int y0, int z0, int edge, float *d, float *w, uint8_t *rgb);
typedef int (*tsdf_header_fn)(void *user, const tsdf_params *p, const tsdf_vol_meta *meta);
int tsdf_hip_save_blocks(const tsdf_params *p, const tsdf_vol_meta *meta, const char *filename,
                         tsdf_block_fn fetch, void *user);
int tsdf_hip_load_blocks(const char *filename, const tsdf_params *defaults, tsdf_header_fn on_header,
                         tsdf_block_fn store, void *user);

/* Whole planes [z0, z0+nz) to / from packed DEVICE buffers ([nz][ny][nx]; rgb as uint32 r|g<<8|b<<16),
 * asynchronous on the handle's stream.  This is the halo-exchange primitive: the buffers are what the
 * caller hands to RCCL send/recv.  Planes may lie in the halo.  Any pointer may be NULL. */
int tsdf_hip_get_planes_device(tsdf_handle h, int z0, int nz, float *d, float *w, uint32_t *rgb);
int tsdf_hip_set_planes_device(tsdf_handle h, int z0, int nz, const float *d, const float *w,
                               const uint32_t *rgb);

/* Raw device pointers of the SoA planes and their geometry: element index of voxel (x,y,z_global) =
 * ((z_global - z_first)*ny + y)*pitch + x.  In the PACKED layout *w is NULL (see tsdf_hip_layout) and the
 * count sits in byte 3 of *rgb; without colour neither is exposed -- use the get/set_planes calls. */
int tsdf_hip_device_planes(tsdf_handle h, float **d, float **w, uint32_t **rgb, int64_t *pitch,
                           int32_t *z_first, int32_t *nz_alloc);

/* TSDF_LAYOUT_F32W or TSDF_LAYOUT_PACKED: what AUTO resolved to for this handle. */
int tsdf_hip_layout(tsdf_handle h);

/* Voxel-centre tables actually used by the kernels: the reference's octree node centres
 * (src/lib/octree.cpp:244-266 split arithmetic) for each axis.  out has res[axis] floats. */
int tsdf_hip_centers(tsdf_handle h, int axis, float *out);

/* Report: the last integrate launch on this handle (slab 0 of a set) -- out[0] = the kernel: 0 k_integrate's general
 * instance, 1 its ALLIN instance (the host proved from the slab's eight corner voxels that EVERY voxel is inside the sensor
 * range and projects inside the image with a pixel to spare -- the camera-outside-the-volume case -- so the per-voxel range /
 * image-bounds tests are compiled out), 2 k_integrate2 (two frames in one sweep) -- in the low byte; bit 8 (0x100) is set
 * when the ALLIN launch ran the software-pipelined row loop (k_integrate_p: PACKED layout without colour); out[1] = 1 with the certified fp32
 * projection; out[2] = 0 no row intervals, 1 row intervals + block flags (the frame sees part of the slab), 2 the same with
 * the reference's frustum cull carried in the intervals; out[3] = blocks launched (0: nothing could be observed). */
int tsdf_hip_last_launch_info(tsdf_handle h, int32_t out[4]);

const char *tsdf_hip_error_string(int code);
const char *tsdf_hip_last_error(void);
int tsdf_hip_device_count(void);
/* ABI version of this header. */
int tsdf_hip_abi_version(void);
#define TSDF_HIP_ABI_VERSION 14

#ifdef __cplusplus
}
#endif
#endif /* TSDF_HIP_H */
