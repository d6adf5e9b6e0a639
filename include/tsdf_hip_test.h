/* libtsdf_hip_test.so -- the test build of the library: everything include/tsdf_hip.h declares (same sources, same
 * kernels, same flags) PLUS the hooks below, compiled in by -DTSDF_HIP_TEST_HOOKS.  The product library
 * (libtsdf_hip.so) exports none of them (tests/test_abi.py checks both export tables).  They exist so that tests can
 * reach inside -- device-side dividers, the projection certificate, the cull predicates on the host, calibration sweeps
 * of exactly known bytes -- and switch launch-shape knobs at run time; nothing here is part of the cpu_tsdf boundary. */
#ifndef TSDF_HIP_TEST_H
#define TSDF_HIP_TEST_H
#include "tsdf_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hooks (not part of the cpu_tsdf interface): the kernels' shared-reciprocal dividers applied
 * element-wise to host arrays, out[i] = a[i] / b[i]; the f64 variant requires b > 0 finite with a
 * float-sized exponent (it is only ever used on (double)g.z).  Used by tests/test_div_gpu.py to prove
 * bit equality with IEEE division. */
int tsdf_hip_selftest_div_f32(const float *a, const float *b, float *out, size_t n);
int tsdf_hip_selftest_div_f64(const double *a, const double *b, double *out, size_t n);

/* Test hook: the PACKED layout's weighted-mean divider, out[i] = a[i] / k[i] for integer counts k in [1, 256]: the table
 * reciprocal + scale-free ladder where its result is a normal number (fast[i] = 1), IEEE division elsewhere. */
int tsdf_hip_selftest_div_count(const float *a, const uint32_t *k, float *out, uint8_t *fast, size_t n);

/* Test hook: out[i] = v_cvt_pk_u8_f32(in[i], byte 1, 0xAABBCCDD) -- the instruction the colour update packs its bytes
 * with; the tests pin its rounding (nearest even), saturation and byte selection. */
int tsdf_hip_selftest_cvt_pk_u8(const float *in, size_t n, uint32_t *out);

/* Test hook: out[i] = the device's std::exp(float) of the variance weighting (the host libm's expf restated). */
int tsdf_hip_selftest_expf(const float *in, size_t n, float *out);

/* Test hooks for TSDF_COLOR_LAB: the device's RGB2LAB of n pixels (b,g,r,a bytes each -> L,A,B,0 floats each) and
 * LAB2RGB of n L,A,B triples (-> r | g<<8 | b<<16 each); references octree.cpp:436-481 and :483-527. */
int tsdf_hip_selftest_rgb2lab(const uint8_t *bgra, size_t n, float *lab4);
int tsdf_hip_selftest_lab2rgb(const float *lab3, size_t n, uint32_t *rgb);

/* Test hook: the integrate kernel's pixel projection (reprojectPoint, tsdf_volume_octree.cpp:611-617) on
 * n arbitrary camera-frame points g (x,y,z triples, z > 0) with this volume's intrinsics: pix_fast =
 * certified-fp32 path with exact fallback (what the kernel uses), pix_exact = fp64 path, both v*W+u or
 * -1; ambiguous[i] = 1 where the fp32 path declined to decide. */
int tsdf_hip_selftest_project(tsdf_handle h, const float *g, size_t n, int32_t *pix_fast,
                              int32_t *pix_exact, uint8_t *ambiguous);

/* Test hook: structured buffer loads (row index v, byte offset 4 u, descriptor of H rows of W floats, image `plane` of
 * `planes` back-to-back images selected by the scalar offset) at n (u, v) pairs, inside and outside the image: the
 * integrate kernel's frame gather relies on what the hardware returns out of range. */
int tsdf_hip_selftest_struct_oob(const float *img, int W, int H, int planes, int plane, const int32_t *uv, uint32_t *out, int n);

/* Test hook: the voxel Octree::getContainingVoxel (src/lib/octree.cpp:112-133,628-643) returns for n points,
 * as the raycast kernel computes it: idx = i, j, k per point, or -1, -1, -1 where the reference returns NULL. */
int tsdf_hip_selftest_containing(tsdf_handle h, const float *xyz, size_t n, int32_t *idx);
/* Test hook, host only: the voxel index box {lo x,y,z, hi x,y,z} (inclusive) the integrate launch is restricted to for
 * this pose; *state = 0 box valid, 1 nothing can be observed, 2 no claim (the launch then covers the whole slab). */
int tsdf_hip_selftest_index_box(const tsdf_params *p, const float cam_from_vol[12], int32_t box[6], int32_t *state);
/* Test hook, host only: the brick cull's per-block predicate over the whole grid (blocks of bx_vox voxels along x by
 * by_rows rows of one plane); flags[(z * gy + by) * gx + bx] with gx = ceil(nx / bx_vox), gy = ceil(ny / by_rows). */
int tsdf_hip_selftest_block_flags(const tsdf_params *p, const float cam_from_vol[12], int bx_vox, int by_rows,
                                  uint8_t *flags);

/* Test hook, host only: the row intervals of a LIVE integrate launch over the whole grid -- per voxel row (y, z) the x
 * range outside of which no voxel is integrated: words[z * res_y + y] = lo | len << 16 (an empty row has lo = 0xffff).
 * Conservative for updateVoxel's own tests; with `planes` (the reference cull's six planes, else NULL) additionally
 * EXACTLY the voxels pcl::FrustumCulling keeps. */
int tsdf_hip_selftest_row_intervals(const tsdf_params *p, const float cam_from_vol[12], const float *planes, uint32_t *words);

/* Test / profiling hook: one read-modify-write sweep of the owned slab's SoA planes with the integrate
 * kernel's access shape and no other work; reports the exact bytes it read and wrote.  Used to
 * calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE (tools/prof_integrate.py). */
int tsdf_hip_selftest_sweep(tsdf_handle h, uint64_t *bytes_read, uint64_t *bytes_written);
/* The same for NARROW reads: a read-only sweep of the owned distance planes that takes one 32-bit word per
 * stride_bytes (4: a wave instruction covers 256 contiguous bytes; 64 / 128: every lane touches its own 64 B / 128 B
 * piece).  Reports the bytes spanned and the words read; FETCH_SIZE of kernel k_calib_read<stride> says what the
 * counter tallies for that shape (bench.py --calib, tools/make_profile_summary.py). */
int tsdf_hip_selftest_read_sweep(tsdf_handle h, int stride_bytes, uint64_t *span_bytes, uint64_t *dwords_read);


/* Test / tuning hook: 256-thread blocks per CU the runtime admits for k_mc_classify (out[0]) and k_mc_emit (out[1]). */
int tsdf_hip_selftest_occupancy_mc(int out[2]);

/* Test / A-B hook: set a launch-shape knob ("rows_per_block", "blocks_per_cu", "fast_projection",
 * "mc_flush_at", "mc_skip", "cull", "vol_chunk", "plain_kernel", "alloc_tries", "allin", "refcull_plain", "live_log2tx", "fuse2" -- the
 * TSDF_HIP_* environment variables, which the product library reads once) at run time.  No knob changes results. */
int tsdf_hip_set_tuning(const char *name, int value);

/* Test hook: position-dependent 64-bit checksums of the handle's OWNED planes, computed on the device (sum over the words
 * of word * odd(index)): out[0] d, out[1] w (F32W), out[2] rgb | count words (colour), out[3] count bytes (colourless
 * PACKED); 0 for a plane the layout does not have.  Lets whole 2048^3 volumes be compared without a 69 GB download. */
int tsdf_hip_selftest_checksum(tsdf_handle h, uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* TSDF_HIP_TEST_H */
