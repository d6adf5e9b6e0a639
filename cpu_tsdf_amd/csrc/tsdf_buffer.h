// Buffer-descriptor memory access for the streaming kernels (device code only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- memory access through buffer descriptors ----------------------------------------------------------
// Every global access of k_integrate (and of k_mc_classify) goes through a 128-bit buffer resource built from
// wave-uniform values
// (blockIdx-derived plane/row-group base, frame base): the per-lane part is a 32-bit byte offset, so the
// row loop carries no 64-bit address arithmetic in the VALU, the per-row step is the instruction's scalar
// offset, and out-of-range lanes (pixel -1, rows past the grid) are absorbed by the hardware bounds check
// (loads return 0, stores are dropped) instead of by exec-mask branches.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// Cache-policy bits of the voxel STREAM accesses (128-bit loads, all stores): on gfx94x / gfx950 bit 0 = sc0, bit 1 = nt,
// bit 4 = sc1.  The voxel planes are touched once per frame and never again before 69 GB of other planes went by, so
// they are marked non-temporal (nt) on BOTH sides: measured on MI355X, 2048^3 + colour, k_integrate 17.6-17.7 ms with the
// default policy, 17.6-18.0 with nt on the loads only, 17.6-17.7 on the stores only, **16.4-16.6 with both**
// (profiles/r03_ab_k_integrate_cache_policy.txt; sc0 / sc1 on top change nothing).  The frame gather keeps the default
// policy: the 2.4 MB frame is what the L2 should hold.  A/B knobs: tools/build_variant.py NAME -DTSDF_STREAM_LD_AUX=0 ...
#ifndef TSDF_STREAM_LD_AUX
#define TSDF_STREAM_LD_AUX 2
#endif
#ifndef TSDF_STREAM_ST_AUX
#define TSDF_STREAM_ST_AUX 2
#endif

static __device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, (int)bytes, 0x00020000);
}
static __device__ __forceinline__ uint32_t bload32(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
}
static __device__ __forceinline__ u4 bload128(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, TSDF_STREAM_LD_AUX);
}
static __device__ __forceinline__ void bstore32(rsrc_t r, unsigned voff, unsigned soff, uint32_t v) {
  __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)voff, (int)soff, TSDF_STREAM_ST_AUX);
}
static __device__ __forceinline__ uint32_t bload8(rsrc_t r, unsigned voff, unsigned soff) {
  return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(r, (int)voff, (int)soff, 0) & 0xffu;
}
static __device__ __forceinline__ void bstore128(rsrc_t r, unsigned voff, unsigned soff, u4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, TSDF_STREAM_ST_AUX);
}

// Structured (2-D) buffer access: the descriptor carries a row stride and a row count, the instruction a row index
// and a byte offset inside the row (`buffer_load_dword ... idxen offen`).  Clang has no builtin for the struct form;
// the LLVM intrinsic is reached through its name (the <4 x i32> descriptor flavour).
typedef int i4_rsrc __attribute__((ext_vector_type(4)));
__device__ unsigned tsdf_struct_buffer_load_u32(i4_rsrc rsrc, int vindex, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.struct.buffer.load.i32");

static __device__ __forceinline__ i4_rsrc make_rsrc_2d(const void *p, unsigned row_bytes, unsigned rows) {
  const unsigned long long a = (unsigned long long)p;
  i4_rsrc r;
  r.x = (int)(a & 0xffffffffull);
  r.y = (int)(((a >> 32) & 0xffffull) | ((unsigned long long)(row_bytes & 0x3fffu) << 16));  // STRIDE: 14 bits
  r.z = (int)rows;        // NUM_RECORDS counts rows when STRIDE != 0
  r.w = 0x00020000;       // DATA_FORMAT = 32, everything else off (no swizzle, no add-tid)
  return r;
}
