// Correctly rounded division with a SHARED reciprocal, for gfx950.
//
// hipcc expands an IEEE fp32/fp64 `a / b` into v_div_scale + v_rcp + a fixed FMA ladder + v_div_fmas +
// v_div_fixup (LLVM AMDGPUTargetLowering::LowerFDIV32/64).  Several quotients in the integrate kernel
// share one divisor (u and v both divide by g.z; the d update and the three colour channels all divide
// by w + 1), so the reciprocal refinement -- the part of the ladder that depends only on b -- is done
// once and each quotient costs only its own residual steps.  The FMA ladder below is the same one the
// compiler emits, minus the exponent scaling (v_div_scale) and special-case fixup (v_div_fixup), which
// are the identity while both operands are finite, non-zero and far from the exponent limits.  Callers
// check that range and use a plain `/` outside it, so results are bit-identical to IEEE division
// everywhere (verified exhaustively-at-random against numpy by tests/test_div_gpu.py through
// tsdf_hip_selftest_div*).
#pragma once
#include <hip/hip_runtime.h>

struct Rcp32 {
  float nb;  // -b
  float y;   // refined reciprocal (LLVM's Fma1)
  bool safe; // b within the no-scaling exponent window
};

static __device__ __forceinline__ bool div_safe_f32(float v) {
  const float a = fabsf(v);
  return a >= 0x1p-40f && a <= 0x1p40f;  // false for 0, denormals, inf, NaN
}

static __device__ __forceinline__ Rcp32 rcp32_prepare(float b) {
  Rcp32 r;
  r.safe = div_safe_f32(b);
  r.nb = -b;
  const float y0 = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(r.nb, y0, 1.0f);
  r.y = __builtin_fmaf(e, y0, y0);
  return r;
}

// a / b, b described by r (prepared from the same b).
static __device__ __forceinline__ float div32(float a, float b, const Rcp32 &r) {
  if (r.safe && div_safe_f32(a)) {
    const float q0 = a * r.y;
    const float r0 = __builtin_fmaf(r.nb, q0, a);
    const float q1 = __builtin_fmaf(r0, r.y, q0);
    const float r1 = __builtin_fmaf(r.nb, q1, a);
    return __builtin_fmaf(r1, r.y, q1);
  }
  return a / b;
}

struct Rcp64 {
  double nb;
  double y;  // LLVM's Fma3
};

// b must be a positive finite value that came from a float (exponent window of fp32 << fp64's).
static __device__ __forceinline__ Rcp64 rcp64_prepare(double b) {
  Rcp64 r;
  r.nb = -b;
  const double y0 = __builtin_amdgcn_rcp(b);
  const double e0 = __builtin_fma(r.nb, y0, 1.0);
  const double y1 = __builtin_fma(y0, e0, y0);
  const double e1 = __builtin_fma(r.nb, y1, 1.0);
  r.y = __builtin_fma(y1, e1, y1);
  return r;
}

static __device__ __forceinline__ double div64(double a, const Rcp64 &r) {
  const double q = a * r.y;
  const double res = __builtin_fma(r.nb, q, a);
  return __builtin_fma(res, r.y, q);
}
