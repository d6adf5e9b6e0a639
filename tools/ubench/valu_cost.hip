// Micro-benchmark: SIMD occupancy of single vector instructions on gfx950, in cycles per wave64 instruction (eight waves per SIMD,
// eight independent destination registers per wave, 4000 x 8 instructions per wave; cycles from the wall clock and the engine
// clock the runtime reports).  Why: a "VALU diet" must be counted in cycles, not instructions --
// this is the price list.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_cost.hip -o tools/ubench/valu_cost ; run: tools/ubench/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef float f2_ __attribute__((ext_vector_type(2)));

#define KERNEL(NAME, TYPE, INIT, ASM, ...)                                                        \
  __global__ void __launch_bounds__(256) NAME(float *out, int iters, float a, float b) {          \
    TYPE x[8];                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) x[i] = INIT;                                    \
    const TYPE A = (TYPE)a, B = (TYPE)b;                                                          \
    (void)A; (void)B;                                                                             \
    for (int it = 0; it < iters; ++it) {                                                          \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(x[i]) : __VA_ARGS__ : "vcc", "scc", "s20", "s22", "s23", "s24", "s25"); \
    }                                                                                             \
    float s = 0.f;                                                                                \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) s += (float)x[i];                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                               \
  }

#define F32(NAME, ASM) KERNEL(NAME, float, (float)(threadIdx.x + i), ASM, "v"(A), "v"(B))
#define U32(NAME, ASM) KERNEL(NAME, unsigned, (unsigned)(threadIdx.x + i), ASM, "v"(A), "v"(B))
#define F64(NAME, ASM) KERNEL(NAME, double, (double)(threadIdx.x + i), ASM, "v"(A), "v"(B))

F32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
F32(k_fmac_f32, "v_fmac_f32 %0, %1, %2")
F32(k_mul_f32, "v_mul_f32 %0, %0, %1")
F32(k_add_f32, "v_add_f32 %0, %0, %1")
F32(k_rcp_f32, "v_rcp_f32 %0, %0")
F32(k_fract_f32, "v_fract_f32 %0, %0")
F32(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
F32(k_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
F32(k_cvt_f32_ubyte2, "v_cvt_f32_ubyte2 %0, %0")
F32(k_cvt_pk_u8_f32, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
F32(k_cmp_gt_f32, "v_cmp_gt_f32 vcc, %0, %1")
F32(k_cmp_class_f32, "v_cmp_class_f32 vcc, %0, %1")
F32(k_div_fixup_f32, "v_div_fixup_f32 %0, %0, %1, %2")
U32(k_mov_b32, "v_mov_b32 %0, %1")
U32(k_add_u32, "v_add_u32 %0, %0, %1")
U32(k_and_b32, "v_and_b32 %0, %0, %1")
U32(k_lshl_or_b32, "v_lshl_or_b32 %0, %0, 3, %1")
U32(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 8")
U32(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
U32(k_min_u32, "v_min_u32 %0, %0, %1")
U32(k_min3_u32, "v_min3_u32 %0, %0, %1, %2")
U32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
U32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
U32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
U32(k_cmp_gt_u32, "v_cmp_gt_u32 vcc, %0, %1")
U32(k_cmp_eq_u16_sdwa, "v_cmp_eq_u16_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:DWORD")
U32(k_readlane, "v_readlane_b32 s20, %0, 3")
U32(k_writelane, "v_writelane_b32 %0, s20, 3")
U32(k_cndmask_e64_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[22:23]")
U32(k_cndmask_other_dst, "v_cndmask_b32 %0, %1, %2, vcc")
U32(k_cmp_then_cndmask, "v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc")
U32(k_cmp_sgpr_then_cndmask, "v_cmp_gt_u32_e64 s[22:23], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[22:23]")
U32(k_cmp_then_4cndmask, "v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc")
U32(k_cmp_sgpr_then_4cndmask, "v_cmp_gt_u32_e64 s[22:23], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[22:23]\n\tv_cndmask_b32_e64 %0, %0, %2, s[22:23]\n\tv_cndmask_b32_e64 %0, %0, %1, s[22:23]\n\tv_cndmask_b32_e64 %0, %0, %2, s[22:23]")
U32(k_salu_vcc_then_cndmask, "s_mov_b64 vcc, s[22:23]\n\tv_cndmask_b32 %0, %0, %1, vcc")
U32(k_cmp_cnd_fma_cnd, "v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_add_u32 %0, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc")
U32(k_cmp_add_cnd, "v_cmp_gt_u32 vcc, %1, %2\n\tv_add_u32 %0, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc")
U32(k_cmp_2add_cnd, "v_cmp_gt_u32 vcc, %1, %2\n\tv_add_u32 %0, %0, %2\n\tv_add_u32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
U32(k_cmp_sand_cnd, "v_cmp_gt_u32 vcc, %1, %2\n\ts_and_b64 vcc, vcc, s[22:23]\n\tv_cndmask_b32 %0, %0, %2, vcc")
U32(k_cmp64_sand_cnd64, "v_cmp_gt_u32_e64 s[24:25], %1, %2\n\ts_and_b64 s[24:25], s[24:25], s[22:23]\n\tv_cndmask_b32_e64 %0, %0, %2, s[24:25]")
U32(k_cmp_cnd_constsrc, "v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, -1, %0, vcc")
U32(k_addc_vcc, "v_cmp_gt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc")
U32(k_bfi_b32, "v_bfi_b32 %0, %1, %0, %2")
U32(k_perm_b32, "v_perm_b32 %0, %0, %1, %2")
U32(k_or3_b32, "v_or3_b32 %0, %0, %1, %2")
U32(k_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
U32(k_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
U32(k_xor_b32, "v_xor_b32 %0, %0, %1")
U32(k_sub_u32, "v_sub_u32 %0, %0, %1")
U32(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
U32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
F32(k_max_f32, "v_max_f32 %0, %0, %1")
F32(k_med3_f32, "v_med3_f32 %0, %0, %1, %2")
F32(k_mul_legacy, "v_mul_legacy_f32 %0, %0, %1")
F64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
F64(k_mul_f64, "v_mul_f64 %0, %0, %1")
F64(k_add_f64, "v_add_f64 %0, %0, %1")
F64(k_rcp_f64, "v_rcp_f64 %0, %0")

__global__ void __launch_bounds__(256) k_pk_fma_f32(float *out, int iters, float a, float b) {
  f2_ x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = f2_{(float)threadIdx.x + i, (float)i};
  const f2_ A = {a, a}, B = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(A), "v"(B));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_pk_mul_f32(float *out, int iters, float a, float b) {
  f2_ x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = f2_{(float)threadIdx.x + i, (float)i};
  const f2_ A = {a, a};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(A));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_pk_add_f32(float *out, int iters, float a, float b) {
  f2_ x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = f2_{(float)threadIdx.x + i, (float)i};
  const f2_ A = {a, a};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(A));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(float *, int, float, float);
struct Entry {
  const char *name;
  kern_t k;
};

int main() {
  const int blocks = 256 * 8 * 2, iters = 4000;  // 8 resident blocks of 4 waves per CU = 8 waves per SIMD, two rounds
  float *d;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double clock_hz = prop.clockRate * 1e3;  // the runtime's engine clock (kHz); the real one may sit below it
  const Entry list[] = {
      {"v_fma_f32", k_fma_f32}, {"v_fmac_f32", k_fmac_f32}, {"v_mul_f32", k_mul_f32}, {"v_add_f32", k_add_f32},
      {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32},
      {"v_rcp_f32", k_rcp_f32}, {"v_fract_f32", k_fract_f32}, {"v_cvt_i32_f32", k_cvt_i32_f32},
      {"v_cvt_f32_ubyte0", k_cvt_f32_ubyte0}, {"v_cvt_f32_ubyte2", k_cvt_f32_ubyte2}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8_f32},
      {"v_cmp_gt_f32", k_cmp_gt_f32}, {"v_cmp_class_f32", k_cmp_class_f32}, {"v_div_fixup_f32", k_div_fixup_f32},
      {"v_mov_b32", k_mov_b32}, {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32}, {"v_lshl_or_b32", k_lshl_or_b32},
      {"v_bfe_u32", k_bfe_u32}, {"v_cndmask_b32", k_cndmask_b32}, {"v_min_u32", k_min_u32}, {"v_min3_u32", k_min3_u32},
      {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_lo_u32", k_mul_lo_u32},
      {"v_cmp_gt_u32", k_cmp_gt_u32}, {"v_cmp_eq_u16_sdwa", k_cmp_eq_u16_sdwa}, {"v_readlane_b32", k_readlane},
      {"v_writelane_b32", k_writelane}, {"v_cndmask_b32_e64 (sgpr pair mask)", k_cndmask_e64_sgpr}, {"v_cndmask_b32 (dst not a source)", k_cndmask_other_dst},
      {"v_cmp_gt_u32 vcc + v_cndmask vcc (pair)", k_cmp_then_cndmask}, {"v_cmp_e64 sgpr + v_cndmask_e64 (pair)", k_cmp_sgpr_then_cndmask},
      {"v_cmp vcc + 4 x v_cndmask vcc (five instr)", k_cmp_then_4cndmask}, {"v_cmp sgpr + 4 x v_cndmask_e64 (five instr)", k_cmp_sgpr_then_4cndmask},
      {"s_mov vcc + v_cndmask vcc (pair)", k_salu_vcc_then_cndmask}, {"v_cmp vcc; cndmask; v_add_u32; cndmask (4 instr)", k_cmp_cnd_fma_cnd}, {"v_cmp vcc; v_add_u32; cndmask (3 instr)", k_cmp_add_cnd},
      {"v_cmp vcc; 2 x v_add_u32; cndmask (4 instr)", k_cmp_2add_cnd}, {"v_cmp vcc; s_and vcc; cndmask vcc (3 instr)", k_cmp_sand_cnd},
      {"v_cmp_e64 sgpr; s_and sgpr; cndmask_e64 (3 instr)", k_cmp64_sand_cnd64}, {"v_cmp vcc; cndmask -1, v, vcc (pair)", k_cmp_cnd_constsrc},
      {"v_cmp vcc; v_addc_co_u32 (pair)", k_addc_vcc}, {"v_bfi_b32", k_bfi_b32}, {"v_perm_b32", k_perm_b32}, {"v_or3_b32", k_or3_b32}, {"v_and_or_b32", k_and_or_b32}, {"v_lshlrev_b32", k_lshlrev_b32},
      {"v_xor_b32", k_xor_b32}, {"v_sub_u32", k_sub_u32}, {"v_add3_u32", k_add3_u32}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_max_f32", k_max_f32},
      {"v_med3_f32", k_med3_f32}, {"v_mul_legacy_f32", k_mul_legacy}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64},
      {"v_rcp_f64", k_rcp_f64}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double wave_instr_per_simd = (double)blocks * 4 /*waves*/ * iters * 8 / (256.0 * 4);
  printf("# engine clock reported: %.0f MHz; cycles = ms * clock / (wave instructions per SIMD = %.0f)\n", clock_hz * 1e-6, wave_instr_per_simd);
  double base = 0;
  for (const Entry &e : list) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * clock_hz / wave_instr_per_simd;
    if (!base) base = ms;
    printf("%-52s %8.3f ms  %6.2f cycles per wave instruction  %5.2f x v_fma_f32\n", e.name, ms, cyc, ms / base);
  }
  return 0;
}
