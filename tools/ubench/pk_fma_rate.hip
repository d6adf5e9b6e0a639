// Micro-benchmark: issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950 (one wave64 instruction = 64 or 128 fmas).
// Why: DESIGN.md 3.1 reads "the packed colour update saves 12 vector instructions per wave-row and buys nothing" as "the
// kernel is not gated by vector issue"; that inference needs v_pk_fma_f32 to cost ONE issue slot, like v_fma_f32.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_fma_rate.hip -o /tmp/pk_fma_rate ; run: /tmp/pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
  float2_ x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = float2_{(float)threadIdx.x + i, (float)i};
  const float2_ A = {a, a}, B = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (PK) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(A), "v"(B));
      } else {
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i].x) : "v"(x[i].x), "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i].y) : "v"(x[i].y), "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int PK>
static double run(float *d, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<PK>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<PK>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const int blocks = 256 * 8 * 4, iters = 4000;  // 8 blocks of 4 waves per CU resident = 8 waves per SIMD, 4 rounds
  float *d;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  const double fmas = (double)blocks * 256 * iters * 16;  // per kernel: 16 fmas per thread and iteration, either way
  const double ms_s = run<0>(d, blocks, iters), ms_p = run<1>(d, blocks, iters);
  printf("{\"scalar_v_fma_f32\": {\"ms\": %.3f, \"Tfma_per_s\": %.2f, \"wave_instructions\": %.3e}, "
         "\"packed_v_pk_fma_f32\": {\"ms\": %.3f, \"Tfma_per_s\": %.2f, \"wave_instructions\": %.3e}, \"packed_speedup\": %.3f}\n",
         ms_s, fmas / ms_s * 1e-9, fmas / 64, ms_p, fmas / ms_p * 1e-9, fmas / 128, ms_s / ms_p);
  return 0;
}
