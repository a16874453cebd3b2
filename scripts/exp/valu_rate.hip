// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_fma_f64 on gfx950 with all SIMDs
// full (8 waves per SIMD), independent accumulators.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[8]; float2v p[8]; double d[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; p[i] = float2v{x[i], x[i] + 1}; d[i] = x[i]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
      if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], float2v{a, a}, float2v{b, b});
      if (MODE == 2) d[i] = __builtin_fma(d[i], (double)a, (double)b);
      if (MODE == 3) x[i] = x[i] * a;  // v_mul_f32
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y + (float)d[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8192 * 4);
  const int iters = 4096, blocks = 256 * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_mul_f32"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) k<0><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
      if (mode == 1) k<1><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
      if (mode == 2) k<2><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
      if (mode == 3) k<3><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // wave-instructions per SIMD = 8 waves * iters * 8
      const double instr_per_simd = 8.0 * iters * 8;
      if (rep) printf("%-14s %.3f ms -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[mode], ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
  }
  return 0;
}
