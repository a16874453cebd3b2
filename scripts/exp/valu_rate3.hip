// Issue rates of the integer / transcendental instructions the noise generator uses (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, uint32_t mm) {
  float x[8]; uint32_t n[8], h[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.37f + i + 1.5f; n[i] = threadIdx.x * 2654435761u + i; h[i] = n[i] ^ 0x9e3779b9u; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, 0.5f);
      if (MODE == 1) n[i] = n[i] * mm;                                              // v_mul_lo_u32
      if (MODE == 2) n[i] = __umulhi(n[i], mm);                                     // v_mul_hi_u32
      if (MODE == 3) { const uint64_t p = (uint64_t)n[i] * mm; n[i] = (uint32_t)p ^ (uint32_t)(p >> 32); }  // mad_u64_u32 + xor
      if (MODE == 4) n[i] = __builtin_amdgcn_alignbit(n[i], n[i], 13) ^ h[i];       // rotate + xor
      if (MODE == 5) n[i] = (n[i] + h[i]) ^ mm;                                      // add + xor (xad?)
      if (MODE == 6) x[i] = __builtin_amdgcn_logf(x[i]) + 3.0f;                      // v_log_f32 + add
      if (MODE == 7) x[i] = __builtin_amdgcn_sinf(x[i]);                             // v_sin_f32
      if (MODE == 8) x[i] = __builtin_amdgcn_sqrtf(x[i]) + 1.0f;                     // v_sqrt_f32 + add
      if (MODE == 9) n[i] = __umul24(n[i], mm) + h[i];               // v_mad_u32_u24
      if (MODE == 10) x[i] = __builtin_amdgcn_rsqf(x[i]) + 1.0f;                     // v_rsq_f32 + add
      if (MODE == 11) x[i] = (float)(n[i] >> 8) * a, n[i] += 77u;                     // shift + cvt_f32_u32 + mul + add
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + n[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int M> float run(float* out, int iters, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); k<M><<<blocks, 256>>>(out, iters, 1.0001f, 0xD2511F53u); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 8192 * 4);
  const int iters = 2048, blocks = 256 * 8;
  const char* names[12] = {"fma_f32", "mul_lo_u32", "mul_hi_u32", "mad_u64_u32+xor", "alignbit+xor", "add+xor", "log+add", "sin", "sqrt+add", "mul_u24+add", "rsq+add", "lshr+cvt+mul+add"};
  float ms[12] = {run<0>(out, iters, blocks), run<1>(out, iters, blocks), run<2>(out, iters, blocks), run<3>(out, iters, blocks), run<4>(out, iters, blocks),
                  run<5>(out, iters, blocks), run<6>(out, iters, blocks), run<7>(out, iters, blocks), run<8>(out, iters, blocks), run<9>(out, iters, blocks), run<10>(out, iters, blocks), run<11>(out, iters, blocks)};
  for (int m = 0; m < 12; ++m) printf("%-22s %.3f ms = %.2f x fma_f32\n", names[m], ms[m], ms[m] / ms[0]);
  return 0;
}
