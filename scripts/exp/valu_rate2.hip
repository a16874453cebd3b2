// Issue rates of the conversion / float64 instructions the sampler could use (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, double b) {
  float x[8]; double d[8]; int n[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.37f + i; d[i] = x[i] * 1.7; n[i] = threadIdx.x + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, 0.5f);
      if (MODE == 1) { n[i] = __double2int_rz(d[i]); d[i] = d[i] + b; }              // cvt_i32_f64 + add_f64
      if (MODE == 2) d[i] = d[i] + b;                                                 // add_f64 alone
      if (MODE == 3) { d[i] = (double)n[i]; n[i] += (int)a; }                          // cvt_f64_i32 (+ int add)
      if (MODE == 4) { x[i] = (float)d[i]; d[i] = d[i] + b; }                          // cvt_f32_f64 + add_f64
      if (MODE == 5) { d[i] = (double)x[i]; x[i] = x[i] * a; }                         // cvt_f64_f32 + mul_f32
      if (MODE == 6) x[i] = __builtin_amdgcn_rcpf(x[i]);                               // v_rcp_f32
      if (MODE == 7) d[i] = __builtin_amdgcn_fract(d[i]) + b;                          // fract_f64 + add_f64
      if (MODE == 8) { n[i] = (int)x[i]; x[i] = x[i] * a; }                            // cvt_i32_f32 + mul
      if (MODE == 9) d[i] = __builtin_floor(d[i]) * b;                                 // floor_f64 + mul_f64
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + (float)d[i] + n[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int M> float run(float* out, int iters, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); k<M><<<blocks, 256>>>(out, iters, 1.0001f, 0.25); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 8192 * 4);
  const int iters = 2048, blocks = 256 * 8;
  const char* names[10] = {"fma_f32", "cvt_i32_f64+add_f64", "add_f64", "cvt_f64_i32+iadd", "cvt_f32_f64+add_f64", "cvt_f64_f32+mul_f32", "rcp_f32", "fract_f64+add_f64", "cvt_i32_f32+mul", "floor_f64+mul_f64"};
  float ms[10] = {run<0>(out, iters, blocks), run<1>(out, iters, blocks), run<2>(out, iters, blocks), run<3>(out, iters, blocks), run<4>(out, iters, blocks),
                  run<5>(out, iters, blocks), run<6>(out, iters, blocks), run<7>(out, iters, blocks), run<8>(out, iters, blocks), run<9>(out, iters, blocks)};
  for (int m = 0; m < 10; ++m) printf("%-22s %.3f ms = %.2f x fma_f32\n", names[m], ms[m], ms[m] / ms[0]);
  return 0;
}
