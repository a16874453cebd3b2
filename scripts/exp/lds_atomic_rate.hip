// LDS atomic add rates on gfx950: f64 / f32 / u64 / u32, addresses conflict-free, random, and pairs of lanes on one address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ T acc[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) acc[i] = T(0);
  __syncthreads();
  uint32_t h = threadIdx.x * 2654435761u + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    int idx;
    if (MODE == 0) idx = (threadIdx.x + it * 256) & 4095;            // conflict-free, consecutive
    else if (MODE == 1) { h = h * 1664525u + 1013904223u; idx = (h >> 12) & 4095; }  // random
    else idx = (((threadIdx.x >> 1) * 37) + it * 64) & 4095;         // two lanes per address
    atomicAdd(&acc[idx], T(1));
  }
  __syncthreads();
  float s = 0;
  for (int i = threadIdx.x; i < 4096; i += 256) s += (float)acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename T, int M> float run(float* out, int iters, int blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0); k<T, M><<<blocks, 256>>>(out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 4096 * 4);
  const int iters = 4096, blocks = 256 * 4;
  const double n = (double)iters * blocks * 256;
  const char* modes[3] = {"consecutive", "random", "pairs"};
#define ROW(T, name) { float a = run<T,0>(out, iters, blocks), b = run<T,1>(out, iters, blocks), c = run<T,2>(out, iters, blocks); \
  printf("%-4s %s %.3f ms (%.1f G/s)  %s %.3f ms (%.1f G/s)  %s %.3f ms (%.1f G/s)\n", name, modes[0], a, n / a / 1e6, modes[1], b, n / b / 1e6, modes[2], c, n / c / 1e6); }
  ROW(double, "f64") ROW(float, "f32") ROW(unsigned long long, "u64") ROW(unsigned int, "u32")
  return 0;
}
