// Development microbenchmark: what separates a plain linear fill (6.0-6.2 TB/s here) from torch.fill_ / hipMemset (6.5)?
// Workgroup size, bytes per thread, workgroups per CU (LDS cap), grid-stride vs one chunk per workgroup.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp_store_shapes.hip -o scripts/exp_store_shapes.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                       \
  do {                                              \
    hipError_t e = (x);                             \
    if (e != hipSuccess) {                          \
      printf("%s: %s\n", #x, hipGetErrorString(e)); \
      exit(1);                                      \
    }                                               \
  } while (0)

typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef float vfloat2 __attribute__((ext_vector_type(2)));

// every thread writes kPer x 16 bytes: thread-contiguous (kContig) or interleaved by the workgroup
template <int kThreads, int kPer, bool kContig>
__global__ __launch_bounds__(kThreads) void fill16(float* out, size_t n) {
  extern __shared__ float pad[];
  const vfloat4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  const size_t base = (size_t)blockIdx.x * kThreads * kPer * 4;
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const size_t at = base + (kContig ? ((size_t)threadIdx.x * kPer + i) * 4 : ((size_t)i * kThreads + threadIdx.x) * 4);
    if (at + 4 <= n) *reinterpret_cast<vfloat4*>(out + at) = v;
  }
}

// grid-stride: a resident grid sweeps the array together
__global__ __launch_bounds__(256) void fill_stride(float* out, size_t n) {
  const vfloat4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (size_t at = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; at + 4 <= n; at += (size_t)gridDim.x * 1024) *reinterpret_cast<vfloat4*>(out + at) = v;
}

template <typename F>
float time_ms(F f, int reps = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a));
    f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
  }
  return best;
}

template <int kThreads, int kPer, bool kContig>
void run(float* out, size_t n, size_t lds, const char* note) {
  const int blocks = (int)((n / 4 + (size_t)kThreads * kPer - 1) / ((size_t)kThreads * kPer));
  if (lds > 48 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill16<kThreads, kPer, kContig>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const float ms = time_ms([&] { hipLaunchKernelGGL((fill16<kThreads, kPer, kContig>), dim3(blocks), dim3(kThreads), lds, 0, out, n); });
  printf("%4d threads x %2d x 16 B %s, %3zu KB LDS%s: %.3f ms %.0f GB/s\n", kThreads, kPer, kContig ? "thread-contiguous" : "interleaved      ", lds / 1024, note,
         ms, 4.0 * n / ms / 1e6);
}

int main() {
  const size_t n = (size_t)10000 * 240000;
  float* out;
  CK(hipMalloc(&out, n * 4));
  CK(hipMemset(out, 0, n * 4));
  {
    const float ms = time_ms([&] { CK(hipMemsetAsync(out, 0, n * 4, 0)); });
    printf("hipMemsetAsync: %.3f ms %.0f GB/s\n", ms, 4.0 * n / ms / 1e6);
  }
  run<64, 1, false>(out, n, 0, "");
  run<128, 1, false>(out, n, 0, "");
  run<256, 1, false>(out, n, 0, "");
  run<1024, 1, false>(out, n, 0, "");
  run<128, 4, false>(out, n, 0, "");
  run<128, 4, true>(out, n, 0, "");
  run<256, 4, false>(out, n, 0, "");
  run<256, 4, true>(out, n, 0, "");
  run<256, 16, false>(out, n, 0, "");
  run<256, 32, false>(out, n, 0, "");
  run<256, 4, false>(out, n, 30 * 1024, " (5 workgroups per CU)");
  run<256, 4, false>(out, n, 64 * 1024, " (2 per CU)");
  run<256, 4, false>(out, n, 150 * 1024, " (1 per CU)");
  run<256, 32, false>(out, n, 64 * 1024, " (2 per CU)");
  for (int per_cu : {1, 2, 4, 8, 16}) {
    const float ms = time_ms([&] { hipLaunchKernelGGL(fill_stride, dim3(256 * per_cu), dim3(256), 0, 0, out, n); });
    printf("grid-stride, %2d workgroups per CU: %.3f ms %.0f GB/s\n", per_cu, ms, 4.0 * n / ms / 1e6);
  }
  return 0;
}
