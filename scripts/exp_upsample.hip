// Development microbenchmark: what bounds the TOD streaming write?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imaria_amd/csrc scripts/exp_upsample.hip -o build/exp_upsample
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../maria_amd/csrc/mrx_spline.hip"

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e = (x);                                              \
    if (e != hipSuccess) {                                           \
      printf("%s: %s\n", #x, hipGetErrorString(e));                  \
      exit(1);                                                       \
    }                                                                \
  } while (0)

typedef float vfloat4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void fill_stream(float* out, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  const vfloat4 v = {1.f, 2.f, 3.f, 4.f};
  for (; i < n4; i += stride) {
    if (NT)
      __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(out) + i);
    else
      reinterpret_cast<vfloat4*>(out)[i] = v;
  }
}

// tile pattern of the upsample kernel, constant payload
template <bool NT, int TILE_DET>
__global__ __launch_bounds__(256) void fill_tiles(float* out, int D, int T, size_t ld) {
  const int sb = blockIdx.x * 1024 + threadIdx.x * 4;
  const int d0 = blockIdx.y * TILE_DET;
  if (sb + 4 > T) return;
  const vfloat4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  const int nd = min(TILE_DET, D - d0);
  for (int dl = 0; dl < nd; ++dl) {
    float* dst = out + (size_t)(d0 + dl) * ld + sb;
    if (NT)
      __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(dst));
    else
      *reinterpret_cast<vfloat4*>(dst) = v;
  }
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

int main() {
  const int D = 10000, T = 240000, Ta = 6000;
  const size_t n = (size_t)D * T;
  float* out;
  CK(hipMalloc(&out, n * 4));
  float2* ym;
  CK(hipMalloc(&ym, (size_t)Ta * D * 8));
  CK(hipMemset(ym, 0, (size_t)Ta * D * 8));
  double* t;
  CK(hipMalloc(&t, T * 8));
  std::vector<double> ht(T);
  for (int i = 0; i < T; ++i) ht[i] = i / 400.0;
  CK(hipMemcpy(t, ht.data(), T * 8, hipMemcpyHostToDevice));
  const double gb = n * 4 / 1e9;

  auto report = [&](const char* name, float ms) { printf("%-34s %7.3f ms  %7.1f GB/s\n", name, ms, gb / ms * 1e3); };

  for (int blocks : {2048, 8192, 65536})
    for (int nt = 0; nt < 2; ++nt) {
      char nm[64];
      snprintf(nm, 64, "stream fill %s grid=%d", nt ? "nt" : "st", blocks);
      report(nm, time_ms([&] {
               if (nt)
                 hipLaunchKernelGGL(fill_stream<true>, dim3(blocks), dim3(256), 0, 0, out, n / 4);
               else
                 hipLaunchKernelGGL(fill_stream<false>, dim3(blocks), dim3(256), 0, 0, out, n / 4);
             }));
    }
  {
    dim3 g16((T + 1023) / 1024, (D + 15) / 16), g4((T + 1023) / 1024, (D + 3) / 4), g64((T + 1023) / 1024, (D + 63) / 64);
    report("tiles 16det st", time_ms([&] { hipLaunchKernelGGL((fill_tiles<false, 16>), g16, dim3(256), 0, 0, out, D, T, (size_t)T); }));
    report("tiles 16det nt", time_ms([&] { hipLaunchKernelGGL((fill_tiles<true, 16>), g16, dim3(256), 0, 0, out, D, T, (size_t)T); }));
    report("tiles 4det st", time_ms([&] { hipLaunchKernelGGL((fill_tiles<false, 4>), g4, dim3(256), 0, 0, out, D, T, (size_t)T); }));
    report("tiles 4det nt", time_ms([&] { hipLaunchKernelGGL((fill_tiles<true, 4>), g4, dim3(256), 0, 0, out, D, T, (size_t)T); }));
    report("tiles 64det st", time_ms([&] { hipLaunchKernelGGL((fill_tiles<false, 64>), g64, dim3(256), 0, 0, out, D, T, (size_t)T); }));
    report("tiles 64det nt", time_ms([&] { hipLaunchKernelGGL((fill_tiles<true, 64>), g64, dim3(256), 0, 0, out, D, T, (size_t)T); }));
  }
  {
    dim3 grid((T + kTileSamples - 1) / kTileSamples, (D + kTileDet - 1) / kTileDet);
    report("library upsample kernel K64", time_ms([&] {
             hipLaunchKernelGGL((spline_upsample_kernel<false, 64>), grid, dim3(kBlock), 0, 0, ym, D, Ta, 0.0, 10.0, t, T,
                                (const float*)nullptr, (const int32_t*)nullptr, out, (size_t)T, 1, 1);
           }));
    report("library upsample kernel K256", time_ms([&] {
             hipLaunchKernelGGL((spline_upsample_kernel<false, 256>), grid, dim3(kBlock), 0, 0, ym, D, Ta, 0.0, 10.0, t, T,
                                (const float*)nullptr, (const int32_t*)nullptr, out, (size_t)T, 1, 1);
           }));
  }
  return 0;
}
