// Development microbenchmark: HBM write rate of a [D][T] float32 array as a function of WHICH addresses the chip has
// in flight together: rows per workgroup x contiguous samples per row, and the order of the workgroups.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp_store_patterns.hip -o scripts/exp_store_patterns.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                               \
  do {                                                      \
    hipError_t e = (x);                                     \
    if (e != hipSuccess) {                                  \
      printf("%s: %s\n", #x, hipGetErrorString(e));         \
      exit(1);                                              \
    }                                                       \
  } while (0)

typedef float vfloat4 __attribute__((ext_vector_type(4)));

// a workgroup writes `rows` rows x `span` samples (span a multiple of 1024: 256 threads x 16 B per pass);
// tiles are numbered along the time axis first (order 0) or along the detector axis first (order 1)
__global__ __launch_bounds__(256) void fill(float* out, int D, int T, size_t ld, int rows, int span, int tiles_t, int tiles_d, int order,
                                            int nt) {
  extern __shared__ float occupancy_pad[];  // dynamic LDS only caps the workgroups per CU
  const int id = blockIdx.x;
  const int tt = order == 0 ? id % tiles_t : id / tiles_d, td = order == 0 ? id / tiles_t : id % tiles_d;
  const vfloat4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  const int d0 = td * rows, s0 = tt * span;
  for (int dl = 0; dl < rows && d0 + dl < D; ++dl)
    for (int s = s0 + threadIdx.x * 4; s < s0 + span && s + 4 <= T; s += 1024) {
      vfloat4* dst = reinterpret_cast<vfloat4*>(out + (size_t)(d0 + dl) * ld + s);
      if (nt) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
}

// the same bytes as one linear sweep: workgroup b writes the b-th chunk of `chunk` floats
__global__ __launch_bounds__(256) void fill_linear(float* out, size_t n, int chunk) {
  const vfloat4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  const size_t base = (size_t)blockIdx.x * chunk;
  for (size_t i = base + threadIdx.x * 4; i < base + chunk && i + 4 <= n; i += 1024) *reinterpret_cast<vfloat4*>(out + i) = v;
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

int main(int argc, char** argv) {
  const int D = 10000, T = 240000;
  float* out;
  CK(hipMalloc(&out, (size_t)D * T * 4));
  const double gb = 4.0 * D * T / 1e9;
  for (int chunk : {4096, 16384, 65536, 262144}) {
    const size_t n = (size_t)D * T;
    const int blocks = (int)((n + chunk - 1) / chunk);
    const float ms = time_ms([&] { hipLaunchKernelGGL(fill_linear, dim3(blocks), dim3(256), 0, 0, out, n, chunk); });
    printf("linear sweep, %6d floats per workgroup: %.3f ms %.0f GB/s\n", chunk, ms, gb / ms * 1e3);
  }
  const int shapes[][2] = {{32, 1024}, {16, 1024}, {8, 1024}, {4, 1024}, {1, 1024}, {16, 2048}, {8, 4096}, {4, 8192}, {2, 16384}, {1, 32768}, {1, 8192},
                           {32, 4096}, {16, 8192}, {8, 2048}};
  for (auto& sh : shapes) {
    const int rows = sh[0], span = sh[1];
    const int tiles_t = (T + span - 1) / span, tiles_d = (D + rows - 1) / rows;
    for (int order = 0; order < 2; ++order) {
      const float ms = time_ms([&] { hipLaunchKernelGGL(fill, dim3(tiles_t * tiles_d), dim3(256), 0, 0, out, D, T, (size_t)T, rows, span, tiles_t, tiles_d, order, 1); });
      printf("%2d rows x %5d samples, %s first: %.3f ms %.0f GB/s\n", rows, span, order == 0 ? "time" : "detector", ms, gb / ms * 1e3);
    }
  }
  // the writer's tile (32 rows x 1024 samples, time first) under an occupancy cap
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fill), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int rows : {32, 16}) {
    const int span = 1024, tiles_t = (T + span - 1) / span, tiles_d = (D + rows - 1) / rows;
    for (int kb : {0, 20, 30, 40, 53, 64, 80, 150}) {
      const float ms = time_ms([&] { hipLaunchKernelGGL(fill, dim3(tiles_t * tiles_d), dim3(256), (size_t)kb * 1024, 0, out, D, T, (size_t)T, rows, span, tiles_t, tiles_d, 0, 1); });
      printf("%2d rows x 1024 samples, time first, %3d KB LDS per workgroup (%s per CU): %.3f ms %.0f GB/s\n", rows, kb,
             kb == 0 ? "8" : kb <= 20 ? "8" : kb <= 30 ? "5" : kb <= 40 ? "4" : kb <= 53 ? "3" : kb <= 80 ? "2" : "1", ms, gb / ms * 1e3);
    }
  }
  return 0;
}
