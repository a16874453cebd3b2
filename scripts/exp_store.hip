// Development microbenchmark: HBM write rate of 16-byte stores under each cache policy
// (gfx950 store modifiers sc0 / sc1 / nt), in the TOD writer's tile pattern.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp_store.hip -o scripts/exp_store.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e = (x);                                              \
    if (e != hipSuccess) {                                           \
      printf("%s: %s\n", #x, hipGetErrorString(e));                  \
      exit(1);                                                       \
    }                                                                \
  } while (0)

typedef float vfloat4 __attribute__((ext_vector_type(4)));

template <int P>
__device__ __forceinline__ void store16(float* dst, vfloat4 v) {
  if (P == 0) *reinterpret_cast<vfloat4*>(dst) = v;
  if (P == 1) __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(dst));
  if (P == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
  if (P == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(v) : "memory");
  if (P == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
  if (P == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(dst), "v"(v) : "memory");
  if (P == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(dst), "v"(v) : "memory");
  if (P == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(dst), "v"(v) : "memory");
}

// tile pattern of the TOD writer: block = 1024 samples x TILE rows, 16 B per thread per row
template <int P, int TILE>
__global__ __launch_bounds__(256) void fill_tiles(float* out, int D, int T, size_t ld) {
  const int sb = blockIdx.x * 1024 + threadIdx.x * 4;
  const int d0 = blockIdx.y * TILE;
  if (sb + 4 > T) return;
  const vfloat4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  const int nd = min(TILE, D - d0);
  for (int dl = 0; dl < nd; ++dl) store16<P>(out + (size_t)(d0 + dl) * ld + sb, v);
}

template <typename F>
float time_ms(F f, int reps = 7) {
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

template <int P>
void run(float* out, int D, int T, const char* name) {
  const dim3 g16((T + 1023) / 1024, (D + 15) / 16), g32((T + 1023) / 1024, (D + 31) / 32);
  const float a = time_ms([&] { hipLaunchKernelGGL((fill_tiles<P, 16>), g16, dim3(256), 0, 0, out, D, T, (size_t)T); });
  const float b = time_ms([&] { hipLaunchKernelGGL((fill_tiles<P, 32>), g32, dim3(256), 0, 0, out, D, T, (size_t)T); });
  const double gb = 4.0 * D * T / 1e9;
  printf("%-12s tile16 %.3f ms %.0f GB/s | tile32 %.3f ms %.0f GB/s\n", name, a, gb / a * 1e3, b, gb / b * 1e3);
}

int main() {
  const int D = 10000, T = 240000;
  float* out;
  CK(hipMalloc(&out, (size_t)D * T * 4));
  run<0>(out, D, T, "plain");
  run<1>(out, D, T, "nt");
  run<2>(out, D, T, "sc0 sc1");
  run<3>(out, D, T, "sc0 sc1 nt");
  run<4>(out, D, T, "sc1");
  run<5>(out, D, T, "sc1 nt");
  run<6>(out, D, T, "sc0");
  run<7>(out, D, T, "sc0 nt");
  return 0;
}
