// Store-only microbenchmark: how fast does the chip write a [D][T] float32 array (D rows of T samples, the TOD's layout)
// when a workgroup's tile is R rows x S samples, a wave storing 1 KiB (16 bytes a lane) of one row per instruction?
// A resident grid of W workgroups per CU takes tiles from a queue, time tile fastest or row group fastest.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/exp/store_shapes.bin scripts/exp/store_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef float vf4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// tile: R rows x S samples (S a multiple of 1024); thread t of 256 stores samples [p * 1024 + 4 t, +4) of every row for
// pass p < S / 1024.  order: 0 = rows inner (all rows of pass p, then pass p + 1), 1 = passes inner (row r: all passes)
__global__ __launch_bounds__(256) void store_tiles(float* out, size_t ld, int D, int T, int R, int S, int order, int time_first,
                                                   int* queue, int n_tiles, int nsx, int nrg, int lds_pad) {
  extern __shared__ int pad[];
  __shared__ int s_next;
  if (lds_pad && threadIdx.x == 0) pad[0] = 0;
  const vf4 v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
  const int passes = S / 1024;
  for (;;) {
    if (threadIdx.x == 0) s_next = atomicAdd(queue, 1);
    __syncthreads();
    const int tile = s_next;
    __syncthreads();
    if (tile >= n_tiles) break;
    int sx, rg;
    if (time_first < 0) {  // blocks of -time_first row groups: block by block, inside a block time tile by time tile (row group fastest)
      const int NB = -time_first, per_block = NB * nsx;
      const int blk = tile / per_block, r = tile - blk * per_block;
      const int nb = min(NB, nrg - blk * NB);  // (the last block may be shorter)
      sx = r / nb;
      rg = blk * NB + (r - sx * nb);
      if (sx >= nsx) continue;
    } else if (time_first > 1) {  // bands of `time_first` time tiles: within a band, row group by row group, the band's time tiles fastest
      const int B = time_first, per_band = B * nrg;
      const int band = tile / per_band, r = tile - band * per_band;
      const int nb = min(B, nsx - band * B);  // (the last band may be shorter)
      rg = r / nb;
      sx = band * B + (r - rg * nb);
      if (rg >= nrg) continue;
    } else {
      sx = time_first ? tile % nsx : tile / nrg;
      rg = time_first ? tile / nsx : tile % nrg;
    }
    float* base = out + (size_t)rg * R * ld + (size_t)sx * S + threadIdx.x * 4;
    if (order == 0) {
      for (int p = 0; p < passes; ++p)
        for (int r = 0; r < R; ++r)
          if (rg * R + r < D && sx * S + p * 1024 + threadIdx.x * 4 + 4 <= T)
            __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(base + (size_t)r * ld + p * 1024));
    } else {
      for (int r = 0; r < R; ++r)
        for (int p = 0; p < passes; ++p)
          if (rg * R + r < D && sx * S + p * 1024 + threadIdx.x * 4 + 4 <= T)
            __builtin_nontemporal_store(v, reinterpret_cast<vf4*>(base + (size_t)r * ld + p * 1024));
    }
  }
}

int main(int argc, char** argv) {
  const int D = 10000, T = 240000;
  const size_t ld = T;
  float* out;
  CK(hipMalloc(&out, (size_t)D * ld * 4));
  int* queue;
  CK(hipMalloc(&queue, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int shapes[][2] = {{32, 1024}};
  for (auto& sh : shapes) {
    const int R = sh[0], S = sh[1];
    const int nsx = (T + S - 1) / S, nrg = (D + R - 1) / R;
    const int n_tiles = nsx * nrg;
    for (int wpc : {5}) {
      for (int time_first : {0, 1, -2, -4, -8, -16, -32, -64, -128}) {
        for (int order : {0, 1}) {
          if (S == 1024 && order == 1) continue;
          float best = 1e9f;
          for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemsetAsync(queue, 0, 4, 0));
            CK(hipEventRecord(e0, 0));
            const int nt = time_first < 0 ? ((nrg - time_first - 1) / -time_first) * -time_first * nsx : time_first > 1 ? ((nsx + time_first - 1) / time_first) * time_first * nrg : n_tiles;
            hipLaunchKernelGGL(store_tiles, dim3(256 * wpc), dim3(256), 0, 0, out, ld, D, T, R, S, order, time_first, queue, nt, nsx, nrg, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
          }
          printf("tile %2d rows x %5d samples (%3d KiB/row)  wgs/CU %d  %s  %s: %.3f ms = %.2f TB/s\n", R, S, S * 4 / 1024, wpc,
                 time_first < 0 ? (std::string("blocks of ") + std::to_string(-time_first) + " row groups, time-major inside").c_str() : time_first > 1 ? (std::string("bands of ") + std::to_string(time_first) + " time tiles").c_str() : time_first ? "time tile fastest" : "row group fastest", order ? "row: all passes" : "pass: all rows ", best,
                 (double)D * T * 4 / best / 1e9);
          fflush(stdout);
        }
      }
    }
  }
  return 0;
}
