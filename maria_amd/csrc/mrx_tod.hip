// TOD pre-processing for the mappers (tod/processing.py:91-204): the streaming passes of
// process_tod on a [D][T] float32 TOD in place --
//   remove_slope   D -= linspace(D[:, 0], D[:, -1], T)                (processing.py:99-105)
//   window         D *= w[t]                                          (processing.py:139-146)
//   filter         remove_slope, then scipy.signal.sosfilt of the Bessel low / high pass
//                  sections along time                                (processing.py:148-176,
//                                                                      utils/signal/filters.py:46-69)
// The GEMM-shaped operations (remove_spline, remove_modes) are plain library products and
// stay with the host side (maria_amd/tod_processing.py, torch.matmul / eigh).
//
// sosfilt is a recursion along time; here it is made time-parallel in the standard way for
// a linear recurrence s[n+1] = A s[n] + B x[n]: (1) every chunk of 256 samples is run from a
// zero state and leaves its end state; (2) one thread per detector chains the chunks,
// s_in(c+1) = A^256 s_in(c) + s_zero(c) (A^256 from the host, float64); (3) every chunk is
// run again from its true initial state and writes the output.  Arithmetic is float64 in the
// transposed direct form II of scipy's _sosfilt; the result differs from the serial loop by
// float64 rounding only.  Lanes are consecutive chunks of one detector: each lane streams
// its own 1 KiB with 16-byte loads.
#include "mrx_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 256;      // samples per chunk
constexpr int kMaxSections = 8;  // biquads in the cascade (low + high pass up to order 3)

typedef float vfloat4 __attribute__((ext_vector_type(4)));

struct SosArgs {
  double b0[kMaxSections], b1[kMaxSections], b2[kMaxSections], a1[kMaxSections], a2[kMaxSections];
  int n_sections;
  const float* in;
  size_t ld_in;
  float* out;
  size_t ld_out;
  int D, T, n_chunks;
  int remove_slope;   // subtract the line through the first and last sample first
  const double* anchors;  // [D][2] (first, last) of the input rows
  double* states;     // [D][n_chunks][2 * n_sections]
  const double* M;    // [2S][2S] = A^kChunk, row-major (state' = M state)
};

// np.linspace(a, b, T)[t] = a + t (b - a)/(T - 1), with the last point set to b exactly
__device__ __forceinline__ double line_at(double first, double last, double step, int t, int T) {
  return t == T - 1 ? last : first + step * (double)t;
}

// (first, last) of every row, read before anything is overwritten: anchors[2 d], [2 d + 1]
__global__ __launch_bounds__(kBlock) void anchors_kernel(const float* __restrict__ data, size_t ld, int D, int T,
                                                       double* __restrict__ anchors) {
  const int d = blockIdx.x * kBlock + threadIdx.x;
  if (d >= D) return;
  anchors[2 * d] = (double)data[(size_t)d * ld];
  anchors[2 * d + 1] = (double)data[(size_t)d * ld + T - 1];
}

// One workgroup = 256 consecutive chunks of one detector (65536 samples), one thread per chunk.
// A thread walks its chunk sequentially, but the workgroup moves the data 32 samples of every
// chunk at a time through LDS: global loads and stores are whole 128-byte lines (8 lanes x 16
// bytes per chunk), and a thread reads its own LDS row (pitch 33 words: conflict-free).
constexpr int kSub = 32;            // samples of every chunk per stage
constexpr int kPitch = kSub + 1;

template <int S, bool kWrite>
__global__ __launch_bounds__(kBlock) void sos_chunk_kernel(SosArgs g) {
  __shared__ float stage[kBlock * kPitch];
  const int c0 = blockIdx.x * kBlock;              // first chunk of the workgroup
  const int c = c0 + threadIdx.x;                  // this thread's chunk
  const int d = blockIdx.y;
  const float* row = g.in + (size_t)d * g.ld_in;
  float* orow = g.out + (size_t)d * g.ld_out;
  const double first = g.anchors[2 * d], last = g.anchors[2 * d + 1];
  const double step = g.T > 1 ? (last - first) / (double)(g.T - 1) : 0.0;
  const bool live = c < g.n_chunks;
  double z0[S], z1[S];
  double* st = g.states + ((size_t)d * g.n_chunks + (live ? c : 0)) * (2 * S);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    z0[s] = (kWrite && live) ? st[2 * s] : 0.0;
    z1[s] = (kWrite && live) ? st[2 * s + 1] : 0.0;
  }
  // cooperative tile moves: lane group of 8 handles one chunk's 32 samples (4 per lane)
  const int part = threadIdx.x & 7, rowgrp = threadIdx.x >> 3;  // 32 chunks per pass of the block
  const bool vec_in = (g.ld_in % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.in) & 15u) == 0);
  const bool vec_out = (g.ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.out) & 15u) == 0);
  for (int j = 0; j < kChunk / kSub; ++j) {
    // ---- load stage j of all 256 chunks
    for (int cc = rowgrp; cc < kBlock; cc += kBlock / 8) {
      const long long t = (long long)(c0 + cc) * kChunk + j * kSub + part * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (t + 4 <= g.T && vec_in) {
        const vfloat4 q = *reinterpret_cast<const vfloat4*>(row + t);
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (t + k < g.T) v[k] = row[t + k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) stage[cc * kPitch + part * 4 + k] = v[k];
    }
    __syncthreads();
    // ---- every thread: 32 steps of its own chunk
    const int tb = c * kChunk + j * kSub;
    if (live) {
#pragma unroll 4
      for (int i = 0; i < kSub; ++i) {
        const int t = tb + i;
        if (t >= g.T) break;
        double x = (double)stage[threadIdx.x * kPitch + i];
        if (g.remove_slope) x -= line_at(first, last, step, t, g.T);  // utils/signal/__init__.py:151-152
#pragma unroll
        for (int s = 0; s < S; ++s) {
          // scipy/signal/_sosfilt.pyx: transposed direct form II
          const double y = g.b0[s] * x + z0[s];
          z0[s] = g.b1[s] * x - g.a1[s] * y + z1[s];
          z1[s] = g.b2[s] * x - g.a2[s] * y;
          x = y;
        }
        if (kWrite) stage[threadIdx.x * kPitch + i] = (float)x;
      }
    }
    __syncthreads();
    if (kWrite) {
      // ---- store stage j
      for (int cc = rowgrp; cc < kBlock; cc += kBlock / 8) {
        const long long t = (long long)(c0 + cc) * kChunk + j * kSub + part * 4;
        if (t >= g.T) continue;
        const float* v = stage + cc * kPitch + part * 4;
        if (t + 4 <= g.T && vec_out) {
          *reinterpret_cast<vfloat4*>(orow + t) = vfloat4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (t + k < g.T) orow[t + k] = v[k];
        }
      }
      __syncthreads();
    }
  }
  if (!kWrite && live) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      st[2 * s] = z0[s];
      st[2 * s + 1] = z1[s];
    }
  }
}

// chains the chunks of one detector: states[c] <- initial state of chunk c
template <int S>
__global__ __launch_bounds__(kBlock) void sos_scan_kernel(SosArgs g) {
  const int d = blockIdx.x * kBlock + threadIdx.x;
  if (d >= g.D) return;
  constexpr int N = 2 * S;
  double M[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) M[i][j] = g.M[i * N + j];
  double s[N];
#pragma unroll
  for (int i = 0; i < N; ++i) s[i] = 0.0;
  double* st = g.states + (size_t)d * g.n_chunks * N;
  for (int c = 0; c < g.n_chunks; ++c) {
    double zs[N], nxt[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      zs[i] = st[(size_t)c * N + i];
      st[(size_t)c * N + i] = s[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double acc = zs[i];
#pragma unroll
      for (int j = 0; j < N; ++j) acc = fma(M[i][j], s[j], acc);
      nxt[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = nxt[i];
  }
}

// remove_slope and / or window in place: v = float32(x - line); v = float32(v * w[t])
__global__ __launch_bounds__(kBlock) void detrend_window_kernel(float* __restrict__ data, size_t ld, int D, int T,
                                                              int remove_slope, const double* __restrict__ window,
                                                              const double* __restrict__ anchors) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= T) return;
  const int d0 = blockIdx.y * 16;
  const double wt = window ? window[t] : 1.0;
  for (int d = d0; d < min(d0 + 16, D); ++d) {
    float* row = data + (size_t)d * ld;
    float v = row[t];
    if (remove_slope) {
      const double first = anchors[2 * d], last = anchors[2 * d + 1];
      const double step = T > 1 ? (last - first) / (double)(T - 1) : 0.0;
      v = (float)((double)v - line_at(first, last, step, t, T));
    }
    if (window) v = (float)((double)v * wt);
    row[t] = v;
  }
}

template <int S>
int launch_sos(mrx_ctx* ctx, const SosArgs& g) {
  const dim3 grid(mrx_ceil_div(g.n_chunks, kBlock), g.D);
  hipLaunchKernelGGL((sos_chunk_kernel<S, false>), grid, dim3(kBlock), 0, ctx->stream, g);
  hipLaunchKernelGGL((sos_scan_kernel<S>), dim3(mrx_ceil_div(g.D, kBlock)), dim3(kBlock), 0, ctx->stream, g);
  hipLaunchKernelGGL((sos_chunk_kernel<S, true>), grid, dim3(kBlock), 0, ctx->stream, g);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

}  // namespace

extern "C" {

int mrx_tod_detrend_window(mrx_ctx* ctx, float* d_data, size_t ld, int D, int T, int remove_slope,
                           const double* d_window, double* d_work) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0 || (!remove_slope && !d_window)) return MRX_OK;
  MRX_REQUIRE(ctx, d_data && ld >= (size_t)T, "null pointer or ld smaller than T");
  MRX_REQUIRE(ctx, !remove_slope || d_work, "remove_slope needs 2 * D doubles of scratch");
  if (remove_slope)  // the anchors are themselves rewritten: read them first
    hipLaunchKernelGGL(anchors_kernel, dim3(mrx_ceil_div(D, kBlock)), dim3(kBlock), 0, ctx->stream, d_data, ld, D, T,
                       d_work);
  const dim3 grid(mrx_ceil_div(T, kBlock), mrx_ceil_div(D, 16));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  hipLaunchKernelGGL(detrend_window_kernel, grid, dim3(kBlock), 0, ctx->stream, d_data, ld, D, T, remove_slope,
                     d_window, d_work);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_sosfilt_work_doubles(int D, int T, int n_sections, size_t* doubles) {
  if (D < 0 || T < 0 || n_sections < 1 || n_sections > kMaxSections || !doubles) return MRX_ERR_INVALID;
  *doubles = 2 * (size_t)D + (size_t)D * (size_t)mrx_ceil_div(T > 0 ? T : 1, kChunk) * (size_t)(2 * n_sections) + 16;
  return MRX_OK;
}

int mrx_sosfilt(mrx_ctx* ctx, const double* sos, int n_sections, const double* d_chunk_matrix,
                const float* d_in, size_t ld_in, int D, int T, int remove_slope, float* d_out,
                size_t ld_out, double* d_work) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, sos && d_chunk_matrix && d_in && d_out && d_work, "null pointer");
  MRX_REQUIRE(ctx, n_sections >= 1 && n_sections <= kMaxSections, "1 <= n_sections <= 8");
  MRX_REQUIRE(ctx, ld_in >= (size_t)T && ld_out >= (size_t)T, "leading dimension smaller than T");
  MRX_REQUIRE(ctx, D <= 65535, "D too large for one launch");
  SosArgs g{};
  for (int s = 0; s < n_sections; ++s) {
    const double a0 = sos[6 * s + 3];
    MRX_REQUIRE(ctx, a0 != 0.0, "a0 of a section is zero");
    g.b0[s] = sos[6 * s + 0] / a0;  // scipy normalises by a0 (1 for the filters built here)
    g.b1[s] = sos[6 * s + 1] / a0;
    g.b2[s] = sos[6 * s + 2] / a0;
    g.a1[s] = sos[6 * s + 4] / a0;
    g.a2[s] = sos[6 * s + 5] / a0;
  }
  g.n_sections = n_sections;
  g.in = d_in;
  g.ld_in = ld_in;
  g.out = d_out;
  g.ld_out = ld_out;
  g.D = D;
  g.T = T;
  g.n_chunks = mrx_ceil_div(T, kChunk);
  g.remove_slope = remove_slope;
  g.anchors = d_work;             // [D][2]
  g.states = d_work + 2 * (size_t)D;
  g.M = d_chunk_matrix;
  hipLaunchKernelGGL(anchors_kernel, dim3(mrx_ceil_div(D, kBlock)), dim3(kBlock), 0, ctx->stream, d_in, ld_in, D, T,
                     d_work);
  switch (n_sections) {
    case 1: return launch_sos<1>(ctx, g);
    case 2: return launch_sos<2>(ctx, g);
    case 3: return launch_sos<3>(ctx, g);
    case 4: return launch_sos<4>(ctx, g);
    case 5: return launch_sos<5>(ctx, g);
    case 6: return launch_sos<6>(ctx, g);
    case 7: return launch_sos<7>(ctx, g);
    default: return launch_sos<8>(ctx, g);
  }
}

int mrx_sosfilt_chunk(void) { return kChunk; }

}  // extern "C"
