// Not-a-knot cubic spline along time for every detector, gfx950.
//
//   spline_prepare_kernel : second-derivative solve (float64), time-parallel
//   spline_upsample_kernel: evaluation at the full sample rate -> TOD (float32)
//
// Reference: scipy interp1d(kind="cubic", fill_value="extrapolate") as called
// at sim/atmosphere.py:72-82, i.e. make_interp_spline(k=3) with not-a-knot ends
// on the uniform knots of coordinates.py:292.
//
// Spline form.  With uniform spacing h and m_i = h^2/6 * S''(x_i):
//   interior:    m_{i-1} + 4 m_i + m_{i+1} = y_{i-1} - 2 y_i + y_{i+1} = delta_i
//   not-a-knot:  m_0 - 2 m_1 + m_2 = 0,  m_{n-3} - 2 m_{n-2} + m_{n-1} = 0
// which gives m_1 = delta_1/6, m_{n-2} = delta_{n-2}/6 and a constant (1,4,1)
// tridiagonal system for i = 2..n-3.
//
// Time-parallel solve.  A left-to-right elimination gives m_i + cL_i m_{i+1} =
// dL_i, a right-to-left one m_i + cR_i m_{i-1} = dR_i, and equation i then
// yields m_i from dL_{i-1} and dR_{i+1} alone (a twisted factorisation).  Both
// sweeps are first-order recurrences whose memory of the start decays like
// (2-sqrt 3)^k = 0.268^k, so a thread that owns a chunk of 16 knots starts its
// sweeps 16 knots outside the chunk from a zero state: the error in m, 0.268^16 =
// 7e-10 of a second difference of y, is 100x below the float32 rounding of m
// itself (the form m is stored in) and ~1e-12 of y.  Within 16 knots of an end
// the sweeps start at the true boundary and are exact.  Lanes are consecutive detectors (time-major data),
// so every index and coefficient below is wave-uniform.
#include "mrx_internal.h"
#include "mrx_spline_tile.h"

namespace {

constexpr int kChunk = 16;  // knots owned by one thread
constexpr int kHalo = 16;   // knots of run-in for each sweep

__global__ __launch_bounds__(kBlock) void spline_prepare_kernel(
    const float* __restrict__ y, int D, int n, float2* __restrict__ ym) {
  const int d = blockIdx.x * kBlock + threadIdx.x;
  if (d >= D) return;
  const int a = blockIdx.y * kChunk;       // first knot of this chunk
  const int b = min(a + kChunk, n);        // one past the last
  const int lo = 2, hi = n - 3;            // interior unknowns (empty if n < 5)

  // ---- chunks far from both ends (all but a few): fixed trip counts -----------
  // Every knot the two sweeps touch is loaded up front (static indices, so the
  // window lives in registers and all loads are in flight together); the pivots
  // have converged to alpha and no boundary term applies.
  if (b == a + kChunk && a - 1 - kHalo - lo >= 16 && hi - (b + kHalo) >= 16) {
    constexpr int kWin = kChunk + 2 * kHalo + 4;  // knots a-2-kHalo .. b+kHalo+1
    const float* base = y + (size_t)(a - 2 - kHalo) * D + d;
    float w[kWin];
#pragma unroll
    for (int k = 0; k < kWin; ++k) w[k] = base[(size_t)k * D];
    // delta of knot (a - 2 - kHalo + k), 1 <= k <= kWin - 2
    auto delta = [&](int k) -> double {
      return ((double)w[k - 1] - 2.0 * (double)w[k]) + (double)w[k + 1];
    };
    double dLm[kChunk];
    double dl = 0.0;
#pragma unroll
    for (int k = 1; k <= kHalo; ++k) dl = (delta(k) - dl) * kAlpha;  // a-1-kHalo .. a-2
#pragma unroll
    for (int k = 0; k < kChunk; ++k) {  // a-1 .. b-2
      dl = (delta(kHalo + 1 + k) - dl) * kAlpha;
      dLm[k] = dl;
    }
    double dr = 0.0;
#pragma unroll
    for (int k = kWin - 2; k >= kChunk + kHalo + 2; --k)  // b+kHalo .. b
      dr = (delta(k) - dr) * kAlpha;
    constexpr double kInvDen = 1.0 / (4.0 - 2.0 * kAlpha);
    float2* dst = ym + (size_t)a * D + d;
#pragma unroll
    for (int k = kChunk - 1; k >= 0; --k) {  // knot a + k
      const double r = delta(kHalo + 2 + k);
      const double m = (r - dLm[k] - dr) * kInvDen;
      dr = (r - dr) * kAlpha;
      dst[(size_t)k * D] = make_float2(w[kHalo + 2 + k], (float)m);
    }
    return;
  }

  auto Y = [&](int i) -> double { return (double)y[(size_t)i * D + d]; };

  const double m1 = (Y(0) - 2.0 * Y(1) + Y(2)) * (1.0 / 6.0);
  const double mn2 = (Y(n - 3) - 2.0 * Y(n - 2) + Y(n - 1)) * (1.0 / 6.0);
  auto rhs = [&](int i, double ym_, double yc_, double yp_) -> double {
    double r = (ym_ - 2.0 * yc_) + yp_;
    if (i == lo) r -= m1;
    if (i == hi) r -= mn2;
    return r;
  };

  // ---- left sweep: dLm[k] = dL_{a-1+k} -----------------------------------
  double dLm[kChunk];
  {
    const int target = a - 1 - kHalo;
    const int s0 = (target - lo < 16) ? lo : target;
    double dl = 0.0;
    int i = s0;
    double ym_ = 0.0, yc_ = 0.0;
    if (s0 <= hi) {
      ym_ = Y(s0 - 1);
      yc_ = Y(s0);
    }
    for (; i < a - 1 && i <= hi; ++i) {
      const double yp_ = Y(i + 1);
      dl = (rhs(i, ym_, yc_, yp_) - dl) * qf(i - lo);
      ym_ = yc_;
      yc_ = yp_;
    }
#pragma unroll
    for (int k = 0; k < kChunk; ++k) {
      const int ii = a - 1 + k;
      if (ii >= s0 && ii <= hi) {
        const double yp_ = Y(ii + 1);
        dl = (rhs(ii, ym_, yc_, yp_) - dl) * qf(ii - lo);
        ym_ = yc_;
        yc_ = yp_;
      }
      dLm[k] = dl;
    }
  }

  // ---- right sweep down to the chunk: dr = dR_b ----------------------------
  double dr = 0.0;
  {
    const int target = b + kHalo;
    const int e0 = (hi - target < 16) ? hi : target;
    if (e0 >= b && e0 >= lo) {
      double yc_ = Y(e0), yp_ = Y(e0 + 1);
      for (int i = e0; i >= b; --i) {
        const double ym_ = Y(i - 1);
        dr = (rhs(i, ym_, yc_, yp_) - dr) * qf(hi - i);
        yp_ = yc_;
        yc_ = ym_;
      }
    }
  }

  // ---- combine, descending through the chunk -------------------------------
  const int top = b - 1;
  double yc_ = Y(top);
  double yp_ = (top + 1 < n) ? Y(top + 1) : 0.0;
#pragma unroll
  for (int k = kChunk - 1; k >= 0; --k) {
    const int i = a + k;
    if (i >= n) continue;
    const double ym_ = (i >= 1) ? Y(i - 1) : 0.0;
    if (i >= 1 && i <= n - 2) {
      const double delta = (ym_ - 2.0 * yc_) + yp_;
      double m;
      if (i == 1 || i == n - 2) {
        m = delta * (1.0 / 6.0);
      } else {
        double r = delta;
        if (i == lo) r -= m1;
        if (i == hi) r -= mn2;
        double num = r, den = 4.0;
        if (i > lo) {
          num -= dLm[k];
          den -= qf(i - 1 - lo);
        }
        if (i < hi) {
          num -= dr;
          den -= qf(hi - i - 1);
        }
        m = num / den;
        dr = (r - dr) * qf(hi - i);  // dR_i, for knot i-1
      }
      ym[(size_t)i * D + d] = make_float2((float)yc_, (float)m);
      if (i == 2)  // m_0 = 2 m_1 - m_2
        ym[d] = make_float2((float)Y(0), (float)(2.0 * m1 - m));
      if (i == n - 3)  // m_{n-1} = 2 m_{n-2} - m_{n-3}
        ym[(size_t)(n - 1) * D + d] =
            make_float2((float)Y(n - 1), (float)(2.0 * mn2 - m));
    }
    yp_ = yc_;
    yc_ = ym_;
  }
}

template <bool kHasScale, int kMaxKnots>
__global__ __launch_bounds__(kBlock) void spline_upsample_kernel(
    const float2* __restrict__ ym, int D, int n, double ta0, double inv_dta,
    const double* __restrict__ t, int T, const float* __restrict__ scale,
    const int32_t* __restrict__ rows, float* __restrict__ out, size_t ld,
    int vec_ok, int groups) {
  constexpr int kPitch = kMaxKnots + 1;
  // destination row of detector d (wave-uniform): the caller may keep its
  // detectors in a locality order and still get the TOD in its own row order
  __shared__ float2 tile[kTileDet * kPitch];
  // the group's destination rows are staged with the knots: a scalar load of rows[d] inside the
  // row loop would stall every iteration on its latency
  __shared__ int row_lds[kTileDet];
  auto row_of = [&](int dl, int d) -> size_t { return rows ? (size_t)row_lds[dl] : (size_t)d; };
  const int s_tile = blockIdx.x * kTileSamples;
  const int sb = s_tile + threadIdx.x * kSamplesPerThread;

  // per-sample interval and weights: computed once, reused for `groups` tiles of
  // 16 detector rows each (the float64 prologue is amortised over 16*groups rows)
  SampleWeights w;
  sample_weights(t, sb, T, n, ta0, inv_dta, w);

  // knot range of the tile (wave-uniform; t ascending)
  const int s_last = min(s_tile + kTileSamples, T) - 1;
  const int jmin = interval_of((t[s_tile] - ta0) * inv_dta, n);
  const int jmax = interval_of((t[s_last] - ta0) * inv_dta, n) + 1;
  const int K = jmax - jmin + 1;
  const bool full = (sb + kSamplesPerThread <= T) && vec_ok;
  int r[kSamplesPerThread];
#pragma unroll
  for (int q = 0; q < kSamplesPerThread; ++q)
    r[q] = min(max(w.j[q] - jmin, 0), max(K - 2, 0));  // in range even if t is unsorted

  for (int g = 0; g < groups; ++g) {
    const int d0 = (blockIdx.y * groups + g) * kTileDet;
    if (d0 >= D) break;
    const int nd = min(kTileDet, D - d0);
    if (K <= kMaxKnots) {
      if (g > 0) __syncthreads();  // the previous group is done with the image
      if (rows && (int)threadIdx.x < nd) row_lds[threadIdx.x] = rows[d0 + threadIdx.x];
      {  // 16 lanes cover the 16 detector rows of one knot: 128 contiguous bytes
        const int dl = threadIdx.x & (kTileDet - 1);
        const int d = d0 + dl;
        for (int rr = threadIdx.x / kTileDet; rr < K; rr += kBlock / kTileDet) {
          float2 v = make_float2(0.f, 0.f);
          if (d < D) v = ym[(size_t)(jmin + rr) * D + d];
          tile[dl * kPitch + rr] = v;
        }
      }
      __syncthreads();
      if (full) {
#pragma unroll 4
        for (int dl = 0; dl < nd; ++dl) {
          const float2* row = tile + dl * kPitch;
          float o[kSamplesPerThread];
#pragma unroll
          for (int q = 0; q < kSamplesPerThread; ++q)
            o[q] = spline_eval(w, q, row[r[q]], row[r[q] + 1]);
          if (kHasScale) {
            const float gsc = scale[d0 + dl];
#pragma unroll
            for (int q = 0; q < kSamplesPerThread; ++q) o[q] *= gsc;
          }
          const vfloat4 v = {o[0], o[1], o[2], o[3]};
#ifdef MRX_PLAIN_STORE
          *reinterpret_cast<vfloat4*>(out + row_of(dl, d0 + dl) * ld + sb) = v;
#else
          __builtin_nontemporal_store(
              v, reinterpret_cast<vfloat4*>(out + row_of(dl, d0 + dl) * ld + sb));
#endif
        }
      } else {
        for (int dl = 0; dl < nd; ++dl) {
          const float2* row = tile + dl * kPitch;
          const float gsc = kHasScale ? scale[d0 + dl] : 1.0f;
          float* dst = out + row_of(dl, d0 + dl) * ld + sb;
#pragma unroll
          for (int q = 0; q < kSamplesPerThread; ++q)
            if (sb + q < T) dst[q] = gsc * spline_eval(w, q, row[r[q]], row[r[q] + 1]);
        }
      }
    } else {
      // low upsampling ratio: knots straight from global memory
      for (int dl = 0; dl < nd; ++dl) {
        const int d = d0 + dl;
        const float gsc = kHasScale ? scale[d] : 1.0f;
        float* dst = out + (rows ? (size_t)rows[d] : (size_t)d) * ld + sb;
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q)
          if (sb + q < T)
            dst[q] = gsc * spline_eval(w, q, ym[(size_t)w.j[q] * D + d],
                                       ym[(size_t)(w.j[q] + 1) * D + d]);
      }
    }
  }
}

// A RESIDENT grid takes tiles from a queue (one agent-scope atomic per tile; tiles numbered row group by row group,
// time tile fastest -- the order a two-dimensional grid is dispatched in): against one workgroup per tile, 1.59 vs
// 1.69 ms for 10 000 x 240 000 alone (0.76 of 8 TB/s): no dispatch per tile, no tail of half-empty CUs.  The queue
// and not a fixed stride, because only some of the grid may be on the chip at first.  A launch that SHARES the chip
// with another kernel's launches to come (the two-stream pipeline: the next block's sampler must find room while
// this writer runs) takes one workgroup per tile instead, MRX_OPT_WRITER_PER_TILE: a resident grid never makes room
// (pipelined atlast_10k 2.61 against 2.3 ms).
template <bool kHasScale, int kMaxKnots, int kG, bool kQueue>
// 5 waves per SIMD = up to 96 registers (the kernel takes 94, nothing spilled).  Beside the resident sampler
// (3 workgroups per CU x 64 registers) 3 writer waves per SIMD still fit (192 + 288 <= 512); with 2 the writer
// loses a fifth of its rate.  The first fused version was capped at 72 registers (the round-2
// sampler took 3 x 96) and spilled 10 values: 44 bytes of scratch per lane = 11 KB per workgroup against the
// 128 KB it writes -- the 8 % of extra WRITE_SIZE in profiles/r03_traffic.json, and 3 % of the step.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MRX_WRITER_WAVES))) void spline_upsample_fused_kernel(
    const float* __restrict__ y, int D, int n, double ta0, double inv_dta,
    const double* __restrict__ t, int T, const float* __restrict__ scale,
    const int32_t* __restrict__ rows, float* __restrict__ out, size_t ld,
    int vec_ok, int batches, int nsx, int total, int* ctl) {
  // dynamic LDS (FusedLds<...>::kBytes): with a static size the compiler derives the occupancy
  // from it and ignores the register bound above
  extern __shared__ __align__(16) unsigned char fused_lds[];
  if (!kQueue) {  // one workgroup per tile, a two-dimensional grid (time tiles x row groups): MRX_OPT_WRITER_PER_TILE
    fused_writer_tile<kHasScale, kMaxKnots, kG>(y, D, D, n, ta0, inv_dta, t, T, scale, rows, out, ld, vec_ok, batches,
                                                 (int)blockIdx.x, (int)blockIdx.y, fused_lds);
    return;
  }
  __shared__ int s_next;
  for (;;) {
    if (threadIdx.x == 0) s_next = __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();  // (also: the previous tile's readers of the LDS images are done)
    const int tile = s_next;
    if (tile >= total) break;
    const int by = tile / nsx;
    fused_writer_tile<kHasScale, kMaxKnots, kG>(y, D, D, n, ta0, inv_dta, t, T, scale, rows, out, ld, vec_ok, batches,
                                                 tile - by * nsx, by, fused_lds);
  }
  synth_leave(ctl, 0);
}

}  // namespace

extern "C" {

int mrx_spline_prepare(mrx_ctx* ctx, const float* d_y, int D, int Ta,
                       float* d_ym) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0, "negative D");
  if (D == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_y && d_ym, "null pointer");
  if (Ta < 4)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED,
                    "cubic interpolation needs at least 4 coarse samples "
                    "(got %d), as scipy interp1d(kind='cubic') does", Ta);
  dim3 grid(mrx_ceil_div(D, kBlock), mrx_ceil_div(Ta, kChunk));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "Ta too large for one launch");
  hipLaunchKernelGGL(spline_prepare_kernel, grid, dim3(kBlock), 0, ctx->stream,
                     d_y, D, Ta, reinterpret_cast<float2*>(d_ym));
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_spline_upsample(mrx_ctx* ctx, const float* d_ym, int D, int Ta,
                        double ta0, double dta, const double* d_t, int T,
                        const float* d_scale, const int32_t* d_rows,
                        float* d_out, size_t ld_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_ym && d_t && d_out, "null pointer");
  MRX_REQUIRE(ctx, dta > 0.0, "coarse step must be positive");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  if (Ta < 4)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED,
                    "cubic interpolation needs at least 4 coarse samples");
  // detector tiles per workgroup: the per-sample prologue is shared by all of them
  int groups = ctx->options[MRX_OPT_UPSAMPLE_GROUPS];
  if (groups <= 0) groups = 2;  // measured best of 1, 2, 4, 8, 16 on atlast_10k
  while (groups > 1 && (long long)mrx_ceil_div(T, kTileSamples) *
                               mrx_ceil_div(D, kTileDet * groups) < 4LL * 256 * 4)
    groups /= 2;  // keep the chip full on small problems
  dim3 grid(mrx_ceil_div(T, kTileSamples), mrx_ceil_div(D, kTileDet * groups));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  const int vec_ok =
      (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0);
  const float2* ym = reinterpret_cast<const float2*>(d_ym);
  // LDS image size from the expected knots per 1024-sample tile (the sample
  // times live on the device; a tile that needs more knots than the image
  // holds falls back to global loads inside the kernel)
  const double knots_per_tile = (double)kTileSamples * (double)Ta / (double)T;
  // the 64-knot image (8 workgroups per CU) measured 8 % slower than the 256-knot
  // one (4 per CU) on the streaming write, so it is kept for reference only
  const bool small = false && knots_per_tile + 4.0 <= 64.0;
#define MRX_LAUNCH_UP(S, K)                                                   \
  hipLaunchKernelGGL((spline_upsample_kernel<S, K>), grid, dim3(kBlock), 0,   \
                     ctx->stream, ym, D, Ta, ta0, 1.0 / dta, d_t, T, d_scale, \
                     d_rows, d_out, ld_out, vec_ok, groups)
  if (d_scale) {
    if (small) MRX_LAUNCH_UP(true, 64); else MRX_LAUNCH_UP(true, 256);
  } else {
    if (small) MRX_LAUNCH_UP(false, 64); else MRX_LAUNCH_UP(false, 256);
  }
#undef MRX_LAUNCH_UP
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

}  // extern "C"

// The control block of the next launch that takes tiles from a queue: a ring of kSynthCtlSlots blocks per context,
// zeroed once (every launch leaves its block zero: synth_leave), so that launches of one context that overlap on
// different streams do not share a queue.
int mrx_synth_ctl(mrx_ctx* ctx, int** out) {
  if (!ctx->d_synth_ctl) {
    MRX_HIP(ctx, hipMalloc(&ctx->d_synth_ctl, sizeof(int) * kSynthCtlInts * mrx_ctx::kSynthCtlSlots));
    MRX_HIP(ctx, hipMemsetAsync(ctx->d_synth_ctl, 0, sizeof(int) * kSynthCtlInts * mrx_ctx::kSynthCtlSlots, ctx->stream));
    MRX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (once per context: later launches may come on other streams)
  }
  *out = ctx->d_synth_ctl + (size_t)kSynthCtlInts * ctx->synth_ctl_next;
  ctx->synth_ctl_next = (ctx->synth_ctl_next + 1) % mrx_ctx::kSynthCtlSlots;
  return MRX_OK;
}

extern "C" {

int mrx_spline_upsample_fused(mrx_ctx* ctx, const float* d_y, int D, int Ta,
                              double ta0, double dta, const double* d_t, int T,
                              const float* d_scale, const int32_t* d_rows,
                              float* d_out, size_t ld_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_y && d_t && d_out, "null pointer");
  MRX_REQUIRE(ctx, dta > 0.0, "coarse step must be positive");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  if (Ta < 4)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED,
                    "cubic interpolation needs at least 4 coarse samples "
                    "(got %d), as scipy interp1d(kind='cubic') does", Ta);
  const int vec_ok =
      (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0);
  // the 64-knot image solves two groups of 16 rows at once (upsampling ratios >= ~18: a
  // tile of 1024 samples then spans <= 58 knots + the widening at the ends); the 256-knot
  // image one group (ratios down to ~4.2); below that the kernel's per-sample path runs
  const double knots_per_tile = (double)kTileSamples * (double)Ta / (double)T;
  const bool small = knots_per_tile + 6.0 <= (double)kSmallKnots;
  const int rows_per_batch = small ? 2 * kTileDet : kTileDet;
  int batches = ctx->options[MRX_OPT_UPSAMPLE_GROUPS];
  // rows per workgroup and occupancy do not matter (round 3, 10 000 x 240 000 alone: 1.77-1.79 ms = 5.35-5.41 TB/s
  // for 1, 2 or 4 batches at 5, 4 or 3 workgroups per CU; 5.5 at 2 per CU): the kernel sits on the chip's
  // store ceiling (5.5-5.8 TB/s, scripts/exp_upsample.hip)
  // (alone 1, 2 and 4 batches run alike; beside the sampler one is best: pipelined TOD synthesis 2.10 / 2.14 / 2.19 ms,
  // scripts/exp_writer_batches.py)
  if (batches <= 0) batches = 1;
  while (batches > 1 && (long long)mrx_ceil_div(T, kTileSamples) *
                                mrx_ceil_div(D, rows_per_batch * batches) < 4LL * 256 * 4)
    batches /= 2;  // keep the chip full on small problems
  const long long nsx = mrx_ceil_div(T, kTileSamples);
  const long long n_tiles = nsx * mrx_ceil_div(D, rows_per_batch * batches);
  MRX_REQUIRE(ctx, n_tiles <= 0x7fffffffLL - 65536, "too many tiles for one launch");
  const size_t lds = small ? FusedLds<kSmallKnots, 2>::kBytes : FusedLds<256, 1>::kBytes;
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  const long long per_cu = std::max<long long>(1, std::min<long long>(MRX_WRITER_WAVES, (long long)(ctx->lds_per_cu > 0 ? ctx->lds_per_cu : 160 * 1024) / (long long)(lds + 64)));
  const bool per_tile = ctx->options[MRX_OPT_WRITER_PER_TILE] != 0;
  const dim3 grid_q((unsigned)std::min(n_tiles, per_cu * n_cu));
  const dim3 grid_t((unsigned)nsx, (unsigned)mrx_ceil_div(D, rows_per_batch * batches));
  MRX_REQUIRE(ctx, !per_tile || grid_t.y <= 65535u, "D too large for one launch");
  int* ctl = nullptr;
  if (!per_tile) {
    const int rc = mrx_synth_ctl(ctx, &ctl);
    if (rc != MRX_OK) return rc;
  }
#define MRX_LAUNCH_UPF_Q(S, K, G, Q)                                                 \
  hipLaunchKernelGGL((spline_upsample_fused_kernel<S, K, G, Q>), Q ? grid_q : grid_t, dim3(kBlock), (FusedLds<K, G>::kBytes), \
                     ctx->stream, d_y, D, Ta, ta0, 1.0 / dta, d_t, T, d_scale,       \
                     d_rows, d_out, ld_out, vec_ok, batches, (int)nsx, (int)n_tiles, ctl)
#define MRX_LAUNCH_UPF(S, K, G) do { if (per_tile) MRX_LAUNCH_UPF_Q(S, K, G, false); else MRX_LAUNCH_UPF_Q(S, K, G, true); } while (0)
  if (d_scale) {
    if (small) MRX_LAUNCH_UPF(true, kSmallKnots, 2); else MRX_LAUNCH_UPF(true, 256, 1);
  } else {
    if (small) MRX_LAUNCH_UPF(false, kSmallKnots, 2); else MRX_LAUNCH_UPF(false, 256, 1);
  }
#undef MRX_LAUNCH_UPF_Q
#undef MRX_LAUNCH_UPF
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

}  // extern "C"
