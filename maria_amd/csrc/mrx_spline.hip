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
#include <type_traits>

#include "mrx_internal.h"
#include "mrx_spectral.h"  // Philox, Box-Muller (the two-rate noise writer)
#define MRX_PX_CONTRACT_FAST_AFTER  // (the sampler's code keeps its two roundings; this TU contracts as before)
#include "mrx_sample_px.h"  // the sampler role of atm_tod_kernel

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 16;  // knots owned by one thread
constexpr int kHalo = 16;   // knots of run-in for each sweep
constexpr int kQTab = 32;   // pivots tabulated before they equal alpha
constexpr double kAlpha = 0.26794919243112270647;  // 2 - sqrt(3)

// q_0 = 1/4, q_{k+1} = 1/(4 - q_k): the elimination multipliers; q_k -> alpha
// with error ~0.072^k, i.e. equal to alpha in float64 from k = 15 on.
struct QTable {
  double q[kQTab];
  constexpr QTable() : q{} {
    double c = 0.25;
    for (int k = 0; k < kQTab; ++k) {
      q[k] = c;
      c = 1.0 / (4.0 - c);
    }
  }
};
__constant__ QTable kQ = QTable();

__device__ __forceinline__ double qf(int k) {
  return k < kQTab ? kQ.q[k] : kAlpha;
}

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

// ---------------------------------------------------------------------------
// Evaluation.  A workgroup writes a tile of kTileDet detector rows x 1024
// consecutive samples.  Each thread owns 4 consecutive samples (one 16-byte
// store per detector row, 1 KiB contiguous per wave), computes their interval
// index and the four basis weights once in float64 and reuses them for all
// rows of the tile.  The (y, m) knots the tile needs are staged through LDS,
// detector-major so that a wave's reads are consecutive 8-byte words
// (conflict-free ds_read_b64).  kMaxKnots is the LDS image's capacity in knots:
// 64 covers upsampling ratios >= 17 with 8 KiB of LDS (8 workgroups per CU),
// 256 covers ratios down to ~4; below that the tile reads its knots from
// global memory (correct, slower: such ratios do not occur in maria, whose
// coarse step is >= 0.1 s).
#ifndef MRX_TILE_DET
#define MRX_TILE_DET 16
#endif
constexpr int kTileDet = MRX_TILE_DET;
constexpr int kSamplesPerThread = 4;
constexpr int kTileSamples = kBlock * kSamplesPerThread;  // 1024

typedef float vfloat4 __attribute__((ext_vector_type(4)));

struct SampleWeights {
  int j[kSamplesPerThread];
  float wb[kSamplesPerThread], wc[kSamplesPerThread], wd[kSamplesPerThread];
};

__device__ __forceinline__ int interval_of(double x, int n) {
  const int jj = (int)floor(fmin(fmax(x, -1.0), 2.0e9));
  return min(max(jj, 0), n - 2);
}

// interval and basis weights of the samples at times tq[] (float64 once per sample, reused by every row)
__device__ __forceinline__ void sample_weights_at(const double (&tq)[kSamplesPerThread], int n, double ta0,
                                                  double inv_dta, SampleWeights& w) {
#pragma unroll
  for (int q = 0; q < kSamplesPerThread; ++q) {
    const double x = (tq[q] - ta0) * inv_dta;
    const int jj = interval_of(x, n);
    const double u = x - (double)jj;  // may be < 0 or > 1: extrapolation
    const double v = 1.0 - u;
    w.j[q] = jj;
    w.wb[q] = (float)u;
    w.wc[q] = (float)(v * (v * v - 1.0));
    w.wd[q] = (float)(u * (u * u - 1.0));
  }
}

__device__ __forceinline__ void sample_weights(const double* __restrict__ t,
                                               int sb, int T, int n, double ta0,
                                               double inv_dta,
                                               SampleWeights& w) {
#pragma unroll
  for (int q = 0; q < kSamplesPerThread; ++q) {
    const int s = min(sb + q, T - 1);
    const double x = (t[s] - ta0) * inv_dta;
    const int jj = interval_of(x, n);
    const double u = x - (double)jj;  // may be < 0 or > 1: extrapolation
    const double v = 1.0 - u;
    w.j[q] = jj;
    w.wb[q] = (float)u;
    w.wc[q] = (float)(v * (v * v - 1.0));
    w.wd[q] = (float)(u * (u * u - 1.0));
  }
}

// y0 + [wb (y1 - y0) + wc m0 + wd m1]: the difference of neighbouring knots is exact in float32 and the
// bracket is small against y0 (the loading's fluctuation is ~1 % of its mean), so the value carries ONE
// rounding at the size of y -- the output's own -- where wa y0 + wb y1 + ... carried three or four
// (measured on the fluctuation at full size: 1e-4 -> see DESIGN 4)
__device__ __forceinline__ float spline_eval(const SampleWeights& w, int q,
                                             float2 k0, float2 k1) {
  float acc = w.wd[q] * k1.y;
  acc = fmaf(w.wc[q], k0.y, acc);
  acc = fmaf(w.wb[q], k1.x - k0.x, acc);
  return k0.x + acc;
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

// ---------------------------------------------------------------------------
// Solve + evaluation in ONE kernel: the writer reads the raw coarse samples y
// (time-major, what the sampler wrote) and computes the second derivatives of its
// own tile in the prologue, so no (y, m) buffer and no separate solve launch exist.
//
// A tile of 1024 samples needs the knots jmin..jmax (~28 at 400 Hz over 0.1 s
// knots).  The tile stages y of those knots plus kFHalo + 3 either side, then
//   1. sweeps: one thread per (row, direction) runs the first-order recurrences
//      of the twisted factorisation over the window -- from a zero state kFHalo
//      knots outside (0.268^16 = 7e-10 of a second difference, see the header of
//      this file) or from the true boundary when the window reaches it, with the
//      true pivots q(k) either way -- leaving dL_{i-1} and dR_{i+1} in knot i's
//      LDS slot;
//   2. combine: one thread per (row, knot) forms m_i and overwrites the slot with
//      (y_i, m_i); the not-a-knot ends m_0, m_{n-1} follow from their neighbours;
//   3. the evaluation loop of spline_upsample_kernel, unchanged.
// kG groups of 16 rows are solved together (32 rows x 2 directions = one wave).
// The sweeps cost latency in one wave (~50 dependent float64 steps), not
// throughput: the other workgroups of the CU keep the store pipeline busy.
constexpr int kFHalo = 16;
#ifndef MRX_WRITER_WAVES
#define MRX_WRITER_WAVES 5  // occupancy target of the fused writer (see the kernel): 6 -> 80 registers, 2 spilled
#endif

template <int kMaxKnots, int kG>
struct FusedLds {
  static constexpr int kRows = kTileDet * kG;
  static constexpr size_t kBytes = sizeof(float2) * kRows * (kMaxKnots + 1) +
                                   sizeof(float) * kRows * ((kMaxKnots + 2 * kFHalo + 6) | 1) + sizeof(int) * kRows;
};

// One tile of the fused writer: time tile `sx` (kTileSamples samples), row group `by` (batches x kRows rows of the
// D rows `y` holds).  The body of spline_upsample_fused_kernel (one tile per workgroup) and of the writer role of
// atm_tod_kernel (a workgroup takes tiles from a queue); `fused_lds` is the dynamic LDS, FusedLds<...>::kBytes;
// `ldy` the pitch of y's rows in floats (D in the stand-alone kernel).
template <bool kHasScale, int kMaxKnots, int kG, bool kHandedOver = false>
__device__ __forceinline__ void fused_writer_tile(
    const float* __restrict__ y, int ldy, int D, int n, double ta0, double inv_dta,
    const double* __restrict__ t, int T, const float* __restrict__ scale,
    const int32_t* __restrict__ rows, float* __restrict__ out, size_t ld,
    int vec_ok, int batches, int sx, int by, unsigned char* fused_lds) {
  constexpr int kRows = kTileDet * kG;
  constexpr int kPitch = kMaxKnots + 1;
  constexpr int kWin = kMaxKnots + 2 * kFHalo + 6;  // raw knots: the image's + (halo + 3) either side
  constexpr int kWPitch = kWin | 1;                 // odd: rows fall on distinct banks
  constexpr int kSegKnots = kMaxKnots - 2;          // knots of one segment before the widening at the ends
  float2* tile = reinterpret_cast<float2*>(fused_lds);                 // [kRows][kPitch]
  float* yraw = reinterpret_cast<float*>(tile + kRows * kPitch);       // [kRows][kWPitch]
  int* row_lds = reinterpret_cast<int*>(yraw + kRows * kWPitch);       // [kRows]
  auto row_of = [&](int dl, int d) -> size_t { return rows ? (size_t)row_lds[dl] : (size_t)d; };
  const int s_tile = sx * kTileSamples;
  const int sb = s_tile + threadIdx.x * kSamplesPerThread;

  // the thread's sample times: loaded now (in flight during the staging below), turned into interval
  // and weights after the solve -- 8 registers live through the prologue instead of 24
  double tq[kSamplesPerThread];
#pragma unroll
  for (int q = 0; q < kSamplesPerThread; ++q) tq[q] = t[min(sb + q, T - 1)];

  // knot range of the tile (workgroup-uniform; t ascending).  It normally fits the image: one
  // segment.  A tile that spans more knots (upsampling ratio below ~1024 / kMaxKnots) is
  // walked in segments of kSegKnots knots, each solved and evaluated on its own.
  const int s_last = min(s_tile + kTileSamples, T) - 1;
  const int j_first = interval_of((t[s_tile] - ta0) * inv_dta, n);
  const int j_end = interval_of((t[s_last] - ta0) * inv_dta, n) + 1;  // last knot needed
  const bool single = j_end - j_first + 1 <= kSegKnots;
  const int lo = 2, hi = n - 3;  // interior unknowns (empty if n < 5)
  const bool full = (sb + kSamplesPerThread <= T) && vec_ok && single;

  for (int g = 0; g < batches; ++g) {
    const int d0 = (by * batches + g) * kRows;
    if (d0 >= D) break;
    const int nd = min(kRows, D - d0);
    for (int ja = j_first; ja < j_end; ja += kSegKnots - 1) {
      // intervals ja .. jb - 1 are evaluated from knots ja .. jb; widened so that the knots the
      // not-a-knot ends are derived from (1, 2 and n-2, n-3) are solved in the same image
      const int jb = min(ja + kSegKnots - 1, j_end);
      int jmin = ja, jmax = jb;
      if (jmin == 0) jmax = max(jmax, 2);
      if (jmax == n - 1) jmin = min(jmin, n - 3);
      const int K = jmax - jmin + 1;  // <= kMaxKnots
      const int w0 = max(jmin - 3 - kFHalo, 0), w1 = min(jmax + 3 + kFHalo, n - 1);
      const int Wn = w1 - w0 + 1;
      if (g > 0 || ja > j_first) __syncthreads();  // the previous pass is done with the images
      if (rows && (int)threadIdx.x < nd) row_lds[threadIdx.x] = rows[d0 + threadIdx.x];
      {  // kRows lanes cover the rows of one knot: 64 or 128 contiguous bytes; 8 loads in flight per thread
        const int dl = threadIdx.x & (kRows - 1);
        const int d = d0 + dl;
        constexpr int kStep = kBlock / kRows, kFly = 8;
        const float* src = y + (size_t)w0 * ldy + min(d, D - 1);
        for (int kk0 = threadIdx.x / kRows; kk0 < Wn; kk0 += kStep * kFly) {
          float v[kFly];
#pragma unroll
          for (int u = 0; u < kFly; ++u) {
            const float* q = src + (size_t)min(kk0 + u * kStep, Wn - 1) * ldy;
            // kHandedOver: y was written in THIS launch by other CUs (write-through stores): global_load_dword sc1,
            // past this CU's L1, which no other CU's store refreshes (atm_tod_kernel)
            v[u] = kHandedOver ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
          }
#pragma unroll
          for (int u = 0; u < kFly; ++u)
            if (kk0 + u * kStep < Wn) yraw[dl * kWPitch + kk0 + u * kStep] = d < D ? v[u] : 0.0f;
        }
      }
      __syncthreads();
      // Far from both ends of the knot axis (all but the first and last tile or two) every pivot the
      // sweeps and the combine touch equals alpha to float32 rounding (q_k - alpha ~ 0.02 x 0.072^k)
      // and no not-a-knot term applies: float32 recurrences with a constant multiplier, every thread
      // busy.  Second differences are formed as (y_{i-1} - y_i) + (y_{i+1} - y_i): both differences
      // are exact in float32 for a smooth series, so nothing is lost against the float64 form.
      const bool interior = jmin - 1 - kFHalo - lo >= 8 && hi - (jmax + 1 + kFHalo) >= 8;  // workgroup-uniform
      if (interior) {
        // ---- 1. sweeps: (row, direction, segment of the knot range), run-in of kFHalo knots each ----
        constexpr int kSeg = kBlock / (2 * kRows);
        constexpr float kAlphaF = (float)kAlpha;
        {
          const int dl = threadIdx.x & (kRows - 1);
          const int rest = threadIdx.x / kRows;
          const int seg = rest % kSeg, dir = rest / kSeg;  // dir is wave-uniform
          const int sgn = dir ? -1 : 1;
          const int Ls = (K + kSeg - 1) / kSeg;
          if (seg * Ls < K) {
            // targets in the order of travel: dL_i for i = jmin-1 .. jmax-1 (stored with knot i+1),
            // dR_i for i = jmax+1 .. jmin+1 (stored with knot i-1)
            const int f = dir ? jmax + 1 - seg * Ls : jmin - 1 + seg * Ls;
            const int cnt = min(Ls, K - seg * Ls);
            const float* yr = yraw + dl * kWPitch - w0;
            float* dst = reinterpret_cast<float*>(tile + dl * kPitch - jmin) + dir;  // dst[2 * knot]
            // the recurrence d <- alpha (delta_i - d) as one fused multiply-add per step on the critical
            // path; knots are fetched eight at a time (reads past the segment's end stay inside the LDS
            // allocation and feed steps whose result is not stored)
            constexpr int kFly = 8;
            static_assert(kFHalo % kFly == 0, "the run-in is a whole number of batches");
            int i = f - sgn * kFHalo;
            float prev = yr[i - sgn], cur = yr[i], d = 0.0f;
            for (int k0 = 0; k0 < kFHalo + cnt; k0 += kFly) {
              float nx[kFly];
#pragma unroll
              for (int u = 0; u < kFly; ++u) nx[u] = yr[i + sgn * (u + 1)];
#pragma unroll
              for (int u = 0; u < kFly; ++u) {
                const float rr = ((prev - cur) + (nx[u] - cur)) * kAlphaF;
                d = fmaf(-kAlphaF, d, rr);
                if (k0 + u >= kFHalo && k0 + u < kFHalo + cnt) dst[2 * (i + sgn * (u + 1))] = d;
                prev = cur;
                cur = nx[u];
              }
              i += sgn * kFly;
            }
          }
        }
        __syncthreads();
        // ---- 2. combine -------------------------------------------------------------------
        {
          const int dl = threadIdx.x & (kRows - 1);
          const float* yr = yraw + dl * kWPitch - w0 + jmin;
          constexpr float kInvDenF = (float)(1.0 / (4.0 - 2.0 * kAlpha));
          for (int kk = threadIdx.x / kRows; kk < K; kk += kBlock / kRows) {
            float2* slot = tile + dl * kPitch + kk;
            const float2 sv = *slot;
            const float yc_ = yr[kk];
            const float delta = (yr[kk - 1] - yc_) + (yr[kk + 1] - yc_);
            *slot = make_float2(yc_, ((delta - sv.x) - sv.y) * kInvDenF);
          }
        }
        __syncthreads();
      } else {
      // ---- 1. sweeps (near an end): one thread per (row, direction), float64, the true pivots ----
      if ((int)threadIdx.x < 2 * kRows) {
        const int dl = threadIdx.x & (kRows - 1);
        const float* yr = yraw + dl * kWPitch - w0;  // yr[i] = y_i of this row
        float2* slot = tile + dl * kPitch - jmin;    // slot[i] = knot i
        // m_1, m_{n-2}: only a sweep that touches lo / hi uses them, and then the window holds
        // the three knots (w0 = 0 whenever a sweep starts at or reaches lo; likewise for hi)
        const double m1 = w0 == 0 ? (((double)yr[0] - 2.0 * (double)yr[1]) + (double)yr[2]) * (1.0 / 6.0) : 0.0;
        const double mn2 = w1 == n - 1 ? (((double)yr[n - 3] - 2.0 * (double)yr[n - 2]) + (double)yr[n - 1]) * (1.0 / 6.0) : 0.0;
        if ((int)threadIdx.x < kRows) {  // left to right: dL_i goes to knot i + 1
          const int s0 = max(lo, jmin - 1 - kFHalo), s1 = min(hi, jmax - 1);
          if (s0 <= s1) {
            double ym_ = (double)yr[s0 - 1], yc_ = (double)yr[s0], dl_ = 0.0;
            for (int i = s0; i <= s1; ++i) {
              const double yp_ = (double)yr[i + 1];
              double rr = (ym_ - 2.0 * yc_) + yp_;
              if (i == lo) rr -= m1;
              if (i == hi) rr -= mn2;
              dl_ = (rr - dl_) * qf(i - lo);
              if (i + 1 >= jmin) slot[i + 1].x = (float)dl_;
              ym_ = yc_;
              yc_ = yp_;
            }
          }
        } else {  // right to left: dR_i goes to knot i - 1
          const int e0 = min(hi, jmax + 1 + kFHalo), e1 = max(lo, jmin + 1);
          if (e0 >= e1) {
            double yp_ = (double)yr[e0 + 1], yc_ = (double)yr[e0], dr_ = 0.0;
            for (int i = e0; i >= e1; --i) {
              const double ym_ = (double)yr[i - 1];
              double rr = (ym_ - 2.0 * yc_) + yp_;
              if (i == lo) rr -= m1;
              if (i == hi) rr -= mn2;
              dr_ = (rr - dr_) * qf(hi - i);
              if (i - 1 <= jmax) slot[i - 1].y = (float)dr_;
              yp_ = yc_;
              yc_ = ym_;
            }
          }
        }
      }
      __syncthreads();
      // ---- 2. combine: (dL_{i-1}, dR_{i+1}) -> (y_i, m_i) for knots 1 .. n-2 --------------
      {
        const int dl = threadIdx.x & (kRows - 1);
        const float* yr = yraw + dl * kWPitch - w0;
        for (int kk = threadIdx.x / kRows; kk < K; kk += kBlock / kRows) {
          const int i = jmin + kk;
          float2* slot = tile + dl * kPitch + kk;
          const double yc_ = (double)yr[i];
          double m = 0.0;
          if (i >= 1 && i <= n - 2) {
            const double delta = ((double)yr[i - 1] - 2.0 * yc_) + (double)yr[i + 1];
            if (i == 1 || i == n - 2) {
              m = delta * (1.0 / 6.0);
            } else {
              const float2 s = *slot;
              double num = delta, den = 4.0;
              if (i == lo) num -= (((double)yr[0] - 2.0 * (double)yr[1]) + (double)yr[2]) * (1.0 / 6.0);
              if (i == hi) num -= (((double)yr[n - 3] - 2.0 * (double)yr[n - 2]) + (double)yr[n - 1]) * (1.0 / 6.0);
              if (i > lo) {
                num -= (double)s.x;
                den -= qf(i - 1 - lo);
              }
              if (i < hi) {
                num -= (double)s.y;
                den -= qf(hi - i - 1);
              }
              m = num / den;
            }
          }
          *slot = make_float2((float)yc_, (float)m);
        }
      }
      __syncthreads();
      if (jmin == 0 || jmax == n - 1) {  // workgroup-uniform: m_0 = 2 m_1 - m_2, m_{n-1} = 2 m_{n-2} - m_{n-3}
        if ((int)threadIdx.x < kRows) {
          float2* row = tile + threadIdx.x * kPitch - jmin;
          if (jmin == 0) row[0].y = (float)(2.0 * (double)row[1].y - (double)row[2].y);
          if (jmax == n - 1) row[n - 1].y = (float)(2.0 * (double)row[n - 2].y - (double)row[n - 3].y);
        }
        __syncthreads();
      }
      }
      // ---- 3. evaluation -------------------------------------------------------------
      SampleWeights w;
      sample_weights_at(tq, n, ta0, inv_dta, w);
      int r[kSamplesPerThread];
#pragma unroll
      for (int q = 0; q < kSamplesPerThread; ++q)
        r[q] = min(max(w.j[q] - jmin, 0), K - 2);  // in range even if t is unsorted
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
          // (nt: beside the sampler it is the policy that costs least -- 2.17 ms against 2.43 plain, 2.47 sc1, 2.28 sc1 nt)
          __builtin_nontemporal_store(
              v, reinterpret_cast<vfloat4*>(out + row_of(dl, d0 + dl) * ld + sb));
        }
      } else {
        // a sample belongs to the segment that holds its interval (every sample of a
        // single-segment tile does, whatever its interval)
        bool mine[kSamplesPerThread];
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q)
          mine[q] = sb + q < T && (single || (w.j[q] >= ja && w.j[q] < jb));
        for (int dl = 0; dl < nd; ++dl) {
          const float2* row = tile + dl * kPitch;
          const float gsc = kHasScale ? scale[d0 + dl] : 1.0f;
          float* dst = out + row_of(dl, d0 + dl) * ld + sb;
#pragma unroll
          for (int q = 0; q < kSamplesPerThread; ++q)
            if (mine[q]) dst[q] = gsc * spline_eval(w, q, row[r[q]], row[r[q] + 1]);
        }
      }
    }
  }
}


constexpr int kSynthMaxBlocks = 1024;
// the control block of a launch: [0] tile queue, [16] workgroups that have left, [32 + b] finished work items of
// block b (each on a line of its own kind: the queue is hammered by every writer, the counters by the samplers).
// All zero between launches: the last workgroup to leave clears what the launch used, so a launch needs no memset.
constexpr int kSynthCtlInts = 32 + kSynthMaxBlocks;

// Leaves the control block as it was found: the workgroup whose exit is the grid's last (atomicInc wraps the exit
// count to 0 by itself) zeroes the queue and the block counters -- nobody reads them any more, and the next launch
// on the stream starts after this one has ended.
__device__ __forceinline__ void synth_leave(int* ctl, int n_blocks) {
  __syncthreads();
  __shared__ int s_last;
  if (threadIdx.x == 0) s_last = atomicInc(reinterpret_cast<unsigned*>(ctl + 16), gridDim.x - 1) == gridDim.x - 1;
  __syncthreads();
  if (s_last) {
    if (threadIdx.x == 0) __hip_atomic_store(ctl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int b = threadIdx.x; b < n_blocks; b += kBlock)
      __hip_atomic_store(ctl + 32 + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// ---------------------------------------------------------------------------
// Evaluation fused with TOD.to("K_RJ") (tod/tod.py:106-142,
// calibration/functions.py:73-90): every sample is divided by
//     den_b(el) = (0.5 if polarized else 1) k_B  Int tau_b(nu) exp(-opacity) dnu
// looked up at the detector's own full-rate elevation (tod.py:90-93), which is
// recomputed here from the full-rate boresight elevation and the detector
// offsets exactly as coords/transforms.py:14-28 does (float32): el = asin(im),
// im = sin(r)cos(p) sin(a) + cos(r) cos(a), a = el_bore - pi/2.  den_b is the
// band's transmission-integral table collapsed by the host at the observation's
// scalar (base temperature, zenith pwv) onto the elevation axis
// (band/band.py:235-255), so the lookup is a 1-D lerp with jax's index rule.
// Same tiling as spline_upsample_kernel (knot image fixed at 256 knots).
struct CalDet {
  float a_re;  // sin(r) cos(p)
  float a_im;  // cos(r)
  int band;
  float scale;
  float dy, sdy, cdy;  // vertical offset and its sine / cosine
  // el_det - el_bore as a linear function of el_bore around the tile's middle boresight
  // elevation (it does not depend on the azimuth): el_det(s) = eb + dm + slope (eb - ebm).
  // Curvature over a tile's elevation range (~0.02 rad) is below 1e-7 rad.  exact = 1 near the
  // zenith, where the detector elevation is not smooth in eb: every sample takes the full formula.
  float dm, slope, ebm;
  int exact;
};

// detector elevation (transforms.py:20-28): im = sin(el) as the chain computes
// it, then el = asin(im) by one Newton step from el0 = el_bore + dy, whose
// sine and cosine follow from the angle-addition formulas (no inverse
// trigonometry per sample); |el - el0| <= r^2 tan(el)/2 ~ 3e-4 rad, so the
// second-order step is exact to float32 rounding.  eb: boresight elevation,
// ca / sa: cos / sin of (eb - pi/2).
__device__ __forceinline__ float det_elevation(const CalDet& c, float eb, float ca, float sa) {
  const float im = __fadd_rn(__fmul_rn(c.a_re, sa), __fmul_rn(c.a_im, ca));
  const float s0 = ca * c.cdy - sa * c.sdy;   // sin(el_bore + dy)
  const float c0 = -sa * c.cdy - ca * c.sdy;  // cos(el_bore + dy)
  const float rc0 = __builtin_amdgcn_rcpf(c0);
  const float dl1 = (im - s0) * rc0;
  float el = (eb + c.dy) + dl1 * (1.0f + 0.5f * dl1 * s0 * rc0);
  // within ~15 deg of the zenith the expansion loses accuracy: take asin there
  const bool steep = !(c0 > 0.25f);
  if (__builtin_amdgcn_ballot_w64(steep) != 0)
    if (steep) el = asinf(im);
  return el;
}

// the linear model of CalDet around the boresight elevation ebm (see CalDet)
constexpr float kModelHalfRange = 2.0e-2f;  // wide enough that float32 rounding of the differences stays below 1e-7 rad over a tile

__device__ __forceinline__ void set_elevation_model(CalDet& c, float ebm) {
  constexpr float h = kModelHalfRange;
  float e[3];
  bool steep = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float eb = ebm + (float)(k - 1) * h;
    const float a = eb - 1.57079637050628662109375f;
    const float ca = cosf(a), sa = sinf(a);
    steep |= !(-sa * c.cdy - ca * c.sdy > 0.3f);  // cos(el_bore + dy): within ~17 deg of the zenith
    e[k] = det_elevation(c, eb, ca, sa) - eb;
  }
  c.ebm = ebm;
  c.dm = e[1];
  c.slope = (e[2] - e[0]) * (0.5f / h);
  c.exact = steep ? 1 : 0;
}

__device__ __forceinline__ CalDet make_cal_det(float dx, float dy, int band, float scale) {
  const float r = sqrtf(dx * dx + dy * dy);
  const float p = atan2f(-dx, -dy);
  CalDet c;
  c.a_re = __fmul_rn(sinf(r), cosf(p));
  c.a_im = cosf(r);
  c.band = band;
  c.scale = scale;
  c.dy = dy;
  c.sdy = sinf(dy);
  c.cdy = cosf(dy);
  return c;
}

// The calibration table as the kernels hold it in LDS: per band, per cell i of the elevation
// axis one float4 (x_i, den_i, 1/(x_{i+1} - x_i), den_{i+1}), so that one 16-byte LDS read
// serves a lookup.  n_el - 1 cells per band.
__device__ __forceinline__ void stage_cal_cells(float4* cells, const float* __restrict__ axis,
                                                const float* __restrict__ values, int n_el, int n_bands) {
  const int nc = n_el - 1;
  for (int i = threadIdx.x; i < nc * n_bands; i += kBlock) {
    const int b = i / nc, k = i - b * nc;
    const float x0 = axis[k], x1 = axis[k + 1];
    cells[i] = make_float4(x0, values[b * n_el + k], 1.0f / (x1 - x0), values[b * n_el + k + 1]);
  }
}

// den at elevation el with jax's _find_indices / linear weights on the elevation axis
// (NaN off the axis): arithmetic guess from the first cell's step (am's axis is uniform but
// for its last node, which the clamp absorbs), corrected by a short walk when the guess is off
// (non-uniform axis, a sample within rounding of a node)
__device__ __forceinline__ float den_lookup(float el, const float4* C, int n_el, float el_first,
                                            float el_last, float el_inv) {
  const int nc = n_el - 1;
  int i = min(max((int)fminf(fmaxf((el - el_first) * el_inv, -1.0f), 2.0e9f), 0), nc - 1);
  while (i < nc - 1 && C[i + 1].x < el) ++i;  // searchsorted(side="left") - 1: x_i < el <= x_{i+1}
  while (i > 0 && C[i].x >= el) --i;
  const float4 c = C[i];
  const float wt = (el - c.x) * c.z;
  float den = 0.0f + c.y * (1.0f - wt);
  den = den + c.w * wt;
  return (el >= el_first && el <= el_last) ? den : __builtin_nanf("");
}

// Per-thread part of the K_RJ conversion that does not depend on the detector row.
struct KrjSamples {
  float eb0, eb3;  // boresight elevation of the thread's first and last sample
  float x0, x3;    // the same minus the tile's reference elevation (CalDet::ebm)
  int curved;      // some thread of the workgroup: its four boresight elevations are NOT linear in the sample index to 1e-6 rad
};

// The K_RJ values of a thread's 4 consecutive samples of one detector.  den is piecewise
// linear in the elevation, and the elevation is linear in the sample index to ~5e-8 rad over 4
// samples (10 ms of scanning, over which den itself moves by ~3e-6 of its value): when the
// first and last sample share a cell of the axis, den -- or its reciprocal, equal to second
// order, 1e-11 -- of the inner two is interpolated between the outer ones (float32 rounding
// apart, the value jax computes); otherwise -- a node between them, a guess that missed, an
// elevation off the axis -- every sample is looked up on its own at the interpolated
// elevation.  `sv` already carries the detector's scale.
template <bool kInverse = false, bool kCurved = false>
__device__ __forceinline__ void krj_row(const CalDet& c, const float4* C, int n_el, float el_first,
                                        float el_last, float el_inv, const KrjSamples& k,
                                        const float (&sv)[kSamplesPerThread], float (&o)[kSamplesPerThread],
                                        const float* __restrict__ bore_el, int sb, int T) {
  constexpr int kL = kSamplesPerThread - 1;
  if constexpr (kCurved) {
    // Low sample rates or tight fast scans (20 Hz, a 0.1 deg daisy at 0.8 deg/s: 3e-4 rad of curvature over a thread's
    // four samples, 1e-4 of den) -- every sample at its own boresight elevation.  The workgroup takes this instance of
    // its row loop when any of its threads sees more than 1e-6 rad (KrjSamples::curved); never at the rates the
    // interpolation below was built for (400 Hz: 6e-8 rad).
    // (one sample at a time, its elevation reloaded: unrolled, or with the four values kept in registers, this
    // instance would set the kernel's register count -- 98 instead of 96 costs a wave per SIMD and 12 %)
#pragma unroll 1
    for (int q = 0; q < kSamplesPerThread; ++q) {
      const float ebq = bore_el[min(sb + q, T - 1)];
      float el;
      if (c.exact) {
        // (the hardware sine and cosine, in revolutions: 1e-6 rad here, where den hardly moves with the elevation;
        // cosf / sinf inlined would set the kernel's register count)
        const float rev = (ebq - 1.57079637050628662109375f) * 0.15915494309189535f;
        el = det_elevation(c, ebq, __builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev));
      } else {
        el = fmaf(c.slope, ebq - c.ebm, ebq + c.dm);
      }
      const float den = den_lookup(el, C, n_el, el_first, el_last, el_inv);
      const float val = kInverse ? sv[q] * den : sv[q] * __builtin_amdgcn_rcpf(den);
      o[0] = q == 0 ? val : o[0];
      o[1] = q == 1 ? val : o[1];
      o[2] = q == 2 ? val : o[2];
      o[3] = q == 3 ? val : o[3];
    }
    return;
  }
  float e0, e3;
  if (c.exact) {  // uniform over the workgroup (one detector row at a time), and rare
    // (sine and cosine of the boresight elevation on the spot, by the hardware instructions in revolutions -- 1e-6 rad,
    // where den hardly moves with the elevation: kept per thread for every row they were four registers of a kernel
    // that sits at a wave-per-SIMD boundary)
    const float r0 = (k.eb0 - 1.57079637050628662109375f) * 0.15915494309189535f, r3 = (k.eb3 - 1.57079637050628662109375f) * 0.15915494309189535f;
    e0 = det_elevation(c, k.eb0, __builtin_amdgcn_cosf(r0), __builtin_amdgcn_sinf(r0));
    e3 = det_elevation(c, k.eb3, __builtin_amdgcn_cosf(r3), __builtin_amdgcn_sinf(r3));
  } else {
    e0 = fmaf(c.slope, k.x0, k.eb0 + c.dm);
    e3 = fmaf(c.slope, k.x3, k.eb3 + c.dm);
  }
  const int i0 = min(max((int)fminf(fmaxf((e0 - el_first) * el_inv, -1.0f), 2.0e9f), 0), n_el - 2);
  const float4 cell = C[i0];
  const float w0 = (e0 - cell.x) * cell.z, w3 = (e3 - cell.x) * cell.z;
  // both ends inside cell i0 (0 < w <= 1; the first cell closed below) and on the axis.  A
  // sample within rounding of a node may be taken for either neighbour: den is continuous there.
  const float wmin = fminf(w0, w3), wmax = fmaxf(w0, w3);
  const bool fast = (wmin > 0.0f || (i0 == 0 && wmin >= 0.0f)) && wmax <= 1.0f &&
                    fminf(e0, e3) >= el_first && fmaxf(e0, e3) <= el_last;
  float d0 = 0.0f + cell.y * (1.0f - w0);
  d0 = d0 + cell.w * w0;
  float d3 = 0.0f + cell.y * (1.0f - w3);
  d3 = d3 + cell.w * w3;
  if (kInverse) {
    const float step = (d3 - d0) * (1.0f / (float)kL);
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) o[q] = sv[q] * (q == 0 ? d0 : q == kL ? d3 : fmaf((float)q, step, d0));
  } else {
    const float r0 = __builtin_amdgcn_rcpf(d0), r3 = __builtin_amdgcn_rcpf(d3);
    const float step = (r3 - r0) * (1.0f / (float)kL);
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) o[q] = sv[q] * (q == 0 ? r0 : q == kL ? r3 : fmaf((float)q, step, r0));
  }
  if (__builtin_amdgcn_ballot_w64(!fast) != 0) {
    if (!fast) {
      const float de = (e3 - e0) * (1.0f / (float)kL);
#pragma unroll 1
      for (int q = 0; q < kSamplesPerThread; ++q) {
        const float el = q == 0 ? e0 : q == kL ? e3 : e0 + (float)q * de;
        const float den = den_lookup(el, C, n_el, el_first, el_last, el_inv);
        o[q] = kInverse ? sv[q] * den : sv[q] * __builtin_amdgcn_rcpf(den);
      }
    }
  }
}

// Shared prologue of the two K_RJ kernels, per workgroup: stage the cell table, reduce the
// boresight elevation range of the tile's 1024 samples (red[8] = lo, red[9] = hi) and return
// this thread's sample constants.  Ends with a barrier.
__device__ __forceinline__ KrjSamples krj_prologue(float4* cells, float* red, const float* __restrict__ bore_el,
                                                   int T, int sb, const float* __restrict__ cal_axis,
                                                   const float* __restrict__ cal_values, int n_el, int n_bands) {
  KrjSamples k;
  k.eb0 = bore_el[min(sb, T - 1)];
  k.eb3 = bore_el[min(sb + kSamplesPerThread - 1, T - 1)];
  float eb_lo, eb_hi;
  {
    static_assert(kSamplesPerThread == 4, "the curvature check below is written for four samples");
    const float eb1 = bore_el[min(sb + 1, T - 1)], eb2 = bore_el[min(sb + 2, T - 1)];
    const float third = (k.eb3 - k.eb0) * (1.0f / 3.0f);
    k.curved = !(fabsf(eb1 - (k.eb0 + third)) <= 1.0e-6f && fabsf(eb2 - (k.eb0 + 2.0f * third)) <= 1.0e-6f);  // (a NaN: per sample too)
    eb_lo = fminf(fminf(k.eb0, eb1), fminf(eb2, k.eb3));
    eb_hi = fmaxf(fmaxf(k.eb0, eb1), fmaxf(eb2, k.eb3));
  }
  // (the samples are monotone enough that the ends of the threads' 4-sample runs bound the
  // range to ~1e-7 rad): lanes -> waves -> workgroup
  float lo = eb_lo, hi = eb_hi;
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, m, 64));
    hi = fmaxf(hi, __shfl_xor(hi, m, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    red[2 * (threadIdx.x >> 6)] = lo;
    red[2 * (threadIdx.x >> 6) + 1] = hi;
  }
  stage_cal_cells(cells, cal_axis, cal_values, n_el, n_bands);
  __syncthreads();
  lo = fminf(fminf(red[0], red[2]), fminf(red[4], red[6]));
  hi = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
  const float ebm = 0.5f * (lo + hi);
  k.x0 = k.eb0 - ebm;
  k.x3 = k.eb3 - ebm;
  if (threadIdx.x == 0) {
    red[8] = lo;
    red[9] = hi;
  }
  k.curved = __syncthreads_or(k.curved);  // (the barrier this prologue ends with)
  return k;
}

// Per group of 16 detector rows: their constants and the tile's elevation model.  The caller
// puts a barrier between this and the rows' use of cdet[].  Returns nothing; cdet[16].exact
// of the LAST entry's neighbour slot red[10] is set when any row needs the full formula.
__device__ __forceinline__ void krj_stage_rows(CalDet* cdet, float* red, const float* __restrict__ dxs,
                                               const float* __restrict__ dys, const int32_t* __restrict__ band,
                                               const float* __restrict__ scale, int n_bands, int d0, int nd) {
  if ((int)threadIdx.x < kTileDet) {
    const float lo = red[8], hi = red[9];
    bool exact = false;
    if ((int)threadIdx.x < nd) {
      const int d = d0 + threadIdx.x;
      CalDet c = make_cal_det(dxs[d], dys[d], min(max(band[d], 0), n_bands - 1), scale ? scale[d] : 1.0f);
      set_elevation_model(c, 0.5f * (lo + hi));
      // the model is a finite difference over ebm +- 0.02 rad: a tile whose boresight sweeps
      // farther (slow sample rates, fast elevation slews) takes the full formula per sample
      if (!(hi - lo <= 2.0f * kModelHalfRange)) c.exact = 1;
      cdet[threadIdx.x] = c;
      exact = c.exact != 0;
    }
    const bool any = __builtin_amdgcn_ballot_w64(exact) != 0;  // the 16 lanes sit in wave 0
    if (threadIdx.x == 0) red[10] = any ? 1.0f : 0.0f;
  }
}

// the static part of the K_RJ writer's LDS, carved from the dynamic buffer behind the cell table (with a static size the
// compiler derives the occupancy from it and ignores the register bound below, as for the fused writer)
struct KrjWriterLds {
  static constexpr int kMaxKnots = 64, kPitch = kMaxKnots + 1;
  float2 tile[kTileDet * kPitch];
  CalDet cdet[kTileDet];
  float red[12];
  int row_lds[kTileDet];
};

// 5 waves per SIMD = 96 registers: the evaluation loop needs 95; the per-sample instance of the loop (KrjSamples::curved,
// rare) would take the kernel to 98 and a wave per SIMD away (K_RJ writer 2.6 -> 3.0 ms), so it spills what is over
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5))) void spline_upsample_krj_kernel(
    const float2* __restrict__ ym, int D, int n, double ta0, double inv_dta,
    const double* __restrict__ t, int T, const float* __restrict__ scale,
    const int32_t* __restrict__ rows, const float* __restrict__ bore_el,
    const float* __restrict__ dxs, const float* __restrict__ dys,
    const int32_t* __restrict__ band, const float* __restrict__ cal_axis,
    const float* __restrict__ cal_values, int n_el, int n_bands,
    float* __restrict__ out, size_t ld, int vec_ok, int groups) {
  constexpr int kMaxKnots = KrjWriterLds::kMaxKnots;  // 8 KiB image: the arithmetic of this writer wants the occupancy
  constexpr int kPitch = KrjWriterLds::kPitch;
  extern __shared__ __align__(16) float4 cal_cells[];  // [n_bands][n_el - 1], see stage_cal_cells; then KrjWriterLds
  KrjWriterLds& L = *reinterpret_cast<KrjWriterLds*>(cal_cells + (size_t)n_bands * (n_el - 1));
  float2* tile = L.tile;
  CalDet* cdet = L.cdet;
  float* red = L.red;
  int* row_lds = L.row_lds;  // destination rows of the group (see spline_upsample_kernel)
  auto row_of = [&](int dl, int d) -> size_t { return rows ? (size_t)row_lds[dl] : (size_t)d; };

  const int s_tile = blockIdx.x * kTileSamples;
  const int sb = s_tile + threadIdx.x * kSamplesPerThread;

  // per-sample interval, weights and boresight: computed once, reused for `groups` tiles of
  // 16 detector rows each (the float64 prologue is a third of a single tile's instructions)
  SampleWeights w;
  sample_weights(t, sb, T, n, ta0, inv_dta, w);
  const KrjSamples ks = krj_prologue(cal_cells, red, bore_el, T, sb, cal_axis, cal_values, n_el, n_bands);

  const int s_last = min(s_tile + kTileSamples, T) - 1;
  const int jmin = interval_of((t[s_tile] - ta0) * inv_dta, n);
  const int jmax = interval_of((t[s_last] - ta0) * inv_dta, n) + 1;
  const int K = jmax - jmin + 1;
  const bool use_lds = K <= kMaxKnots;
  const float el_first = cal_cells[0].x, el_last = cal_axis[n_el - 1];
  const float el_inv = cal_cells[0].z;
  const bool full = (sb + kSamplesPerThread <= T) && vec_ok;
  int r[kSamplesPerThread];
#pragma unroll
  for (int q = 0; q < kSamplesPerThread; ++q) r[q] = min(max(w.j[q] - jmin, 0), max(K - 2, 0));

  for (int g = 0; g < groups; ++g) {
    const int d0 = (blockIdx.y * groups + g) * kTileDet;
    if (d0 >= D) break;
    const int nd = min(kTileDet, D - d0);
    if (g > 0) __syncthreads();  // the previous group is done with tile[] and cdet[]
    if (use_lds) {
      const int dl = threadIdx.x & (kTileDet - 1);
      const int d = d0 + dl;
      for (int rr = threadIdx.x / kTileDet; rr < K; rr += kBlock / kTileDet) {
        float2 v = make_float2(0.f, 0.f);
        if (d < D) v = ym[(size_t)(jmin + rr) * D + d];
        tile[dl * kPitch + rr] = v;
      }
    }
    krj_stage_rows(cdet, red, dxs, dys, band, scale, n_bands, d0, nd);
    if (rows && (int)threadIdx.x < nd) row_lds[threadIdx.x] = rows[d0 + threadIdx.x];
    __syncthreads();
    // the loop body is instantiated once per knot source so that each instance
    // addresses one memory space (a runtime select would force flat loads)
    auto body = [&](auto from_lds, auto curved) {
    for (int dl = 0; dl < nd; ++dl) {
      const CalDet c = cdet[dl];
      const float4* C = cal_cells + c.band * (n_el - 1);
      float o[kSamplesPerThread], sv[kSamplesPerThread];
#pragma unroll
      for (int q = 0; q < kSamplesPerThread; ++q) {
        float2 k0, k1;
        if constexpr (decltype(from_lds)::value) {
          k0 = tile[dl * kPitch + r[q]];
          k1 = tile[dl * kPitch + r[q] + 1];
        } else {
          k0 = ym[(size_t)w.j[q] * D + d0 + dl];
          k1 = ym[(size_t)(w.j[q] + 1) * D + d0 + dl];
        }
        sv[q] = c.scale * spline_eval(w, q, k0, k1);
      }
      krj_row<false, decltype(curved)::value>(c, C, n_el, el_first, el_last, el_inv, ks, sv, o, bore_el, sb, T);
      float* dst = out + row_of(dl, d0 + dl) * ld + sb;
      if (full) {
        const vfloat4 v = {o[0], o[1], o[2], o[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(dst));
      } else {
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q)
          if (sb + q < T) dst[q] = o[q];
      }
    }
    };
    if (ks.curved) {  // (uniform)
      if (use_lds) body(std::true_type{}, std::true_type{}); else body(std::false_type{}, std::true_type{});
    } else {
      if (use_lds) body(std::true_type{}, std::false_type{}); else body(std::false_type{}, std::false_type{});
    }
  }
}

// ---------------------------------------------------------------------------
// Atmosphere -> TOD in ONE launch (mrx_atm_synthesize): the sampler and the writer as two ROLES of one grid, the
// hand-over on the device.  Replaces, for one observation, screens -> [sampler of block b on a side stream |
// writer of block b behind an event] x blocks: there the writer of block 0 cannot start before the whole first
// block is sampled (0.2 ms of the 2.3-ms step of atlast_10k with HBM idle) and every launch boundary drains and
// refills the chip (4-14 per step).
//
//   * the first `n_sampler_wgs` workgroups (lowest block indices: the dispatcher hands them out first, so they are
//     resident before any writer -- and they never wait for anything, so the grid always drains) run
//     px_sample_items over the detector blocks in order: block b's coarse loading is its own [Ta][rows] array;
//   * the loading leaves the sampler's CUs as write-through (sc1) 16-byte stores, every 128-byte line written whole
//     by one store instruction (px_sample_items<..., kWriteThrough>); after each finished work item every wave
//     drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, and ONE lane adds 1 to done[b]
//     (agent scope, relaxed);
//   * every other workgroup is a writer: it takes tile numbers from a queue (one agent-scope atomic per tile; tiles
//     are numbered block by block, time tile fastest -- the order the two-dimensional grid of
//     spline_upsample_fused_kernel is dispatched in), and before its FIRST tile of a block one lane polls done[b]
//     (global_load_dword sc1, s_sleep between) until all of the block's items are in, then the workgroup meets at a
//     barrier; the tile stages its knots with sc1 loads (past the CU's L1), then fused_writer_tile as in the
//     stand-alone kernel.
// The per-XCD L2s are not coherent and a CU's L1 is not refreshed by other CUs' stores.  Write-through stores +
// drained waves + one agent-scope add per workgroup on the producer's side, an sc1 poll + a workgroup barrier + sc1
// loads on the consumer's, is one of the forms MI355X_MICROARCH.md lists as measured valid on gfx950 (its table of
// hand-offs without fences, third row).  The first version used the fenced form (release fence per work item, acquire
// per writer and block): correct too, but a release writes back the XCD's whole L2 under a streaming writer --
// 0.18 ms of a 1.9-ms step in fences, 0.15-0.3 more in waits.
// A poll gives up after `poll_limit` tries and raises MRX_FLAG_HANDOVER (the host then fails the call): a bound,
// not a path -- it cannot trigger while the sampler role is resident.
// Results: the same bits as mrx_atm_sample + mrx_spline_upsample_fused per block (same bodies, same order of
// operations; tests/test_gpu_synthesize.py).
// kKrj (mrx_atm_synthesize_krj): TOD.to("K_RJ") on the coarse grid, in the sampler role's epilogue -- what
// coarse_krj_kernel does to a finished block between the two calls (same functions, same operands: the same bits),
// the last knots kept aside in pW for the samples past the last knot.
// who samples what: blocks [end[p-1], end[p]) by the first wgs[p] sampler workgroups (wgs descending)
struct SynthPhases {
  int n;
  int end[4];
  int wgs[4];
};

// The calibration of the K_RJ form (DevicePath.set_calibration: the band's denominators on the elevation axis).
struct SynthCal {
  const float* dx;      // [D] detector offsets as the calibration holds them
  const float* dy;
  const float* axis;    // [n_el]
  const float* values;  // [n_bands][n_el]
  int n_el, n_bands;
  float* tail;          // [tail_knots][ld_tail] the last knots in pW, or null
  int tail_first;       // Ta - tail_knots
  size_t ld_tail;
};

// What the sampler role does beside sampling: the K_RJ division per coarse sample and the block counters.
template <bool kKrj>
struct SynthHooks {
  int* ctl;
  const int32_t* band;
  SynthCal cal;
  const float4* cells;  // the staged table (stage_cal_cells)
  float el_first, el_last, el_inv;
  float a_re = 0.0f, a_im = 0.0f;  // of the work item's detector (make_cal_det)
  const float4* C = nullptr;
  __device__ __forceinline__ void item(int, int d) {
    if (kKrj) {
      const CalDet c = make_cal_det(cal.dx[d], cal.dy[d], min(max(band[d], 0), cal.n_bands - 1), 1.0f);
      a_re = c.a_re;
      a_im = c.a_im;
      C = cells + c.band * (cal.n_el - 1);
    }
  }
  // coarse_krj_kernel's arithmetic: im = sin(el_det) from the step's (cos, sin) of (boresight elevation - pi/2)
  __device__ __forceinline__ float value(float v, const float4& bt, int t, int d, bool real) const {
    if (!kKrj) return v;
    const float im = __fadd_rn(__fmul_rn(a_re, bt.y), __fmul_rn(a_im, bt.x));
    const float den = den_lookup(asinf(im), C, cal.n_el, el_first, el_last, el_inv);
    if (cal.tail && real && t >= cal.tail_first) cal.tail[(size_t)(t - cal.tail_first) * cal.ld_tail + d] = v;
    return v * __builtin_amdgcn_rcpf(den);
  }
  __device__ __forceinline__ void done(int blk) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's (write-through) stores of the item are out
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctl + 32 + blk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

template <bool kLdsTables, bool kHasScale, int kMaxKnots, int kG, bool kKrj>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MRX_WRITER_WAVES, MRX_WRITER_WAVES))) void atm_tod_kernel(
    const mrx_layer_fast* __restrict__ fast, const mrx_layer_px* __restrict__ lpx, int n_layers,
    const double2* __restrict__ offpx, const mrx_table_dev* __restrict__ tables, int n_tables,
    const float* __restrict__ table_data, int table_floats, const float* __restrict__ az,
    const float* __restrict__ el, int Ta, const float* __restrict__ dxs, const float* __restrict__ dys,
    const int32_t* __restrict__ band, const float* __restrict__ mueller00, int D, double pwv0,
    float* loading,  // written by the sampler role, read by the writer role: no __restrict__, no const
    uint32_t* __restrict__ flags, int chunk, int nby, int block_rows, int n_blocks, int n_sampler_wgs,
    SynthPhases phases, double ta0, double inv_dta, const double* __restrict__ t, int T,
    const float* __restrict__ scale, const int32_t* __restrict__ rows, float* __restrict__ out, size_t ld, int vec_ok,
    int batches, int* ctl, int poll_limit, SynthCal cal) {
  extern __shared__ __align__(16) unsigned char synth_lds[];
  if ((int)blockIdx.x < n_sampler_wgs) {
    SynthHooks<kKrj> hooks;
    hooks.ctl = ctl;
    hooks.band = band;
    hooks.cal = cal;
    if (kKrj) {
      // the cell table behind the anchors, the band tables and the four turned steps (16-byte aligned)
      float4* cells = reinterpret_cast<float4*>(synth_lds) + 2 * chunk * n_layers + (kLdsTables ? (table_floats + 3) / 4 : 0) + kBlock;
      stage_cal_cells(cells, cal.axis, cal.values, cal.n_el, cal.n_bands);
      __syncthreads();
      hooks.cells = cells;
      hooks.el_first = cells[0].x;
      hooks.el_last = cal.axis[cal.n_el - 1];
      hooks.el_inv = cells[0].z;
    }
    // phases: the first blocks by ALL sampler workgroups (nothing else is resident yet: the chip is theirs), the next
    // ones by fewer and fewer of them -- those that leave make room for writers --, the rest by the first few
    int b0 = 0;
    for (int ph = 0; ph < phases.n; ++ph) {
      const int nw = phases.wgs[ph];
      if ((int)blockIdx.x >= nw) break;
      const int b1 = phases.end[ph];
      if (b1 > b0)
        mrx_px::px_sample_items<kLdsTables, 1, true, true>(
            fast, lpx, n_layers, offpx, tables, n_tables, table_data, table_floats, az, el, Ta, dxs, dys, band, mueller00,
            D, pwv0, nullptr, loading, flags, chunk, nby, block_rows, n_blocks, b0, b1, (int)blockIdx.x, nw,
            reinterpret_cast<float4*>(synth_lds), hooks);
      b0 = max(b0, b1);
    }
    synth_leave(ctl, n_blocks);
    return;
  }
  constexpr int kRows = kTileDet * kG;
  __shared__ int s_next;
  const int nsx = (T + kTileSamples - 1) / kTileSamples;
  const int rows_per_tile = kRows * batches;
  const int last_rows = D - (n_blocks - 1) * block_rows;
  const int tiles_full = nsx * ((block_rows + rows_per_tile - 1) / rows_per_tile);
  const int total = tiles_full * (n_blocks - 1) + nsx * ((last_rows + rows_per_tile - 1) / rows_per_tile);
  int have = -1;  // blocks up to this one are known to be sampled
  for (;;) {
    if (threadIdx.x == 0) s_next = __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();  // (also: the previous tile's readers of the LDS images are done)
    const int tile = s_next;
    if (tile >= total) break;
    const int blk = min(tile / tiles_full, n_blocks - 1);
    const int rem = tile - blk * tiles_full;
    const int by = rem / nsx, sx = rem - by * nsx;
    const int Db = blk == n_blocks - 1 ? last_rows : block_rows;
    if (blk > have) {  // workgroup-uniform
      if (threadIdx.x == 0) {
        const int want = nby * ((Db + mrx_px::kPxBlock - 1) / mrx_px::kPxBlock);
        int tries = 0;
        while (__hip_atomic_load(ctl + 32 + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          if (++tries > poll_limit) {
            atomicOr(flags, MRX_FLAG_HANDOVER);
            break;
          }
          __builtin_amdgcn_s_sleep(32);
        }
      }
      __syncthreads();  // between the poll and EVERY load of the block's bytes, the polling wave's own too
      have = blk;
    }
    const size_t row0 = (size_t)blk * block_rows;
    fused_writer_tile<kHasScale, kMaxKnots, kG, true>(loading + (size_t)Ta * row0, (Db + 31) & ~31, Db, Ta, ta0, inv_dta, t, T,
                                                       kHasScale ? scale + row0 : nullptr, rows ? rows + row0 : nullptr,
                                                       rows ? out : out + row0 * ld, ld, vec_ok, batches, sx, by, synth_lds);
  }
  synth_leave(ctl, n_blocks);
}

// TOD.to("K_RJ") on the COARSE grid, before the spline: out[j][d] = loading[j][d] /
// den_band(d)(el_det(d, j)) with the detector elevation of the full formula at coarse step j.
// The reference divides the full-rate spline S[y](t) by g(t) = den(el_det(t)) sample by sample;
// S[y / g] differs from S[y] / g by the spline's interpolation error on g alone (g is smooth in
// time but for the kinks where the elevation crosses a node of the table's axis): the host
// bounds it before choosing this form (DevicePath.coarse_krj_bound) and the TOD is then written
// by the plain pW writer.  Time-major like the sampler's output: lanes are detectors.
constexpr int kCoarseKrjSteps = 32;

// (`loading` and `out` may be the same buffer -- the pipelined run converts a block's coarse loading in place --: no
// __restrict__ on either; every element is read once, by the thread that then writes it)
__global__ __launch_bounds__(kBlock) void coarse_krj_kernel(
    const float* loading, int D, int Ta, const float* __restrict__ bore_el,
    const float* __restrict__ dxs, const float* __restrict__ dys, const int32_t* __restrict__ band,
    const float* __restrict__ cal_axis, const float* __restrict__ cal_values, int n_el, int n_bands,
    float* out, float* __restrict__ tail, int tail_first, size_t ld_tail) {
  extern __shared__ __align__(16) float4 cal_cells[];  // [n_bands][n_el - 1], see stage_cal_cells
  __shared__ float2 trig[kCoarseKrjSteps];  // (cos, sin) of (boresight elevation - pi/2) of the block's steps
  stage_cal_cells(cal_cells, cal_axis, cal_values, n_el, n_bands);
  const int j0 = blockIdx.y * kCoarseKrjSteps, j1 = min(j0 + kCoarseKrjSteps, Ta);
  if ((int)threadIdx.x < j1 - j0) {
    const float a = bore_el[j0 + threadIdx.x] - 1.57079637050628662109375f;
    trig[threadIdx.x] = make_float2(cosf(a), sinf(a));
  }
  __syncthreads();
  const int d = blockIdx.x * kBlock + threadIdx.x;
  if (d >= D) return;
  const CalDet c = make_cal_det(dxs[d], dys[d], min(max(band[d], 0), n_bands - 1), 1.0f);
  const float4* C = cal_cells + c.band * (n_el - 1);
  const float el_first = cal_cells[0].x, el_last = cal_axis[n_el - 1], el_inv = cal_cells[0].z;
  // eight steps' loads in flight per thread (one at a time, the loop was a chain of 32 memory latencies: 65 us for a block
  // of 2 500 rows beside the TOD writer, on the sampler's stream of the pipelined step)
  constexpr int kAhead = 8;
  for (int jb = j0; jb < j1; jb += kAhead) {
    float v[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) v[k] = loading[(size_t)min(jb + k, j1 - 1) * D + d];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      const int j = jb + k;
      if (j >= j1) break;
      // coords/transforms.py:20-28 in float32: im = sin(el_det), el_det = asin(im)
      const float2 cs = trig[j - j0];
      const float im = __fadd_rn(__fmul_rn(c.a_re, cs.y), __fmul_rn(c.a_im, cs.x));
      const float den = den_lookup(asinf(im), C, n_el, el_first, el_last, el_inv);
      out[(size_t)j * D + d] = v[k] * __builtin_amdgcn_rcpf(den);
      if (tail && j >= tail_first) tail[(size_t)(j - tail_first) * ld_tail + d] = v[k];  // (uniform: j is the block's)
    }
  }
}

// TOD.to("K_RJ") of a field that is already at the full rate (noise, map, cmb;
// tod/tod.py:106-142), in place: data[row(d)][s] *= scale_d / den_band(d)(el(d, s)).
// Same tile as the fused writer: 16 detectors x 1024 samples per workgroup, 16-byte
// loads and non-temporal stores; 8 B of HBM traffic per sample.
template <bool kInverse>
__global__ __launch_bounds__(kBlock) void tod_krj_kernel(
    float* __restrict__ data, size_t ld, int D, int T, const float* __restrict__ scale,
    const int32_t* __restrict__ rows, const float* __restrict__ bore_el,
    const float* __restrict__ dxs, const float* __restrict__ dys,
    const int32_t* __restrict__ band, const float* __restrict__ cal_axis,
    const float* __restrict__ cal_values, int n_el, int n_bands, int vec_ok) {
  extern __shared__ __align__(16) float4 cal_cells[];
  __shared__ CalDet cdet[kTileDet];
  __shared__ float red[12];
  const int s_tile = blockIdx.x * kTileSamples;
  const int d0 = blockIdx.y * kTileDet;
  const int sb = s_tile + threadIdx.x * kSamplesPerThread;
  const int nd = min(kTileDet, D - d0);
  __shared__ int row_lds[kTileDet];
  KrjSamples ks = krj_prologue(cal_cells, red, bore_el, T, sb, cal_axis, cal_values, n_el, n_bands);
  krj_stage_rows(cdet, red, dxs, dys, band, scale, n_bands, d0, nd);
  if ((int)threadIdx.x < nd) row_lds[threadIdx.x] = rows ? rows[d0 + threadIdx.x] : d0 + (int)threadIdx.x;
  __syncthreads();
  if (sb >= T) return;
  const float el_first = cal_cells[0].x, el_last = cal_axis[n_el - 1];
  const float el_inv = cal_cells[0].z;
  const bool full = (sb + kSamplesPerThread <= T) && vec_ok;
  auto rows_loop = [&](auto curved) {
  for (int dl = 0; dl < nd; ++dl) {
    const CalDet c = cdet[dl];
    const float4* C = cal_cells + c.band * (n_el - 1);
    float* row = data + (size_t)row_lds[dl] * ld + sb;
    float v[kSamplesPerThread];
    if (full) {
      const vfloat4 x = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(row));
      v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    } else {
#pragma unroll
      for (int q = 0; q < kSamplesPerThread; ++q) v[q] = sb + q < T ? row[q] : 0.0f;
    }
    float sv[kSamplesPerThread];
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) sv[q] = c.scale * v[q];
    krj_row<kInverse, decltype(curved)::value>(c, C, n_el, el_first, el_last, el_inv, ks, sv, v, bore_el, sb, T);
    if (full) {
      const vfloat4 x = {v[0], v[1], v[2], v[3]};
      __builtin_nontemporal_store(x, reinterpret_cast<vfloat4*>(row));
    } else {
#pragma unroll
      for (int q = 0; q < kSamplesPerThread; ++q)
        if (sb + q < T) row[q] = v[q];
    }
  }
  };
  if (ks.curved) rows_loop(std::true_type{}); else rows_loop(std::false_type{});  // (uniform)
}


// ---- the writer of the two-rate noise generator (mrx_noise.hip: noise_generate_two_rate) -------------------
// noise[d,t] = amp(d,t) ( sqrt(fs) w[d,t] + sqrt(c) sqrt(fs) sum_m B[d,m] w'[m,t] + P_d(t) ) [/ den(el(d,t)): K_RJ]
// White noise is drawn here, per sample (Philox keyed by row and sample index, as the white-only path draws it);
// the modes' white parts come from a table of n_modes unit series shared by all detectors; P_d -- the detector's own
// pink part plus the modes' pink parts -- was synthesised in the frequency domain at fs / rate (its spectrum above
// that Nyquist frequency holds less than 2 % of the white level: mrx_noise.hip picks the rate so) and is
// interpolated to the full rate with the four-point Catmull-Rom cubic; the K_RJ division is the one of
// mrx_tod_to_krj (same per-tile elevation model, same lookup).  Tile: 16 rows x 1024 samples; a thread owns 4
// consecutive samples -- one interval of the slow series at rate 4, two at rate 2.
typedef float nvfloat4 __attribute__((ext_vector_type(4)));
typedef float nvfloat4u __attribute__((ext_vector_type(4), aligned(4)));  // 16 bytes at any 4-byte address: one global_load_dwordx4

template <bool kKrj, int kModes>  // kModes: 0, 5 (up to five modes: the reference's spatial basis) or 8
__global__ __launch_bounds__(kBlock) void noise_two_rate_kernel(const mrx_two_rate_args a) {
  extern __shared__ __align__(16) float4 cal_cells[];  // K_RJ: [n_bands][n_el - 1]
  __shared__ CalDet cdet[kTileDet];
  __shared__ float red[12];
  __shared__ float coef[kTileDet][8];  // w_corr sqrt(fs) B[row][m]
  const int s_tile = blockIdx.x * kTileSamples;
  const int r0 = blockIdx.y * kTileDet;  // within the launch
  const int nd = min(kTileDet, a.rows - r0);
  const int sb = s_tile + threadIdx.x * kSamplesPerThread;
  KrjSamples ks{};
  if constexpr (kKrj) {
    ks = krj_prologue(cal_cells, red, a.bore_el, a.T, sb, a.cal_axis, a.cal_values, a.n_el, a.n_bands);
    krj_stage_rows(cdet, red, a.dx, a.dy, a.band, nullptr, a.n_bands, a.row0 + r0, nd);
  }
  for (int i = threadIdx.x; i < kTileDet * 8; i += kBlock) {
    const int dl = i >> 3, m = i & 7;
    coef[dl][m] = (dl < nd && m < a.n_modes) ? a.w_corr * a.sqrt_fs * a.basis[(size_t)(a.row0 + r0 + dl) * a.n_modes + m] : 0.0f;
  }
  __syncthreads();
  if (sb >= a.T) return;
  // the modes' white parts of this thread's four samples
  float mw[kModes > 0 ? kModes : 1][kSamplesPerThread];
#pragma unroll
  for (int m = 0; m < kModes; ++m) {
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) mw[m][q] = 0.0f;
    if (m < a.n_modes) {  // (uniform)
      const float* src = a.mode_white + (size_t)m * a.ld_mw + sb;  // ld_mw and the buffer are padded to whole groups of 4
      const nvfloat4 v = *reinterpret_cast<const nvfloat4*>(src);
      mw[m][0] = v[0]; mw[m][1] = v[1]; mw[m][2] = v[2]; mw[m][3] = v[3];
    }
  }
  const bool full = sb + kSamplesPerThread <= a.T && (a.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
  const float el_first = kKrj ? cal_cells[0].x : 0.0f, el_last = kKrj ? a.cal_axis[a.n_el - 1] : 0.0f, el_inv = kKrj ? cal_cells[0].z : 0.0f;
  // Catmull-Rom weights of (P[-1], P[0], P[1], P[2]) at u = 1/4, 1/2, 3/4 (u = 0: P[0] itself)
  constexpr float kW14[4] = {-0.0703125f, 0.8671875f, 0.2265625f, -0.0234375f};
  constexpr float kW12[4] = {-0.0625f, 0.5625f, 0.5625f, -0.0625f};
  constexpr float kW34[4] = {-0.0234375f, 0.2265625f, 0.8671875f, -0.0703125f};
  auto rows_loop = [&](auto curved) {
#pragma unroll 2
    for (int dl = 0; dl < nd; ++dl) {
      const int row = a.row0 + r0 + dl;  // row of the call
      // the slow part: samples t' - 1 .. t' + 2 (.. t' + 3 at rate 2) around the thread's interval(s), stored one to the right
      const float* lo = a.lo + (size_t)(r0 + dl) * a.ld_lo;
      float p[kSamplesPerThread];
      if (a.rate == 4) {
        const nvfloat4u q = *reinterpret_cast<const nvfloat4u*>(lo + (sb >> 2));
        const float p0 = q[0], p1 = q[1], p2 = q[2], p3 = q[3];
        p[0] = p1;
        p[1] = kW14[0] * p0 + kW14[1] * p1 + kW14[2] * p2 + kW14[3] * p3;
        p[2] = kW12[0] * p0 + kW12[1] * p1 + kW12[2] * p2 + kW12[3] * p3;
        p[3] = kW34[0] * p0 + kW34[1] * p1 + kW34[2] * p2 + kW34[3] * p3;
      } else {
        const nvfloat4u q = *reinterpret_cast<const nvfloat4u*>(lo + (sb >> 1));
        const float p0 = q[0], p1 = q[1], p2 = q[2], p3 = q[3], p4 = lo[(sb >> 1) + 4];
        p[0] = p1;
        p[1] = kW12[0] * p0 + kW12[1] * p1 + kW12[2] * p2 + kW12[3] * p3;
        p[2] = p2;
        p[3] = kW12[0] * p1 + kW12[1] * p2 + kW12[2] * p3 + kW12[3] * p4;
      }
      const mrx_dev::U4 rnd = mrx_dev::philox4x32_10(
          mrx_dev::U4{(uint32_t)(sb >> 2), a.id0 + (uint32_t)row, 0u, kMrxTagWhite}, a.key0, a.key1);
      const float2 g0 = mrx_dev::box_muller(rnd.x, rnd.y), g1 = mrx_dev::box_muller(rnd.z, rnd.w);
      float v[kSamplesPerThread] = {a.sqrt_fs * g0.x + p[0], a.sqrt_fs * g0.y + p[1], a.sqrt_fs * g1.x + p[2], a.sqrt_fs * g1.y + p[3]};
#pragma unroll
      for (int m = 0; m < kModes; ++m) {
        const float cm = coef[dl][m];
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q) v[q] = fmaf(cm, mw[m][q], v[q]);
      }
      const float sc = a.scale ? a.scale[row] : 1.0f;
      float sv[kSamplesPerThread];
      if (a.loading) {  // total NEP of a sample: NEP + NEP_per_loading x loading (sim/noise.py:35-37)
        const float* L = a.loading + (size_t)row * a.ld_loading + sb;
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q) sv[q] = v[q] * (sc + a.per_loading * (sb + q < a.T ? L[q] : 0.0f));
      } else {
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q) sv[q] = v[q] * sc;
      }
      float o[kSamplesPerThread];
      if constexpr (kKrj) {
        const CalDet c = cdet[dl];
        krj_row<false, decltype(curved)::value>(c, cal_cells + c.band * (a.n_el - 1), a.n_el, el_first, el_last, el_inv, ks, sv, o,
                                                a.bore_el, sb, a.T);
      } else {
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q) o[q] = sv[q];
      }
      float* dst = a.out + (size_t)row * a.ld + sb;
      if (full) {
        nvfloat4 x = {o[0], o[1], o[2], o[3]};
        nvfloat4* d4 = reinterpret_cast<nvfloat4*>(dst);
        if (a.accumulate) x += *d4;
        __builtin_nontemporal_store(x, d4);
      } else {
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q)
          if (sb + q < a.T) dst[q] = a.accumulate ? dst[q] + o[q] : o[q];
      }
    }
  };
  if (kKrj && ks.curved) rows_loop(std::true_type{}); else rows_loop(std::false_type{});  // (uniform)
}

// Linear interpolation of the coarse pwv (float64, time-major) to the full
// rate: sim/atmosphere.py:30-37.  Same tiling as the cubic kernel without the
// LDS stage; only the optional map/cmb consumers need it.
__global__ __launch_bounds__(kBlock) void linear_upsample_kernel(
    const double* __restrict__ pwv, int D, int n, double ta0, double inv_dta,
    double dta, const double* __restrict__ t, int T, float* __restrict__ out,
    size_t ld) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= T) return;
  const double x = (t[s] - ta0) * inv_dta;
  int jj = (int)floor(fmin(fmax(x, -1.0), 2.0e9));
  jj = min(max(jj, 0), n - 2);
  const double u = (t[s] - (ta0 + (double)jj * dta)) * inv_dta;
  const int d0 = blockIdx.y * kTileDet;
  const int nd = min(kTileDet, D - d0);
  for (int dl = 0; dl < nd; ++dl) {
    const int d = d0 + dl;
    const double y0 = pwv[(size_t)jj * D + d];
    const double y1 = pwv[(size_t)(jj + 1) * D + d];
    out[(size_t)d * ld + s] = (float)(y0 + u * (y1 - y0));
  }
}

// Full-rate detector pointing (coords/coordinates.py:378-386 at the sample rate,
// sim/observation.py:55-58): az/el [D][T] float32 from the boresight and the
// detector offsets, the float32 chain of coords/transforms.py:10-29.  Same tile
// as the TOD writer: 16 rows x 1024 samples, two 16-byte stores per row.
__global__ __launch_bounds__(kBlock) void pointing_broadcast_kernel(
    const float* __restrict__ az, const float* __restrict__ el, int T,
    const float* __restrict__ dxs, const float* __restrict__ dys, int D,
    float* __restrict__ az_out, float* __restrict__ el_out, size_t ld,
    int vec_ok) {
  __shared__ float4 pdet[kTileDet];  // sin(r)cos(p), cos(r), sin(r)sin(p)
  const int d0 = blockIdx.y * kTileDet;
  const int sb = blockIdx.x * kTileSamples + threadIdx.x * kSamplesPerThread;
  const int nd = min(kTileDet, D - d0);
  if ((int)threadIdx.x < nd) {
    const float dx = dxs[d0 + threadIdx.x], dy = dys[d0 + threadIdx.x];
    const float r = sqrtf(dx * dx + dy * dy);
    const float p = atan2f(-dx, -dy);
    const float sr = sinf(r);
    pdet[threadIdx.x] = make_float4(__fmul_rn(sr, cosf(p)), cosf(r), __fmul_rn(sr, sinf(p)), 0.f);
  }
  float ca[kSamplesPerThread], sa[kSamplesPerThread], zz[kSamplesPerThread];
#pragma unroll
  for (int q = 0; q < kSamplesPerThread; ++q) {
    const int s = min(sb + q, T - 1);
    const float a = el[s] - 1.57079637050628662109375f;
    ca[q] = cosf(a);
    sa[q] = sinf(a);
    zz[q] = az[s];
  }
  __syncthreads();
  const bool full = (sb + kSamplesPerThread <= T) && vec_ok;
  for (int dl = 0; dl < nd; ++dl) {
    const float4 c = pdet[dl];
    float oa[kSamplesPerThread], oe[kSamplesPerThread];
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) {
      const float re = __fsub_rn(__fmul_rn(c.x, ca[q]), __fmul_rn(c.y, sa[q]));
      const float im = __fadd_rn(__fmul_rn(c.x, sa[q]), __fmul_rn(c.y, ca[q]));
      oa[q] = __fadd_rn(atan2f(c.z, re), zz[q]);
      oe[q] = asinf(im);
    }
    float* da = az_out + (size_t)(d0 + dl) * ld + sb;
    float* de = el_out + (size_t)(d0 + dl) * ld + sb;
    if (full) {
      const vfloat4 va = {oa[0], oa[1], oa[2], oa[3]};
      const vfloat4 ve = {oe[0], oe[1], oe[2], oe[3]};
      __builtin_nontemporal_store(va, reinterpret_cast<vfloat4*>(da));
      __builtin_nontemporal_store(ve, reinterpret_cast<vfloat4*>(de));
    } else {
#pragma unroll
      for (int q = 0; q < kSamplesPerThread; ++q)
        if (sb + q < T) {
          da[q] = oa[q];
          de[q] = oe[q];
        }
    }
  }
}

}  // namespace

int mrx_noise_two_rate_write(mrx_ctx* ctx, hipStream_t stream, const mrx_two_rate_args& a) {
  const dim3 grid(mrx_ceil_div(a.T, kTileSamples), mrx_ceil_div(a.rows, kTileDet));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "too many rows for one launch");
  if (a.bore_el) {
    MRX_REQUIRE(ctx, a.n_el >= 2 && a.n_bands >= 1 && (size_t)(a.n_el - 1) * a.n_bands <= 6144,
                "calibration tables need 2 <= n_el and (n_el-1)*n_bands <= 6144");
    const size_t lds = sizeof(float4) * (size_t)(a.n_el - 1) * a.n_bands;
#define MRX_TWO_RATE(K, M)                                                                     \
  do {                                                                                         \
    MRX_LDS_CAP(ctx, (noise_two_rate_kernel<K, M>), lds);                                      \
    hipLaunchKernelGGL((noise_two_rate_kernel<K, M>), grid, dim3(kBlock), lds, stream, a);     \
  } while (0)
    if (a.n_modes == 0) MRX_TWO_RATE(true, 0); else if (a.n_modes <= 5) MRX_TWO_RATE(true, 5); else MRX_TWO_RATE(true, 8);
  } else {
    const size_t lds = 0;
    if (a.n_modes == 0) MRX_TWO_RATE(false, 0); else if (a.n_modes <= 5) MRX_TWO_RATE(false, 5); else MRX_TWO_RATE(false, 8);
#undef MRX_TWO_RATE
  }
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

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

// The control block of the next launch that takes tiles from a queue: a ring of kSynthCtlSlots blocks per context,
// zeroed once (every launch leaves its block zero: synth_leave), so that launches of one context that overlap on
// different streams do not share a queue.
static int mrx_synth_ctl(mrx_ctx* ctx, int** out) {
  if (!ctx->d_synth_ctl) {
    MRX_HIP(ctx, hipMalloc(&ctx->d_synth_ctl, sizeof(int) * kSynthCtlInts * mrx_ctx::kSynthCtlSlots));
    MRX_HIP(ctx, hipMemsetAsync(ctx->d_synth_ctl, 0, sizeof(int) * kSynthCtlInts * mrx_ctx::kSynthCtlSlots, ctx->stream));
    MRX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (once per context: later launches may come on other streams)
  }
  *out = ctx->d_synth_ctl + (size_t)kSynthCtlInts * ctx->synth_ctl_next;
  ctx->synth_ctl_next = (ctx->synth_ctl_next + 1) % mrx_ctx::kSynthCtlSlots;
  return MRX_OK;
}

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
  const bool small = knots_per_tile + 6.0 <= 64.0;
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
  const size_t lds = small ? FusedLds<64, 2>::kBytes : FusedLds<256, 1>::kBytes;
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
    if (small) MRX_LAUNCH_UPF(true, 64, 2); else MRX_LAUNCH_UPF(true, 256, 1);
  } else {
    if (small) MRX_LAUNCH_UPF(false, 64, 2); else MRX_LAUNCH_UPF(false, 256, 1);
  }
#undef MRX_LAUNCH_UPF_Q
#undef MRX_LAUNCH_UPF
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

static int atm_synthesize(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az, const float* d_el, int Ta,
                          const float* d_dx, const float* d_dy, const int32_t* d_band, const float* d_mueller00, int D,
                          double pwv0, float* d_coarse, int block_rows, int head_rows, uint32_t* d_flags, double ta0,
                          double dta, const double* d_t, int T, const float* d_scale, const int32_t* d_rows, float* d_out,
                          size_t ld_out, const SynthCal* krj) {
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && Ta >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, plan != nullptr, "plan is null");
  MRX_REQUIRE(ctx, d_az && d_el && d_dx && d_dy && d_band && d_mueller00, "null input pointer");
  MRX_REQUIRE(ctx, d_coarse && d_flags && d_t && d_out, "null pointer");
  MRX_REQUIRE(ctx, plan->n_layers == 0 || Ta == plan->n_t, "Ta differs from the plan's n_t (length of the wind offsets)");
  MRX_REQUIRE(ctx, dta > 0.0, "coarse step must be positive");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  if (Ta < 4)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "cubic interpolation needs at least 4 coarse samples (got %d), as scipy "
                    "interp1d(kind='cubic') does", Ta);
  // the one-launch form exists for what the default step runs: every layer on a verified-uniform axis, the
  // reference's default cell rule and pointing, linear band tables (otherwise: mrx_atm_sample + mrx_spline_upsample_fused)
  const bool literal = ctx->options[MRX_OPT_AXIS_LITERAL] != 0 || ctx->options[MRX_OPT_POINTING_CHAIN] != 0;
  if (!plan->all_pixel || literal || plan->any_cubic || plan->n_layers > mrx_px::kMaxAnchors)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "mrx_atm_synthesize: this plan or option set takes the two-call form");
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  if (block_rows <= 0) block_rows = D;
  block_rows = std::min(mrx_ceil_div(block_rows, kBlock) * kBlock, mrx_ceil_div(D, 32) * 32);  // whole groups of 256 lanes; rows of whole lines
  const int n_blocks = mrx_ceil_div(D, block_rows);
  MRX_REQUIRE(ctx, n_blocks <= kSynthMaxBlocks, "too many detector blocks");
  MRX_REQUIRE(ctx, (long long)Ta * block_rows * 4 < (1LL << 31), "a block's coarse array must stay below 2 GiB");
  MRX_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(d_coarse) & 127u) == 0, "d_coarse must be 128-byte aligned");
  // ---- writer role: mrx_spline_upsample_fused's choices ----
  const int vec_ok = (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0);
  const double knots_per_tile = (double)kTileSamples * (double)Ta / (double)T;
  const bool small = knots_per_tile + 6.0 <= 64.0;
  const int rows_per_batch = small ? 2 * kTileDet : kTileDet;
  int batches = ctx->options[MRX_OPT_UPSAMPLE_GROUPS];
  if (batches <= 0) batches = 1;
  while (batches > 1 && (long long)mrx_ceil_div(T, kTileSamples) * mrx_ceil_div(D, rows_per_batch * batches) < 4LL * 256 * 4)
    batches /= 2;
  const long long rows_per_tile = (long long)rows_per_batch * batches;
  const long long nsx = mrx_ceil_div(T, kTileSamples);
  const long long n_tiles = nsx * ((long long)(n_blocks - 1) * ((block_rows + rows_per_tile - 1) / rows_per_tile) +
                                   ((D - (long long)(n_blocks - 1) * block_rows) + rows_per_tile - 1) / rows_per_tile);
  MRX_REQUIRE(ctx, n_tiles <= 0x7fffffffLL - 65536, "too many tiles for one launch");
  const size_t lds_w = small ? FusedLds<64, 2>::kBytes : FusedLds<256, 1>::kBytes;
  // ---- sampler role: mrx_atm_sample's choices for a resident grid ----
  int chunk = ctx->options[MRX_OPT_SAMPLE_CHUNK];
  if (chunk <= 0) {
    chunk = mrx_px::kMaxChunk;
    const long long want = 24LL * n_cu;
    while (chunk > 1 && (long long)mrx_ceil_div(D, kBlock) * mrx_ceil_div(Ta, chunk) < want) chunk /= 2;
  }
  chunk = chunk < 1 ? 1 : chunk > mrx_px::kMaxChunk ? mrx_px::kMaxChunk : chunk;
  while (chunk > 1 && chunk * plan->n_layers > mrx_px::kMaxAnchors) chunk /= 2;
  // the launch has ONE dynamic LDS size and a CU's LDS is what bounds its writers: the sampler's anchors stay under
  // the writer's images (16 layers at 64 steps a work item took 37 KB -- and a writer's place on every CU)
  const size_t lds_turn = sizeof(float) * 4 * kBlock;  // four steps of every lane (px_sample_items<..., kWriteThrough>)
  while (chunk > 8 && 2 * sizeof(float4) * (size_t)chunk * plan->n_layers + lds_turn > lds_w) chunk /= 2;
  const int nby = mrx_ceil_div(Ta, chunk);
  const long long n_items = (long long)nby * ((long long)(n_blocks - 1) * mrx_ceil_div(block_rows, kBlock) +
                                              mrx_ceil_div(D - (n_blocks - 1) * block_rows, kBlock));
  MRX_REQUIRE(ctx, n_items <= 0x7fffffffLL, "too many work items for one launch");
  int per_cu = ctx->options[MRX_OPT_SAMPLE_WGS_PER_CU];
  if (per_cu <= 0 || per_cu >= 8) per_cu = 2;
  long long wgs_s = std::min(n_items, (long long)per_cu * n_cu);
  if (wgs_s >= 8) wgs_s &= ~7LL;
  // the head start: the first blocks by a grid that fills the chip (MRX_WRITER_WAVES workgroups per CU)
  const int head_blocks = std::max(0, std::min(mrx_ceil_div(std::max(head_rows, 0), block_rows), n_blocks));
  const long long wgs_full = std::max(wgs_s, std::min(n_items, (long long)MRX_WRITER_WAVES * n_cu) & ~7LL);
  SynthPhases phases = {};
  // (a staircase -- the head's blocks in shares to 5, 4, 3 workgroups per CU, writers entering as each step leaves --
  //  measured no better than one step: 1.98-2.05 against 1.93-1.98 ms)
  if (head_blocks > 0) {
    phases.n = 2;
    phases.end[0] = head_blocks; phases.wgs[0] = (int)wgs_full;
    phases.end[1] = n_blocks;    phases.wgs[1] = (int)wgs_s;
  } else {
    phases.n = 1;
    phases.end[0] = n_blocks; phases.wgs[0] = (int)wgs_s;
  }
  const long long wgs_head = phases.wgs[0];
  const size_t lds_anchor = 2 * sizeof(float4) * (size_t)chunk * plan->n_layers;
  const size_t lds_tables = sizeof(float) * (size_t)((plan->table_floats + 3) / 4 * 4);
  // band tables in LDS only where they fit under the writer's images too
  const size_t lds_cal = krj ? sizeof(float4) * (size_t)(krj->n_el - 1) * krj->n_bands : 0;  // the K_RJ cell table
  const bool lds_tab = plan->table_floats <= mrx_px::kMaxLdsTableFloats && lds_anchor + lds_tables + lds_turn + lds_cal <= lds_w;
  if (lds_anchor + (lds_tab ? lds_tables : 0) + lds_turn + lds_cal > lds_w)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "mrx_atm_synthesize: the calibration table does not fit beside the sampler's LDS: use the two calls");
  const size_t lds = lds_w;
  // writers: as many as fit a CU once the samplers have left (the surplus is dispatched as those exit)
  const long long per_cu_w = std::max<long long>(1, std::min<long long>(MRX_WRITER_WAVES, (long long)(ctx->lds_per_cu > 0 ? ctx->lds_per_cu : 160 * 1024) / (long long)(lds + 1552)));
  const long long wgs_w = std::min(n_tiles, per_cu_w * n_cu);
  int* ctl = nullptr;
  {
    const int rc = mrx_synth_ctl(ctx, &ctl);
    if (rc != MRX_OK) return rc;
  }
  const dim3 grid((unsigned)(wgs_head + wgs_w));
  const int poll_limit = 1 << 22;  // x >= 0.5 us a try: seconds
  const SynthCal cal = krj ? *krj : SynthCal{};
#define MRX_LAUNCH_SYNTH(L, S, K, G, J)                                                                           \
  do {                                                                                                            \
    MRX_LDS_CAP(ctx, (atm_tod_kernel<L, S, K, G, J>), lds);                                                       \
    hipLaunchKernelGGL((atm_tod_kernel<L, S, K, G, J>), grid, dim3(kBlock), lds, ctx->stream, plan->d_fast,       \
                       plan->d_px, plan->n_layers, plan->d_offpx, plan->d_tables, plan->n_tables,                 \
                       plan->d_table_data, plan->table_floats, d_az, d_el, Ta, d_dx, d_dy, d_band, d_mueller00,   \
                       D, pwv0, d_coarse, d_flags, chunk, nby, block_rows, n_blocks, (int)wgs_head, phases, ta0, 1.0 / dta,           \
                       d_t, T, d_scale, d_rows, d_out, ld_out, vec_ok, batches, ctl, poll_limit, cal);           \
  } while (0)
#define MRX_LAUNCH_SYNTH_J(L, S, K, G) do { if (krj) MRX_LAUNCH_SYNTH(L, S, K, G, true); else MRX_LAUNCH_SYNTH(L, S, K, G, false); } while (0)
#define MRX_LAUNCH_SYNTH_S(L, S) do { if (small) MRX_LAUNCH_SYNTH_J(L, S, 64, 2); else MRX_LAUNCH_SYNTH_J(L, S, 256, 1); } while (0)
  if (lds_tab) {
    if (d_scale) MRX_LAUNCH_SYNTH_S(true, true); else MRX_LAUNCH_SYNTH_S(true, false);
  } else {
    if (d_scale) MRX_LAUNCH_SYNTH_S(false, true); else MRX_LAUNCH_SYNTH_S(false, false);
  }
#undef MRX_LAUNCH_SYNTH_S
#undef MRX_LAUNCH_SYNTH_J
#undef MRX_LAUNCH_SYNTH
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_atm_synthesize(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az, const float* d_el, int Ta,
                       const float* d_dx, const float* d_dy, const int32_t* d_band, const float* d_mueller00, int D,
                       double pwv0, float* d_coarse, int block_rows, int head_rows, uint32_t* d_flags, double ta0,
                       double dta, const double* d_t, int T, const float* d_scale, const int32_t* d_rows, float* d_out,
                       size_t ld_out) {
  MRX_ENTER(ctx);
  return atm_synthesize(ctx, plan, d_az, d_el, Ta, d_dx, d_dy, d_band, d_mueller00, D, pwv0, d_coarse, block_rows, head_rows,
                        d_flags, ta0, dta, d_t, T, d_scale, d_rows, d_out, ld_out, nullptr);
}

int mrx_atm_synthesize_krj(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az, const float* d_el, int Ta,
                           const float* d_dx, const float* d_dy, const int32_t* d_band, const float* d_mueller00, int D,
                           double pwv0, float* d_coarse, int block_rows, int head_rows, uint32_t* d_flags, double ta0,
                           double dta, const double* d_t, int T, const float* d_scale, const int32_t* d_rows, float* d_out,
                           size_t ld_out, const float* d_cal_dx, const float* d_cal_dy, const float* d_cal_axis_el,
                           const float* d_cal_values, int n_el, int n_bands, float* d_tail_pw, int tail_knots, size_t ld_tail) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, d_cal_dx && d_cal_dy && d_cal_axis_el && d_cal_values, "null calibration pointer");
  MRX_REQUIRE(ctx, n_el >= 2 && n_bands >= 1, "calibration tables need 2 <= n_el, 1 <= n_bands");
  MRX_REQUIRE(ctx, tail_knots >= 0 && tail_knots <= Ta && (!d_tail_pw || ld_tail >= (size_t)D), "bad tail window");
  SynthCal cal{d_cal_dx, d_cal_dy, d_cal_axis_el, d_cal_values, n_el, n_bands, tail_knots > 0 ? d_tail_pw : nullptr, Ta - tail_knots, ld_tail};
  return atm_synthesize(ctx, plan, d_az, d_el, Ta, d_dx, d_dy, d_band, d_mueller00, D, pwv0, d_coarse, block_rows, head_rows,
                        d_flags, ta0, dta, d_t, T, d_scale, d_rows, d_out, ld_out, &cal);
}

int mrx_spline_upsample_krj(mrx_ctx* ctx, const float* d_ym, int D, int Ta,
                            double ta0, double dta, const double* d_t, int T,
                            const float* d_scale, const int32_t* d_rows,
                            const float* d_bore_el, const float* d_dx,
                            const float* d_dy, const int32_t* d_band,
                            const float* d_cal_axis_el,
                            const float* d_cal_values, int n_el, int n_bands,
                            float* d_out, size_t ld_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_ym && d_t && d_out, "null pointer");
  MRX_REQUIRE(ctx, d_bore_el && d_dx && d_dy && d_band && d_cal_axis_el &&
                       d_cal_values,
              "null calibration pointer");
  // the cell table lives in LDS as one float4 per (band, cell): 96 KiB beside the kernels' static images
  MRX_REQUIRE(ctx, n_el >= 2 && n_bands >= 1 && (size_t)(n_el - 1) * n_bands <= 6144,
              "calibration tables need 2 <= n_el and (n_el-1)*n_bands <= 6144");
  MRX_REQUIRE(ctx, dta > 0.0, "coarse step must be positive");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  if (Ta < 4)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED,
                    "cubic interpolation needs at least 4 coarse samples");
  // detector tiles per workgroup: the per-sample prologue is shared by all of them
  int groups = ctx->options[MRX_OPT_UPSAMPLE_GROUPS];
  if (groups <= 0) groups = 4;  // measured on atlast_10k: 1 -> 2.69 ms, 2 -> 2.55, 4 -> 2.48
  while (groups > 1 && (long long)mrx_ceil_div(T, kTileSamples) *
                               mrx_ceil_div(D, kTileDet * groups) < 4LL * 256 * 4)
    groups /= 2;  // keep the chip full on small problems
  dim3 grid(mrx_ceil_div(T, kTileSamples), mrx_ceil_div(D, kTileDet * groups));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  const int vec_ok =
      (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0);
  const size_t lds = sizeof(float4) * (size_t)(n_el - 1) * n_bands + sizeof(KrjWriterLds);  // the cell table, the knot image
  MRX_LDS_CAP(ctx, spline_upsample_krj_kernel, lds);
  hipLaunchKernelGGL(spline_upsample_krj_kernel, grid, dim3(kBlock), lds,
                     ctx->stream, reinterpret_cast<const float2*>(d_ym), D, Ta,
                     ta0, 1.0 / dta, d_t, T, d_scale, d_rows, d_bore_el, d_dx,
                     d_dy, d_band, d_cal_axis_el, d_cal_values, n_el, n_bands,
                     d_out, ld_out, vec_ok, groups);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_coarse_to_krj(mrx_ctx* ctx, const float* d_loading, int D, int Ta, const float* d_bore_el_coarse,
                      const float* d_dx, const float* d_dy, const int32_t* d_band,
                      const float* d_cal_axis_el, const float* d_cal_values, int n_el, int n_bands,
                      float* d_out) {
  return mrx_coarse_to_krj_keep_tail(ctx, d_loading, D, Ta, d_bore_el_coarse, d_dx, d_dy, d_band, d_cal_axis_el, d_cal_values,
                                     n_el, n_bands, d_out, nullptr, 0, 0);
}

int mrx_coarse_to_krj_keep_tail(mrx_ctx* ctx, const float* d_loading, int D, int Ta, const float* d_bore_el_coarse,
                                const float* d_dx, const float* d_dy, const int32_t* d_band,
                                const float* d_cal_axis_el, const float* d_cal_values, int n_el, int n_bands,
                                float* d_out, float* d_tail_pw, int tail_knots, size_t ld_tail) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && Ta >= 0, "negative size");
  MRX_REQUIRE(ctx, tail_knots >= 0 && tail_knots <= Ta, "tail_knots must lie in [0, Ta]");
  if (!d_tail_pw || tail_knots == 0) { d_tail_pw = nullptr; tail_knots = 0; }
  MRX_REQUIRE(ctx, !d_tail_pw || ld_tail >= (size_t)D, "ld_tail is shorter than a row of D detectors");
  if (D == 0 || Ta == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_loading && d_out && d_bore_el_coarse && d_dx && d_dy && d_band && d_cal_axis_el && d_cal_values,
              "null pointer");
  // the cell table lives in LDS as one float4 per (band, cell): 96 KiB beside the kernels' static images
  MRX_REQUIRE(ctx, n_el >= 2 && n_bands >= 1 && (size_t)(n_el - 1) * n_bands <= 6144,
              "calibration tables need 2 <= n_el and (n_el-1)*n_bands <= 6144");
  const dim3 grid(mrx_ceil_div(D, kBlock), mrx_ceil_div(Ta, kCoarseKrjSteps));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "Ta too large for one launch");
  const size_t lds = sizeof(float4) * (size_t)(n_el - 1) * n_bands;
  MRX_LDS_CAP(ctx, coarse_krj_kernel, lds);
  hipLaunchKernelGGL(coarse_krj_kernel, grid, dim3(kBlock), lds, ctx->stream, d_loading, D, Ta, d_bore_el_coarse,
                     d_dx, d_dy, d_band, d_cal_axis_el, d_cal_values, n_el, n_bands, d_out, d_tail_pw, Ta - tail_knots, ld_tail);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

static int tod_convert(mrx_ctx* ctx, bool inverse, float* d_data, size_t ld, int D, int T,
                       const float* d_scale, const int32_t* d_rows,
                       const float* d_bore_el, const float* d_dx, const float* d_dy,
                       const int32_t* d_band, const float* d_cal_axis_el,
                       const float* d_cal_values, int n_el, int n_bands) {
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_data && d_bore_el && d_dx && d_dy && d_band && d_cal_axis_el && d_cal_values,
              "null pointer");
  // the cell table lives in LDS as one float4 per (band, cell): 96 KiB beside the kernels' static images
  MRX_REQUIRE(ctx, n_el >= 2 && n_bands >= 1 && (size_t)(n_el - 1) * n_bands <= 6144,
              "calibration tables need 2 <= n_el and (n_el-1)*n_bands <= 6144");
  MRX_REQUIRE(ctx, ld >= (size_t)T, "ld smaller than T");
  dim3 grid(mrx_ceil_div(T, kTileSamples), mrx_ceil_div(D, kTileDet));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  const int vec_ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_data) & 15u) == 0);
  const size_t lds = sizeof(float4) * (size_t)(n_el - 1) * n_bands;  // the cell table
  if (inverse) MRX_LDS_CAP(ctx, tod_krj_kernel<true>, lds); else MRX_LDS_CAP(ctx, tod_krj_kernel<false>, lds);
  if (inverse)
    hipLaunchKernelGGL(tod_krj_kernel<true>, grid, dim3(kBlock), lds, ctx->stream, d_data, ld, D, T,
                       d_scale, d_rows, d_bore_el, d_dx, d_dy, d_band, d_cal_axis_el,
                       d_cal_values, n_el, n_bands, vec_ok);
  else
    hipLaunchKernelGGL(tod_krj_kernel<false>, grid, dim3(kBlock), lds, ctx->stream, d_data, ld, D, T,
                       d_scale, d_rows, d_bore_el, d_dx, d_dy, d_band, d_cal_axis_el,
                       d_cal_values, n_el, n_bands, vec_ok);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_tod_to_krj(mrx_ctx* ctx, float* d_data, size_t ld, int D, int T,
                   const float* d_scale, const int32_t* d_rows,
                   const float* d_bore_el, const float* d_dx, const float* d_dy,
                   const int32_t* d_band, const float* d_cal_axis_el,
                   const float* d_cal_values, int n_el, int n_bands) {
  MRX_ENTER(ctx);
  return tod_convert(ctx, false, d_data, ld, D, T, d_scale, d_rows, d_bore_el, d_dx, d_dy, d_band,
                     d_cal_axis_el, d_cal_values, n_el, n_bands);
}

int mrx_tod_from_krj(mrx_ctx* ctx, float* d_data, size_t ld, int D, int T,
                     const float* d_scale, const int32_t* d_rows,
                     const float* d_bore_el, const float* d_dx, const float* d_dy,
                     const int32_t* d_band, const float* d_cal_axis_el,
                     const float* d_cal_values, int n_el, int n_bands) {
  MRX_ENTER(ctx);
  return tod_convert(ctx, true, d_data, ld, D, T, d_scale, d_rows, d_bore_el, d_dx, d_dy, d_band,
                     d_cal_axis_el, d_cal_values, n_el, n_bands);
}

int mrx_pointing_broadcast(mrx_ctx* ctx, const float* d_az, const float* d_el,
                           int T, const float* d_dx, const float* d_dy, int D,
                           float* d_az_out, float* d_el_out, size_t ld_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_az && d_el && d_dx && d_dy && d_az_out && d_el_out,
              "null pointer");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  dim3 grid(mrx_ceil_div(T, kTileSamples), mrx_ceil_div(D, kTileDet));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  const int vec_ok = (ld_out % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(d_az_out) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(d_el_out) & 15u) == 0);
  hipLaunchKernelGGL(pointing_broadcast_kernel, grid, dim3(kBlock), 0,
                     ctx->stream, d_az, d_el, T, d_dx, d_dy, D, d_az_out,
                     d_el_out, ld_out, vec_ok);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_linear_upsample(mrx_ctx* ctx, const double* d_pwv, int D, int Ta,
                        double ta0, double dta, const double* d_t, int T,
                        float* d_out, size_t ld_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_pwv && d_t && d_out, "null pointer");
  MRX_REQUIRE(ctx, dta > 0.0, "coarse step must be positive");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  MRX_REQUIRE(ctx, Ta >= 2, "linear interpolation needs 2 coarse samples");
  dim3 grid(mrx_ceil_div(T, kBlock), mrx_ceil_div(D, kTileDet));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  hipLaunchKernelGGL(linear_upsample_kernel, grid, dim3(kBlock), 0, ctx->stream,
                     d_pwv, D, Ta, ta0, 1.0 / dta, dta, d_t, T, d_out, ld_out);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

}  // extern "C"
