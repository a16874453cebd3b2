// The fused spline writer's tile (solve + evaluation of one tile of the TOD from the raw coarse samples): the body of
// spline_upsample_fused_kernel (mrx_spline.hip) and of the writer role of the one-launch synthesis (mrx_synth.hip),
// and the tile queue's control block the two share.
#pragma once

#include "mrx_tile.h"

namespace {

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

// The small image (upsampling ratios from ~20 up: a tile of 1024 samples then spans <= 52 knots + the widening at the
// ends) evaluates its samples in Horner form from per-interval coefficients (see fused_writer_tile, 3.): 58 knots is
// what lets the coefficient image of 32 rows x 57 intervals (29 184 bytes) lie over the solve's two images, so that
// the workgroup keeps its 30 KB and the CU its five workgroups.
constexpr int kSmallKnots = 58;

template <int kMaxKnots, int kG>
struct FusedLds {
  static constexpr int kRows = kTileDet * kG;
  static constexpr bool kHorner = kMaxKnots <= 64;
  static constexpr size_t kSolveBytes = sizeof(float2) * kRows * (kMaxKnots + 1) + sizeof(float) * kRows * ((kMaxKnots + 2 * kFHalo + 6) | 1);
  static constexpr size_t kCoefBytes = kHorner ? sizeof(float4) * kRows * (kMaxKnots - 1) : 0;
  static constexpr size_t kImageBytes = (((kSolveBytes > kCoefBytes ? kSolveBytes : kCoefBytes) + 15) / 16) * 16;
  // the images, then per row its place in the TOD (row x ld, 8 bytes)
  static constexpr size_t kBytes = kImageBytes + sizeof(unsigned long long) * kRows;
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
  // Horner form (the small image): after the solve, the knots (y, m) of a row become per-INTERVAL coefficients
  //   S(u) = c0 + u (c1 + u (c2 + u c3)),  c0 = y_j, c1 = (y_{j+1} - y_j) - 2 m_j - m_{j+1}, c2 = 3 m_j, c3 = m_{j+1} - m_j
  // (the same cubic as y_j + [u (y_{j+1} - y_j) + ((1-u)^3 - (1-u)) m_j + (u^3 - u) m_{j+1}], expanded in u; the
  // detector's scale goes into the coefficients), one float4 per (row, interval) in an image that lies OVER the solve's
  // two: a sample then costs one 16-byte LDS read and three fused multiply-adds instead of two 8-byte reads, five
  // operations and a multiply by the scale -- the row loop was 85 % of the writer's vector instructions (40 per row and
  // thread), and the launch that writes beside the sampler is bound by vector issue (SIMDs 92 % busy).
  constexpr bool kHorner = FusedLds<kMaxKnots, kG>::kHorner;
  constexpr int kCPitch = kMaxKnots - 1;  // intervals of a row (odd for the small image: 57)
  float4* coef = reinterpret_cast<float4*>(fused_lds);                 // [kRows][kCPitch], over tile and yraw
  // where a row starts in the TOD, in floats: staged per pass, read back in the row loop (one 64-bit add per lane and
  // row; computed there from the row number it was a 64-bit multiply per lane and row)
  unsigned long long* row_off = reinterpret_cast<unsigned long long*>(fused_lds + FusedLds<kMaxKnots, kG>::kImageBytes);  // [kRows]
  const int s_tile = sx * kTileSamples;
  const int sb = s_tile + threadIdx.x * kSamplesPerThread;
  float* const out_sb = out + sb;  // the thread's four samples in row 0

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
      // the thread's sample times: loaded now (in flight during the staging below), turned into interval and weights
      // after the solve -- 8 registers live through the prologue instead of 24 -- and loaded again in every pass (a
      // tile is one pass but for several row batches or small upsampling ratios; the empty asm keeps the loads from
      // being hoisted): kept across the row loop they were 8 registers of a kernel that sits at its bound
      double tq[kSamplesPerThread];
      {
        int sbq = sb;
        asm volatile("" : "+v"(sbq));
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q) tq[q] = t[min(sbq + q, T - 1)];
      }
      if (g > 0 || ja > j_first) __syncthreads();  // the previous pass is done with the images
      if ((int)threadIdx.x < nd) row_off[threadIdx.x] = (unsigned long long)(rows ? rows[d0 + threadIdx.x] : d0 + (int)threadIdx.x) * ld;
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
      if constexpr (kHorner) {
        // (y, m) per knot -> (c0 .. c3) per interval, through registers (the coefficient image lies over the knots): a
        // thread takes kSpan CONSECUTIVE intervals of one row -- kSpan + 1 knots, 2 (kSpan + 1) registers across the
        // barrier (one interval in kBlock / kRows, as the combine step deals them, would hold 4 kSpan)
        constexpr int kSpan = (kCPitch + kBlock / kRows - 1) / (kBlock / kRows);  // 8 for the small image
        {
          const int dl = threadIdx.x & (kRows - 1);
          const int k0 = (threadIdx.x / kRows) * kSpan;  // wave-uniform but for the two halves of a wave
          const float gsc = kHasScale ? scale[min(d0 + dl, D - 1)] : 1.0f;
          float2 kn[kSpan + 1];
#pragma unroll
          for (int i = 0; i <= kSpan; ++i) kn[i] = tile[dl * kPitch + min(k0 + i, K - 1)];
          __syncthreads();  // every thread has read its knots
#pragma unroll
          for (int i = 0; i < kSpan; ++i)
            if (k0 + i < K - 1) {
              const float2 s0 = kn[i], s1 = kn[i + 1];
              float4 c = make_float4(s0.x, ((s1.x - s0.x) - 2.0f * s0.y) - s1.y, 3.0f * s0.y, s1.y - s0.y);
              if (kHasScale) c = make_float4(gsc * c.x, gsc * c.y, gsc * c.z, gsc * c.w);
              coef[dl * kCPitch + k0 + i] = c;
            }
          __syncthreads();
        }
        // interval and position within it, float64 once per sample (u < 0 or > 1: extrapolation, the same cubic)
        int r[kSamplesPerThread], jq[kSamplesPerThread];
        float u[kSamplesPerThread];
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q) {
          const double x = (tq[q] - ta0) * inv_dta;
          jq[q] = interval_of(x, n);
          r[q] = min(max(jq[q] - jmin, 0), K - 2);  // in range even if t is unsorted
          u[q] = (float)(x - (double)(jmin + r[q]));
        }
        if (full) {
#pragma unroll 4
          for (int dl = 0; dl < nd; ++dl) {
            const float4* row = coef + dl * kCPitch;
            float o[kSamplesPerThread];
#pragma unroll
            for (int q = 0; q < kSamplesPerThread; ++q) {
              const float4 cf = row[r[q]];
              o[q] = fmaf(u[q], fmaf(u[q], fmaf(u[q], cf.w, cf.z), cf.y), cf.x);
            }
            const vfloat4 v = {o[0], o[1], o[2], o[3]};
            // (nt: beside the sampler it is the policy that costs least -- 2.17 ms against 2.43 plain, 2.47 sc1, 2.28 sc1 nt)
            __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(out_sb + row_off[dl]));
          }
        } else {
          // a sample belongs to the segment that holds its interval (every sample of a
          // single-segment tile does, whatever its interval)
          bool mine[kSamplesPerThread];
#pragma unroll
          for (int q = 0; q < kSamplesPerThread; ++q)
            mine[q] = sb + q < T && (single || (jq[q] >= ja && jq[q] < jb));
          for (int dl = 0; dl < nd; ++dl) {
            const float4* row = coef + dl * kCPitch;
            float* dst = out_sb + row_off[dl];
#pragma unroll
            for (int q = 0; q < kSamplesPerThread; ++q)
              if (mine[q]) {
                const float4 cf = row[r[q]];
                dst[q] = fmaf(u[q], fmaf(u[q], fmaf(u[q], cf.w, cf.z), cf.y), cf.x);
              }
          }
        }
      } else {
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
          __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(out_sb + row_off[dl]));
        }
      } else {
        bool mine[kSamplesPerThread];
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q)
          mine[q] = sb + q < T && (single || (w.j[q] >= ja && w.j[q] < jb));
        for (int dl = 0; dl < nd; ++dl) {
          const float2* row = tile + dl * kPitch;
          const float gsc = kHasScale ? scale[d0 + dl] : 1.0f;
          float* dst = out_sb + row_off[dl];
#pragma unroll
          for (int q = 0; q < kSamplesPerThread; ++q)
            if (mine[q]) dst[q] = gsc * spline_eval(w, q, row[r[q]], row[r[q] + 1]);
        }
      }
      }
    }
  }
}


// The raw knots [lo, hi] that the tile of time tile `sx` reads over all its passes (what fused_writer_tile stages:
// every pass widens its knot range by kFHalo + 3 either side and, at an end of the axis, to the knots the not-a-knot
// terms are derived from).  For a writer that must know which coarse samples it depends on (mrx_synth.hip).
__device__ __forceinline__ void fused_tile_knots(const double* __restrict__ t, int T, int n, double ta0, double inv_dta,
                                                 int sx, int& lo, int& hi) {
  const int s_tile = sx * kTileSamples;
  const int s_last = min(s_tile + kTileSamples, T) - 1;
  const int j_first = interval_of((t[s_tile] - ta0) * inv_dta, n);
  const int j_end = interval_of((t[s_last] - ta0) * inv_dta, n) + 1;
  lo = max(min(j_first, n - 3) - 3 - kFHalo, 0);
  hi = min(max(j_end, 2) + 3 + kFHalo, n - 1);
}

// The control block of a launch that takes its work from queues: [0] the tile queue, [16] workgroups that have left,
// [32] the sampler's work-item queue (the one-launch synthesis), [64 + s] finished work items of hand-over unit s
// (a detector block's time chunk: mrx_synth.hip) -- each kind on lines of its own: the tile queue is hammered by every
// writer, the counters by the samplers.  All zero between launches: the last workgroup to leave clears what the
// launch used, so a launch needs no memset.
constexpr int kCtlTiles = 0, kCtlLeft = 16, kCtlItems = 32, kCtlDone = 64;
constexpr int kSynthMaxSlots = 32768;
constexpr int kSynthCtlInts = kCtlDone + kSynthMaxSlots;

// Leaves the control block as it was found: the workgroup whose exit is the grid's last (atomicInc wraps the exit
// count to 0 by itself) zeroes the queues and the first n_slots counters -- nobody reads them any more, and the next
// launch on the stream starts after this one has ended.
__device__ __forceinline__ void synth_leave(int* ctl, int n_slots) {
  __syncthreads();
  __shared__ int s_last;
  if (threadIdx.x == 0) s_last = atomicInc(reinterpret_cast<unsigned*>(ctl + kCtlLeft), gridDim.x - 1) == gridDim.x - 1;
  __syncthreads();
  if (s_last) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(ctl + kCtlTiles, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctl + kCtlItems, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int b = threadIdx.x; b < n_slots; b += kBlock)
      __hip_atomic_store(ctl + kCtlDone + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace

// The control block of the next launch that takes tiles from a queue (mrx_spline.hip)
int mrx_synth_ctl(mrx_ctx* ctx, int** out);
