// The TOD tile shared by every kernel that writes or reads the [D][T] array a tile at a time (the spline writers, the
// K_RJ conversions, the pointing broadcast, the two-rate noise writer): kTileDet detector rows x 1024 consecutive samples
// per 256-thread workgroup, a thread owns 4 consecutive samples (one 16-byte store per row, 1 KiB contiguous per wave);
// and the cubic's per-sample interval / basis weights (float64 once per sample, reused by every row).
#pragma once

#include "mrx_internal.h"

namespace {

using ::mrx_dev_common::kBlock;

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

}  // namespace
