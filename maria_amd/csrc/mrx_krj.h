// TOD.to("K_RJ") on the device: the per-detector constants, the calibration table as the kernels hold it in LDS and
// the per-tile elevation model shared by the K_RJ writers and conversions (mrx_krj.hip), the sampler role of the
// one-launch synthesis (mrx_synth.hip) and the two-rate noise writer (mrx_noise.hip).
#pragma once

#include "mrx_tile.h"

// The K_RJ division must round alike wherever it is inlined -- tod_krj_kernel, the spline writers, the noise writer, the
// one-launch sampler, the map sampler -- because callers compare their fields bit for bit (a field written in K_RJ against
// the same field converted afterwards), and those kernels live in translation units built with different contraction
// rules (mrx_map.hip: -ffp-contract=off).  "Let the compiler fuse" is not a rule two translation units apply alike (the
// same source gave 15 fused multiply-adds in one and 12 in the other, and a per-row elevation model 1e-7 rad apart: one
// value in fifteen a float32 ulp off).  So every function here states `contract(off)` for its own body -- the statement
// form of the pragma, which leaves nothing behind for the including file -- and writes the fused operations it wants as
// fmaf: IEEE operations in a fixed order, the same bits wherever they are inlined.
#define MRX_KRJ_FP _Pragma("clang fp contract(off)")

namespace {

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
  // lin = 1: over the tile's whole boresight range the row stays inside ONE cell of the table's elevation axis (a cell
  // is a few degrees, a tile's 1 024 samples sweep a degree or so) and on the axis, and the model above holds -- so
  // den is ONE linear function of x = el_bore - ebm for every sample of the row in this tile: den = lin_a + lin_b x
  // (set_linear_den; krj_row then needs neither the cell nor its checks per thread: the division cost ~21 vector
  // instructions a sample wherever it rode on another kernel's store)
  // Rows that meet ONE node of the axis inside the tile (a cell is a few degrees, a tile's sweep a degree: 40 % of the
  // rows at the benchmark's 4-degree cells) are linear either side of it: den = lin_a + lin_b x + lin_db max(x - lin_xn, 0),
  // lin_xn the x at which the row's elevation passes the node (3e38 without one).
  float lin_a, lin_b, lin_db, lin_xn;
  int lin;
};

// detector elevation (transforms.py:20-28): im = sin(el) as the chain computes
// it, then el = asin(im) by one Newton step from el0 = el_bore + dy, whose
// sine and cosine follow from the angle-addition formulas (no inverse
// trigonometry per sample); |el - el0| <= r^2 tan(el)/2 ~ 3e-4 rad, so the
// second-order step is exact to float32 rounding.  eb: boresight elevation,
// ca / sa: cos / sin of (eb - pi/2).
// sin(el_det) as the float32 chain forms it (transforms.py:20-28): two products and a sum, each rounded.  ca / sa: cos / sin
// of (boresight elevation - pi/2).  One definition for every kernel that divides by den(el_det) -- here, coarse_krj_kernel
// and the one-launch sampler (mrx_synth.hip), whose fields are compared bit for bit.
__device__ __forceinline__ float det_sin_elevation(float a_re, float a_im, float ca, float sa) {
  MRX_KRJ_FP
  return a_re * sa + a_im * ca;
}

__device__ __forceinline__ float det_elevation(const CalDet& c, float eb, float ca, float sa) {
  MRX_KRJ_FP
  const float im = det_sin_elevation(c.a_re, c.a_im, ca, sa);
  const float s0 = fmaf(ca, c.cdy, -(sa * c.sdy));   // sin(el_bore + dy)
  const float c0 = -fmaf(sa, c.cdy, ca * c.sdy);     // cos(el_bore + dy)
  const float rc0 = __builtin_amdgcn_rcpf(c0);
  const float dl1 = (im - s0) * rc0;
  float el = fmaf(dl1, 1.0f + 0.5f * dl1 * s0 * rc0, eb + c.dy);
  // within ~15 deg of the zenith the expansion loses accuracy: take asin there
  const bool steep = !(c0 > 0.25f);
  if (__builtin_amdgcn_ballot_w64(steep) != 0)
    if (steep) el = asinf(im);
  return el;
}

// the linear model of CalDet around the boresight elevation ebm (see CalDet)
constexpr float kModelHalfRange = 2.0e-2f;  // wide enough that float32 rounding of the differences stays below 1e-7 rad over a tile

// (cos, sin) of (ebm - pi/2) and of that angle -+ kModelHalfRange: ONE cosine and sine (the accurate ones) and the
// angle-addition formulas with the constants cos / sin(0.02) -- the model's three points used to cost six calls of the
// accurate functions per tile, on sixteen lanes of one wave while the workgroup's other 240 threads waited at the barrier.
struct BoreTrig {
  float ca[3], sa[3];
};

__device__ __forceinline__ BoreTrig bore_trig(float ebm) {
  MRX_KRJ_FP
  constexpr float ch = 0.99980000666657776f, sh = 0.01999866669333308f;  // cos, sin of kModelHalfRange = 0.02
  const float a = ebm - 1.57079637050628662109375f;
  const float c = cosf(a), s = sinf(a);
  BoreTrig b;
  b.ca[1] = c;
  b.sa[1] = s;
  b.ca[0] = fmaf(c, ch, s * sh);   // cos(a - h)
  b.sa[0] = fmaf(s, ch, -(c * sh));
  b.ca[2] = fmaf(c, ch, -(s * sh));  // cos(a + h)
  b.sa[2] = fmaf(s, ch, c * sh);
  return b;
}

__device__ __forceinline__ void set_elevation_model(CalDet& c, float ebm, const BoreTrig& b) {
  MRX_KRJ_FP
  constexpr float h = kModelHalfRange;
  float e[3];
  bool steep = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float eb = ebm + (float)(k - 1) * h;
    steep |= !(-fmaf(b.sa[k], c.cdy, b.ca[k] * c.sdy) > 0.3f);  // cos(el_bore + dy): within ~17 deg of the zenith
    e[k] = det_elevation(c, eb, b.ca[k], b.sa[k]) - eb;
  }
  c.ebm = ebm;
  c.dm = e[1];
  c.slope = (e[2] - e[0]) * (0.5f / h);
  c.exact = steep ? 1 : 0;
}

// The row's den as one linear function of x = el_bore - ebm over the tile (CalDet::lin).  [lo, hi]: the tile's boresight
// elevation range (krj_prologue: red[8], red[9]); C: the band's cells.  den_lookup's value at the model's elevation
// e(x) = (ebm + dm) + (1 + slope) x inside cell i: y_i + (y_{i+1} - y_i) z_i (e - x_i), expanded in x -- it differs from
// den_lookup's own float32 evaluation by a rounding or two of den (1e-7), alike in every kernel that includes this header.
__device__ __forceinline__ void set_linear_den(CalDet& c, float lo, float hi, const float4* C, int n_el, float el_first,
                                               float el_inv) {
  MRX_KRJ_FP
  c.lin = 0;
  c.lin_a = c.lin_b = c.lin_db = 0.0f;
  c.lin_xn = 3.0e38f;
  if (c.exact) return;
  const float S = 1.0f + c.slope, E0 = c.ebm + c.dm;  // the row's elevation over the tile: E0 + S x  (S > 0)
  const float e0 = fmaf(c.slope, lo - c.ebm, lo + c.dm), e1 = fmaf(c.slope, hi - c.ebm, hi + c.dm);
  const int i0 = min(max((int)fminf(fmaxf((e0 - el_first) * el_inv, -1.0f), 2.0e9f), 0), n_el - 2);
  const float4 cell = C[i0];
  const float w0 = (e0 - cell.x) * cell.z, w1 = (e1 - cell.x) * cell.z;
  // the low end inside cell i0, with a margin of 1e-4 of its width (the threads' own elevations are rounded separately; den
  // is continuous at a node anyway) -- hence on the axis; a guess that missed the cell (non-uniform axis) or a NaN anywhere:
  // not linear, and the row takes the per-sample form
  if (!(S > 0.5f && e0 <= e1 && w0 >= 1.0e-4f)) return;
  const float g1 = (cell.w - cell.y) * cell.z;  // d den / d el in the cell
  c.lin_a = fmaf(g1, E0 - cell.x, cell.y);
  c.lin_b = g1 * S;
  if (w1 <= 1.0f - 1.0e-4f) {  // the whole tile inside the cell
    c.lin = 1;
    return;
  }
  if (i0 + 1 > n_el - 2) return;  // (past the last node: off the axis)
  const float4 up = C[i0 + 1];   // (x_{i+1}, den_{i+1}, 1 / (x_{i+2} - x_{i+1}), den_{i+2})
  if (!((e1 - up.x) * up.z <= 1.0f - 1.0e-4f)) return;  // the high end beyond the next cell too
  const float g2 = (up.w - up.y) * up.z;
  c.lin_db = (g2 - g1) * S;
  c.lin_xn = (up.x - E0) / S;
  c.lin = 1;
}

// sin(x) / x and cos(x) for a focal-plane offset (|x| below 0.3 rad = 17 deg: the series to x^8 are exact to float32 there)
__device__ __forceinline__ float sinc_small(float x2) {
  MRX_KRJ_FP
  return fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 1.0f / 362880.0f, -1.0f / 5040.0f), 1.0f / 120.0f), -1.0f / 6.0f), 1.0f);
}
__device__ __forceinline__ float cos_small(float x2) {
  MRX_KRJ_FP
  return fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 1.0f / 40320.0f, -1.0f / 720.0f), 1.0f / 24.0f), -0.5f), 1.0f);
}

// The detector's constants of transforms.py:14-23: r = |offset|, p = atan2(-dx, -dy), a_re = sin(r) cos(p), a_im = cos(r).
// Round 6: cos(p) = -dy / r, so a_re = -dy sin(r) / r -- the series above instead of a square root, an arctangent and five
// sines and cosines per row and tile (a thousand instructions of one wave, serial, behind the tile's barrier); the values
// are the exact ones to a float32 rounding, where the reference's own float32 chain carries three.  Offsets beyond 0.3 rad
// keep the functions.
__device__ __forceinline__ CalDet make_cal_det(float dx, float dy, int band, float scale) {
  MRX_KRJ_FP
  const float r2 = fmaf(dx, dx, dy * dy), y2 = dy * dy;
  CalDet c;
  if (r2 < 0.09f) {
    c.a_re = -dy * sinc_small(r2);
    c.a_im = cos_small(r2);
    c.sdy = dy * sinc_small(y2);
    c.cdy = cos_small(y2);
  } else {
    const float r = sqrtf(r2);
    const float p = atan2f(-dx, -dy);
    c.a_re = sinf(r) * cosf(p);
    c.a_im = cosf(r);
    c.sdy = sinf(dy);
    c.cdy = cosf(dy);
  }
  c.band = band;
  c.scale = scale;
  c.dy = dy;
  return c;
}

// The calibration table as the kernels hold it in LDS: per band, per cell i of the elevation
// axis one float4 (x_i, den_i, 1/(x_{i+1} - x_i), den_{i+1}), so that one 16-byte LDS read
// serves a lookup.  n_el - 1 cells per band.
__device__ __forceinline__ void stage_cal_cells(float4* cells, const float* __restrict__ axis,
                                                const float* __restrict__ values, int n_el, int n_bands) {
  MRX_KRJ_FP
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
  MRX_KRJ_FP
  const int nc = n_el - 1;
  int i = min(max((int)fminf(fmaxf((el - el_first) * el_inv, -1.0f), 2.0e9f), 0), nc - 1);
  while (i < nc - 1 && C[i + 1].x < el) ++i;  // searchsorted(side="left") - 1: x_i < el <= x_{i+1}
  while (i > 0 && C[i].x >= el) --i;
  const float4 c = C[i];
  const float wt = (el - c.x) * c.z;
  const float den = fmaf(c.w, wt, c.y * (1.0f - wt));  // jax: 0 + y (1 - wt), then + w wt
  return (el >= el_first && el <= el_last) ? den : __builtin_nanf("");
}

// Per-thread part of the K_RJ conversion that does not depend on the detector row.
struct KrjSamples {
  float x0, x3;    // boresight elevation of the thread's first and last sample minus the tile's reference elevation (CalDet::ebm)
  int curved;      // some thread of the workgroup: its four boresight elevations are NOT linear in the sample index to 1e-6 rad
};

// The K_RJ values of a thread's 4 consecutive samples of one detector.  `sv` already carries the detector's scale.
// Round 6, two forms:
//  * the usual row (CalDet::lin): den is linear in the boresight elevation over the whole tile, with at most one kink --
//    its value at the thread's first and last sample, two reciprocals, the inner two samples on the chord between them
//    (the elevation is linear in the sample index to ~5e-8 rad over four samples -- 10 ms of scanning, over which den
//    itself moves by ~3e-6 of its value; a kink inside those 10 ms is missed by 1e-8 of den);
//  * every other row -- near the zenith (CalDet::exact), off the axis (NaN, as jax fills), a tile that sweeps more than two
//    cells, a workgroup whose boresight is not linear over a thread's four samples (KrjSamples::curved: 20 Hz, a 0.1 deg
//    daisy at 0.8 deg/s) -- one sample at a time at its own boresight elevation through den_lookup.  That loop is not
//    unrolled and reloads what it needs: it must not set the register count of the kernels it is inlined into (the noise
//    writer ran at five waves per SIMD for it, the pW instance at eight).
// Rounds 2-5 had a third form between them (per thread: the cell of its first sample, checks that both ends sit in it,
// the per-sample loop for the threads that straddle a node): 21 instructions a sample wherever the division rode on
// another kernel's store, and the registers of all three.
template <bool kInverse = false>
__device__ __forceinline__ void krj_row(const CalDet& c, const float4* C, int n_el, const KrjSamples& k,
                                        const float (&sv)[kSamplesPerThread], float (&o)[kSamplesPerThread],
                                        const float* __restrict__ bore_el, int sb, int T, const float4* cells0,
                                        const float* __restrict__ cal_axis) {
  MRX_KRJ_FP
  constexpr int kL = kSamplesPerThread - 1;
  if (__builtin_amdgcn_readfirstlane(c.lin) != 0 && !k.curved) {  // (a scalar branch: the row is the workgroup's)
    const float d0 = fmaf(c.lin_db, fmaxf(k.x0 - c.lin_xn, 0.0f), fmaf(c.lin_b, k.x0, c.lin_a));
    const float d3 = fmaf(c.lin_db, fmaxf(k.x3 - c.lin_xn, 0.0f), fmaf(c.lin_b, k.x3, c.lin_a));
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
    return;
  }
  const float el_first = cells0[0].x, el_inv = cells0[0].z, el_last = cal_axis[n_el - 1];
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
}

// Shared prologue of the two K_RJ kernels, per workgroup: stage the cell table, reduce the
// boresight elevation range of the tile's 1024 samples (red[8] = lo, red[9] = hi) and return
// this thread's sample constants.  Ends with a barrier.
__device__ __forceinline__ KrjSamples krj_prologue(float4* cells, float* red, const float* __restrict__ bore_el,
                                                   int T, int sb, const float* __restrict__ cal_axis,
                                                   const float* __restrict__ cal_values, int n_el, int n_bands) {
  MRX_KRJ_FP
  KrjSamples k;
  const float eb0 = bore_el[min(sb, T - 1)], eb3 = bore_el[min(sb + kSamplesPerThread - 1, T - 1)];
  float eb_lo, eb_hi;
  {
    static_assert(kSamplesPerThread == 4, "the curvature check below is written for four samples");
    const float eb1 = bore_el[min(sb + 1, T - 1)], eb2 = bore_el[min(sb + 2, T - 1)];
    const float third = (eb3 - eb0) * (1.0f / 3.0f);
    k.curved = !(fabsf(eb1 - (eb0 + third)) <= 1.0e-6f && fabsf(eb2 - fmaf(2.0f, third, eb0)) <= 1.0e-6f);  // (a NaN: per sample too)
    eb_lo = fminf(fminf(eb0, eb1), fminf(eb2, eb3));
    eb_hi = fmaxf(fmaxf(eb0, eb1), fmaxf(eb2, eb3));
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
  k.x0 = eb0 - ebm;
  k.x3 = eb3 - ebm;
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
                                               const float* __restrict__ scale, int n_bands, int d0, int nd,
                                               const float4* cells, int n_el) {
  MRX_KRJ_FP
  if ((int)threadIdx.x < kTileDet) {
    const float lo = red[8], hi = red[9];
    bool exact = false;
    if ((int)threadIdx.x < nd) {
      const int d = d0 + threadIdx.x;
      CalDet c = make_cal_det(dxs[d], dys[d], min(max(band[d], 0), n_bands - 1), scale ? scale[d] : 1.0f);
      set_elevation_model(c, 0.5f * (lo + hi), bore_trig(0.5f * (lo + hi)));
      // the model is a finite difference over ebm +- 0.02 rad: a tile whose boresight sweeps
      // farther (slow sample rates, fast elevation slews) takes the full formula per sample
      if (!(hi - lo <= 2.0f * kModelHalfRange)) c.exact = 1;
      set_linear_den(c, lo, hi, cells + c.band * (n_el - 1), n_el, cells[0].x, cells[0].z);
      cdet[threadIdx.x] = c;
      exact = c.exact != 0;
    }
    const bool any = __builtin_amdgcn_ballot_w64(exact) != 0;  // the 16 lanes sit in wave 0
    if (threadIdx.x == 0) red[10] = any ? 1.0f : 0.0f;
  }
}

}  // namespace
