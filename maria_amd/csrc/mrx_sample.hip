// Fused coarse-rate atmosphere sampling for gfx950.
//
// One thread = one (detector, coarse time step).  Lanes of a wave are
// consecutive detectors at the same time step, so everything that depends on
// time only (boresight, wind offsets, layer descriptors) is wave-uniform and
// sits in SGPRs, the time-major stores are fully coalesced, and the bilinear
// gathers of neighbouring detectors land in the same few cache lines of the
// screen (a focal plane is compact on every layer).
//
// Arithmetic follows the reference's rounding points (float32 where jax
// computes, float64 where numpy does); see oracle/hotpath.py for the CPU
// restatement this kernel is checked against.  This TU is compiled with
// -ffp-contract=off so that a*b+c rounds twice, like numpy / XLA do.
#include "mrx_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTimesPerBlock = 4;

// float32(pi/2): jax folds the weak-typed python float pi/2 to float32
// (coords/transforms.py:22) and numpy clips float32 elevations to it
// (sim/atmosphere.py:60).
constexpr float kHalfPiF = 1.57079637050628662109375f;

struct Cell {
  int i;
  float w;    // normalised distance to the lower node
  bool oob;
};

// jax.scipy.interpolate.RegularGridInterpolator._find_indices for one axis:
// i = clip(searchsorted(g, x, side="left") - 1, 0, n-2), w = (x-g[i])/(g[i+1]-g[i]),
// out of bounds when x < g[0] or x > g[n-1].  `guess` starts the search (the
// screens' axes are uniform up to float32 rounding, so it is off by at most one).
__device__ __forceinline__ Cell find_cell_guess(const float* __restrict__ g,
                                                int n, float x, int guess) {
  int i = min(max(guess, 0), n - 2);
  float lo = g[i], hi = g[i + 1];
  while (i < n - 2 && hi < x) {
    ++i;
    lo = hi;
    hi = g[i + 1];
  }
  while (i > 0 && lo >= x) {
    --i;
    hi = lo;
    lo = g[i];
  }
  Cell c;
  c.i = i;
  c.w = (x - lo) / (hi - lo);
  c.oob = !(x >= g[0] && x <= g[n - 1]);  // also true for NaN
  return c;
}

// Same contract, by bisection: for the (short, possibly non-uniform) axes of
// the emission tables.
__device__ __forceinline__ Cell find_cell_bisect(const float* __restrict__ g,
                                                 int n, float x) {
  int lo = 0, hi = n;  // first index with g[k] >= x
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (g[mid] < x) lo = mid + 1; else hi = mid;
  }
  int i = min(max(lo - 1, 0), n - 2);
  float a = g[i], b = g[i + 1];
  Cell c;
  c.i = i;
  c.w = (x - a) / (b - a);
  c.oob = !(x >= g[0] && x <= g[n - 1]);
  return c;
}

__global__ __launch_bounds__(kBlock) void atm_sample_kernel(
    const mrx_layer* __restrict__ layers, int n_layers,
    const mrx_band_table* __restrict__ tables, int n_tables,
    const float* __restrict__ az, const float* __restrict__ el, int Ta,
    const float* __restrict__ dxs, const float* __restrict__ dys,
    const int32_t* __restrict__ band, const float* __restrict__ mueller00,
    int D, double pwv0, double* __restrict__ pwv_out,
    float* __restrict__ loading, uint32_t* __restrict__ flags) {
  const int d = blockIdx.x * kBlock + threadIdx.x;
  const bool live = d < D;
  const int dd = live ? d : D - 1;  // keep addresses valid; stores are masked

  // ---- per-detector constants (coords/transforms.py:14-23), float32 ------
  const float dx = dxs[dd], dy = dys[dd];
  const float r = sqrtf(dx * dx + dy * dy);
  const float p = atan2f(-dx, -dy);
  const float sr = sinf(r), cr = cosf(r);
  const float sp = sinf(p), cp = cosf(p);
  const float A = sr * cp;  // sin(r) cos(p): real part before the tilt
  const float Y = sr * sp;  // sin(r) sin(p)
  const int b = band[dd];
  const float m00 = mueller00[dd];
  const mrx_band_table tb = tables[min(max(b, 0), n_tables - 1)];

  uint32_t myflags = (b < 0 || b >= n_tables) ? MRX_FLAG_NAN : 0u;

  const int t0 = blockIdx.y * kTimesPerBlock;
  for (int tt = 0; tt < kTimesPerBlock; ++tt) {
    const int t = t0 + tt;
    if (t >= Ta) break;

    // ---- pointing of this detector (transforms.py:20-28) ------------------
    const float a = el[t] - kHalfPiF;
    const float ca = cosf(a), sa = sinf(a);
    const float re = A * ca - cr * sa;
    const float im = A * sa + cr * ca;
    const float phi = atan2f(Y, re) + az[t];
    const float theta = asinf(im);

    // ---- unit-height ground projection (coordinates.py:339-347) -----------
    // numpy float32 tan/cos/sin and float32 division, then float64.
    const float tth = tanf(theta);
    const double px = (double)(cosf(phi) / tth);
    const double py = (double)(sinf(phi) / tth);

    // ---- layer stack (atmosphere/atmosphere.py:317-373) -------------------
    double pwv = pwv0;
    for (int l = 0; l < n_layers; ++l) {
      const mrx_layer& ly = layers[l];
      const double e64 = fma(ly.h, fma(px, ly.r00, py * ly.r10), ly.d_off_e[t]);
      const double c64 = fma(ly.h, fma(px, ly.r01, py * ly.r11), ly.d_off_c[t]);
      const float xe = (float)e64, xc = (float)c64;

      const float* ge = ly.d_axis_e;
      const float* gc = ly.d_axis_c;
      // uniform-grid guess from the first two nodes
      const float e0 = ge[0], c0 = gc[0];
      const float ide = 1.0f / (ge[1] - e0), idc = 1.0f / (gc[1] - c0);
      const float fe = (xe - e0) * ide, fc = (xc - c0) * idc;
      // a NaN or huge coordinate must not become an undefined int conversion
      const int gi = (int)fminf(fmaxf(fe, -1.0f), 2.0e9f) ;
      const int gj = (int)fminf(fmaxf(fc, -1.0f), 2.0e9f);
      const Cell ce = find_cell_guess(ge, ly.n_e, xe, gi);
      const Cell cc = find_cell_guess(gc, ly.n_c, xc, gj);

      const float* v = ly.d_values + (size_t)ce.i * ly.n_c + cc.i;
      const float v00 = v[0], v01 = v[1];
      const float v10 = v[ly.n_c], v11 = v[ly.n_c + 1];
      // jax _evaluate_linear: edges in itertools.product order, weight built
      // as (1*we)*wc, summed into 0.0 in float32.
      const float we0 = 1.0f - ce.w, we1 = ce.w;
      const float wc0 = 1.0f - cc.w, wc1 = cc.w;
      float y = 0.0f;
      y = y + v00 * (we0 * wc0);
      y = y + v01 * (we0 * wc1);
      y = y + v10 * (we1 * wc0);
      y = y + v11 * (we1 * wc1);
      if (ce.oob || cc.oob) {
        y = __builtin_nanf("");
        myflags |= MRX_FLAG_SCREEN_OOB;
      }
      // layer.pwv_rms * y is a float32 product (jax array), accumulated into
      // the float64 numpy array (atmosphere.py:373).
      pwv += (double)(ly.pwv_rms * y);
    }

    // ---- band emission (band/band.py:264-286), float32 --------------------
    const float xp = (float)pwv;
    const float xel = fminf(theta, kHalfPiF);  // .clip(max=pi/2), sim/atmosphere.py:60
    const Cell cp_ = find_cell_bisect(tb.d_axis_pwv, tb.n_pwv, xp);
    const Cell cl = find_cell_bisect(tb.d_axis_el, tb.n_el, xel);
    const size_t slab = (size_t)tb.n_pwv * tb.n_el;
    const float* q = tb.d_values + (size_t)cp_.i * tb.n_el + cl.i;
    float val = 0.0f;
#pragma unroll
    for (int ia = 0; ia < 2; ++ia) {
      const float w1 = 1.0f * (ia ? tb.w_t : 1.0f - tb.w_t);
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        const float w2 = w1 * (ib ? cp_.w : 1.0f - cp_.w);
#pragma unroll
        for (int ic = 0; ic < 2; ++ic) {
          const float w3 = w2 * (ic ? cl.w : 1.0f - cl.w);
          val = val + q[ia * slab + (size_t)ib * tb.n_el + ic] * w3;
        }
      }
    }
    if (cp_.oob || cl.oob || tb.t_oob) {
      val = __builtin_nanf("");
      myflags |= MRX_FLAG_TABLE_OOB;
    }
    const float out = m00 * val;
    if (out != out) myflags |= MRX_FLAG_NAN;

    if (live) {
      const size_t o = (size_t)t * D + d;
      loading[o] = out;
      if (pwv_out) pwv_out[o] = pwv;
    }
  }
  if (live && myflags) atomicOr(flags, myflags);
}

}  // namespace

extern "C" int mrx_atm_sample(mrx_ctx* ctx, const mrx_atm_plan* plan,
                              const float* d_az, const float* d_el, int Ta,
                              const float* d_dx, const float* d_dy,
                              const int32_t* d_band,
                              const float* d_mueller00, int D, double pwv0,
                              double* d_pwv, float* d_loading,
                              uint32_t* d_flags) {
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && Ta >= 0, "negative size");
  if (D == 0 || Ta == 0) return MRX_OK;  // empty shard: nothing to do
  MRX_REQUIRE(ctx, plan != nullptr, "plan is null");
  MRX_REQUIRE(ctx, d_az && d_el && d_dx && d_dy && d_band && d_mueller00,
              "null input pointer");
  MRX_REQUIRE(ctx, d_loading && d_flags, "null output pointer");
  dim3 grid(mrx_ceil_div(D, kBlock), mrx_ceil_div(Ta, kTimesPerBlock));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "Ta too large for one launch");
  hipLaunchKernelGGL(atm_sample_kernel, grid, dim3(kBlock), 0, ctx->stream,
                     plan->d_layers, plan->n_layers, plan->d_tables,
                     plan->n_tables, d_az, d_el, Ta, d_dx, d_dy, d_band,
                     d_mueller00, D, pwv0, d_pwv, d_loading, d_flags);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}
