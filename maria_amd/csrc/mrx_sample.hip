// Fused coarse-rate atmosphere sampling for gfx950, and its device-side plan.
//
// One thread = one detector x kTimes consecutive coarse time steps.  Lanes of a
// wave are consecutive detectors at the same time steps, so everything that
// depends on time only (boresight, wind offsets, layer descriptors) is
// wave-uniform and sits in SGPRs, the time-major stores are fully coalesced,
// and the bilinear gathers of neighbouring detectors land in the same few
// cache lines of the screen (a focal plane is compact on every layer).
//
// The kernel is latency-, not bandwidth-bound, so the plan is laid out to keep
// dependent loads off the critical path:
//   * layer descriptors are packed scalars (one s_load burst per layer, reused
//     for kTimes time steps);
//   * the per-time wind offsets of all layers are packed [t][layer];
//   * on a uniform axis -- one whose float32 nodes a device-side check at plan
//     creation finds equal to float32(g0 + i*dg) bit for bit -- cell and weight
//     come from the line of sight's position in PIXELS, evaluated in float64
//     (two fused multiply-adds per axis); otherwise the axis array is searched
//     with jax's float32 rule (MRX_OPT_AXIS_LITERAL forces that everywhere);
//   * the band tables and their axes are staged in LDS once per workgroup.
// Per layer and sample that leaves only the 2x2 gather itself.
//
// Arithmetic follows the reference's rounding points (float32 where jax
// computes, float64 where numpy does); see oracle/hotpath.py for the CPU
// restatement this kernel is checked against.  This TU is compiled with
// -ffp-contract=off so that a*b+c rounds twice, like numpy / XLA do.
#include <algorithm>
#include <vector>

#include "mrx_internal.h"

#include "mrx_sample_px.h"

using namespace mrx_px;

namespace {

constexpr int kBlock = kPxBlock;
constexpr int kTimes = 1;                  // coarse time steps per thread (default; measured best)

}  // namespace

namespace {


// jax.scipy.interpolate.RegularGridInterpolator._find_indices for one axis:
// i = clip(searchsorted(g, x, side="left") - 1, 0, n-2), w = (x-g[i])/(g[i+1]-g[i]),
// out of bounds when x < g[0] or x > g[n-1].  NodeFn returns float32 node i.
template <typename NodeFn>
__device__ __forceinline__ Cell find_cell(NodeFn node, int n, float x,
                                          float first, float inv, float last) {
  // a NaN or huge coordinate must not become an undefined int conversion
  const float f = fminf(fmaxf((x - first) * inv, -1.0f), 2.0e9f);
  int i = min(max((int)f, 0), n - 2);
  float lo = node(i), hi = node(i + 1);
  while (i < n - 2 && hi < x) {
    ++i;
    lo = hi;
    hi = node(i + 1);
  }
  while (i > 0 && lo >= x) {
    --i;
    hi = lo;
    lo = node(i);
  }
  Cell c;
  c.i = i;
  c.w = (x - lo) / (hi - lo);
  c.oob = !(x >= first && x <= last);  // also true for NaN
  return c;
}

typedef __attribute__((address_space(1))) const float gfloat;  // global memory
typedef __attribute__((address_space(1))) const char char_g;


// One-probe variant: takes the arithmetic guess, fetches the two nodes that
// bracket it (PairFn: recomputed, or loaded from the axis array) and reports in
// `miss` whether the guess was not the searchsorted cell -- x within float32
// rounding of a node (~1e-4 of the samples), or a non-uniform axis; the caller
// then redoes that sample through find_cell.  Straight-line code, so the
// compiler interleaves the kTimes samples of a thread.  kExactDiv selects the
// IEEE division the reference performs; otherwise x * v_rcp_f32 (1 ulp).
template <bool kExactDiv, typename PairFn>
__device__ __forceinline__ Cell probe_cell(PairFn pair, int n, float x,
                                           float first, float inv, float last,
                                           bool& miss) {
  const float f = fminf(fmaxf((x - first) * inv, -1.0f), 2.0e9f);
  const int i = min(max((int)f, 0), n - 2);
  float lo, hi;
  pair(i, lo, hi);
  miss |= (i < n - 2 && hi < x) || (i > 0 && lo >= x);
  Cell c;
  c.i = i;
  c.w = kExactDiv ? (x - lo) / (hi - lo)
                  : (x - lo) * __builtin_amdgcn_rcpf(hi - lo);
  c.oob = !(x >= first && x <= last);
  return c;
}

// The 2x2 gather and blend of jax _evaluate_linear: corners in
// itertools.product order, weight built as (1*we)*wc, summed into 0.0 in
// float32; NaN outside the grid.
__device__ __forceinline__ float bilinear(gfloat* values, int nc,
                                          const Cell& ce, const Cell& cc) {
  gfloat* v = values + (size_t)ce.i * nc + cc.i;
  // each row's two corners as ONE 8-byte load (4-byte aligned is enough for
  // global_load_dwordx2): the kernel is bound by vector-memory instruction issue
  const pair4 r0 = *(gpair*)v;
  const pair4 r1 = *(gpair*)(v + nc);
  const float v00 = r0.x, v01 = r0.y;
  const float v10 = r1.x, v11 = r1.y;
  const float we0 = 1.0f - ce.w, we1 = ce.w;
  const float wc0 = 1.0f - cc.w, wc1 = cc.w;
  float y = 0.0f;
  y = y + v00 * (we0 * wc0);
  y = y + v01 * (we0 * wc1);
  y = y + v10 * (we1 * wc0);
  y = y + v11 * (we1 * wc1);
  return (ce.oob || cc.oob) ? __builtin_nanf("") : y;
}

// Band emission of one (detector, step) (band/band.py:264-300) and the Mueller weight
// (sim/atmosphere.py:64-65): the float32 trilinear lookup in the reference's summation order,
// or the float64 bicubic of interpolation_method="cubic".  `tdata` holds the band tables (LDS or
// global).  in_t: the step exists (flags are only raised for real steps).
// kCubicPath = false: a kernel instance for plans without cubic tables (the code is not compiled in).
template <bool kCubicPath = true>
__device__ __forceinline__ float band_loading(const mrx_table_dev& tb, const float* __restrict__ tdata, double pwv,
                                              float theta, float m00, bool in_t, uint32_t& iflags) {
  const float* __restrict__ ax_p = tdata + tb.off_pwv;
  const float* __restrict__ ax_e = tdata + tb.off_el;
  const float* __restrict__ tv = tdata + tb.off_values;
  const int slab = tb.n_pwv * tb.n_el;
  const float p_last = ax_p[tb.n_pwv - 1], e_last = ax_e[tb.n_el - 1];
  const float xp = (float)pwv;
  const float xel = fminf(theta, kHalfPiF);  // .clip(max=pi/2), sim/atmosphere.py:60
  const Cell cp_ = find_cell([=](int i) { return ax_p[i]; }, tb.n_pwv, xp,
                             tb.p_first, tb.p_inv, p_last);
  const Cell cl = find_cell([=](int i) { return ax_e[i]; }, tb.n_el, xel,
                            tb.e_first, tb.e_inv, e_last);
  const float* q = tv + cp_.i * tb.n_el + cl.i;
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
        val = val + q[ia * slab + ib * tb.n_el + ic] * w3;
      }
    }
  }
  if ((!kCubicPath || !tb.cubic) && (cp_.oob || cl.oob || tb.t_oob)) {
    val = __builtin_nanf("");
    if (in_t) iflags |= MRX_FLAG_TABLE_OOB;
  }
  float out = m00 * val;
  if (kCubicPath && tb.cubic) {
    // interpolation_method="cubic" (band/band.py:288-300): scipy's tensor-product cubic
    // spline on (pwv, el) in float64, expanded by the host into a bicubic per cell; the
    // float64 product with the Mueller weight is rounded once (sim/atmosphere.py:64-65)
    const double* __restrict__ X = tb.cubic;
    const double* __restrict__ Y = X + tb.n_pwv;
    const double* __restrict__ Cc = Y + tb.n_el;
    const double xd = pwv, yd = (double)xel;
    int i = min(max(cp_.i, 0), tb.n_pwv - 2), j = min(max(cl.i, 0), tb.n_el - 2);
    while (i < tb.n_pwv - 2 && X[i + 1] <= xd) ++i;  // the float32 cell is a guess for the float64 axis
    while (i > 0 && X[i] > xd) --i;
    while (j < tb.n_el - 2 && Y[j + 1] <= yd) ++j;
    while (j > 0 && Y[j] > yd) --j;
    const double u = xd - X[i], v = yd - Y[j];
    const double* __restrict__ c = Cc + ((size_t)i * (tb.n_el - 1) + j) * 16;
    double acc = 0.0;
#pragma unroll
    for (int k = 3; k >= 0; --k) {
      const double row = fma(fma(fma(c[4 * k + 3], u, c[4 * k + 2]), u, c[4 * k + 1]), u, c[4 * k]);
      acc = fma(acc, v, row);
    }
    const bool inside = xd >= X[0] && xd <= X[tb.n_pwv - 1] && yd >= Y[0] && yd <= Y[tb.n_el - 1];
    out = inside ? (float)((double)m00 * acc) : __builtin_nanf("");
    if (!inside && in_t) iflags |= MRX_FLAG_TABLE_OOB;
  }
  if (out != out && in_t) iflags |= MRX_FLAG_NAN;
  return out;
}

// kChain = true follows the reference's float32 chain literally (atan2 -> phi,
// asin -> theta, then tan/cos/sin of those: coords/transforms.py:20-28 and
// coordinates.py:339-347).  kChain = false (default) uses the identity behind
// that chain: (Y, re, im) is the unit line-of-sight vector in the local frame
// before the azimuth turn, so the unit-height ground projection is
//     px = (re cos az - Y sin az)/im,   py = (Y cos az + re sin az)/im
// with no inverse trigonometry and no cancellation near the zenith; only
// theta = asin(im) is still needed (elevation axis of the emission table).  The
// two differ by the float32 rounding noise of the chain itself (<= ~1e-6
// relative in px, py near the zenith), far inside the 1e-5 parity tolerance.
template <bool kLdsTables, bool kChain, int kT>
// kT = 1: 8 waves per SIMD (<= 64 VGPRs): alone on the chip the kernel hides its gather latency
// with occupancy.  kT = 2, 4: two or four time steps of a detector interleaved in one thread at 5
// waves per SIMD (<= 96 VGPRs) -- for the small resident grid that runs beside the TOD writer
// (3 workgroups per CU: 3 x 96 registers for the sampler, 3 x 72 for the writer), where
// instruction-level parallelism has to stand in for the waves the writer's share takes away
// (pipelined step 2.73 -> 2.62 ms; at 64 registers these instances spill 41 / 77 values)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kT == 1 ? 8 : 5, kT == 1 ? 8 : 5))) void atm_sample_kernel(
    const mrx_layer_dev* __restrict__ layers, const mrx_layer_fast* __restrict__ fast, int n_layers,
    const double2* __restrict__ off, const double2* __restrict__ offpx,
    const mrx_table_dev* __restrict__ tables,
    int n_tables, const float* __restrict__ table_data, int table_floats,
    const float* __restrict__ az, const float* __restrict__ el, int Ta,
    const float* __restrict__ dxs, const float* __restrict__ dys,
    const int32_t* __restrict__ band, const float* __restrict__ mueller00,
    int D, double pwv0, double* __restrict__ pwv_out,
    float* __restrict__ loading, uint32_t* __restrict__ flags,
    int force_arrays, int chunk, int nbx, int n_items) {
  // A work item = 256 detectors x `chunk` consecutive time steps (a multiple of kT,
  // <= kMaxChunk), walked kT steps at a time: the per-detector constants are paid once per
  // item.  Items are dealt to workgroups round-robin (item = blockIdx.x, += gridDim.x), so
  // the launch may be the whole item list or a small resident grid that leaves most wave
  // slots of every CU to a concurrent HBM-bound kernel (MRX_OPT_SAMPLE_WGS_PER_CU).
  extern __shared__ float lds_tables[];
  __shared__ float4 bore[kMaxChunk];  // per time step: cos/sin of (el - pi/2), az
  if (kLdsTables)
    for (int i = threadIdx.x; i < table_floats; i += kBlock)
      lds_tables[i] = table_data[i];
  const float* __restrict__ tdata = kLdsTables ? lds_tables : table_data;
  uint32_t myflags = 0u;

  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
  const int t_first = (item / nbx) * chunk;
  uint32_t iflags = 0u;
  __syncthreads();  // the previous item's readers of bore[] are done
  if ((int)threadIdx.x < chunk) {
    const int t = min(t_first + (int)threadIdx.x, Ta - 1);
    const float a = el[t] - kHalfPiF;  // transforms.py:22
    const float z = az[t];
    bore[threadIdx.x] = make_float4(cosf(a), sinf(a), cosf(z), sinf(z));
  }
  __syncthreads();

  const int d = (item % nbx) * kBlock + threadIdx.x;
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
  if (live && (b < 0 || b >= n_tables)) iflags |= MRX_FLAG_NAN;
  const mrx_table_dev tb = tables[min(max(b, 0), n_tables - 1)];

  for (int it = 0; it < chunk && t_first + it < Ta; it += kT) {
  const int t0 = t_first + it;
  // ---- pointing and unit-height ground projection for kT steps --------
  float theta[kT];
  double px[kT], py[kT], pwv[kT];
#pragma unroll
  for (int tt = 0; tt < kT; ++tt) {
    // transforms.py:20-28
    const float4 bt = bore[it + tt];
    const float re = A * bt.x - cr * bt.y;
    const float im = A * bt.y + cr * bt.x;
    theta[tt] = asinf(im);
    if (kChain) {
      const int t = min(t0 + tt, Ta - 1);
      const float phi = atan2f(Y, re) + az[t];
      // coordinates.py:339-347: numpy float32 tan/cos/sin and division, then f64
      const float tth = tanf(theta[tt]);
      px[tt] = (double)(cosf(phi) / tth);
      py[tt] = (double)(sinf(phi) / tth);
    } else {
      const float inv_im = 1.0f / im;
      px[tt] = (double)((re * bt.z - Y * bt.w) * inv_im);
      py[tt] = (double)((Y * bt.z + re * bt.w) * inv_im);
    }
    pwv[tt] = pwv0;
  }

  // ---- layer stack (atmosphere/atmosphere.py:317-373) ---------------------
  for (int l = 0; l < n_layers; ++l) {
    const mrx_layer_fast lf = fast[l];
    if (lf.pixel && !force_arrays) {  // wave-uniform
      // Uniform axes: the position on the grid in pixels, in float64.  The reference rounds
      // the metre coordinate to float32 first (jax) and divides float32 differences; at 10 km
      // from the origin that quantises the position to ~2e-4 pixel.  Here the weight keeps
      // the float64 position -- closer to the exact bilinear value than the reference's own
      // arithmetic, and within ~1e-8 of it in the loading -- for half the instructions: no
      // node fetch, no cell search, no division.
      gfloat* values = (gfloat*)lf.values;
#pragma unroll
      for (int tt = 0; tt < kT; ++tt) {
        const int t = min(t0 + tt, Ta - 1);
        const double2 o = offpx[(size_t)t * n_layers + l];
        const double fe = fma(px[tt], lf.pe_x, fma(py[tt], lf.pe_y, o.x));
        const double fc = fma(px[tt], lf.pc_x, fma(py[tt], lf.pc_y, o.y));
        Cell ce, cc;
        ce.i = min(max(__double2int_rz(fe), 0), lf.n_e - 2);
        cc.i = min(max(__double2int_rz(fc), 0), lf.n_c - 2);
        ce.w = (float)(fe - (double)ce.i);
        cc.w = (float)(fc - (double)cc.i);
        // inside the grid: 0 <= f <= n - 1 (the last node belongs to the last cell); NaN is outside
        ce.oob = !(ce.w >= 0.0f && ce.w <= 1.0f);
        cc.oob = !(cc.w >= 0.0f && cc.w <= 1.0f);
        const float yv = bilinear(values, lf.n_c, ce, cc);
        pwv[tt] += (double)(lf.pwv_rms * yv);  // a NaN (line of sight off the screen) stays in the sum
      }
      continue;
    }
    const mrx_layer_dev& ly = layers[l];
    const int nc = ly.n_c;
    float xe[kT], xc[kT], y[kT];
#pragma unroll
    for (int tt = 0; tt < kT; ++tt) {
      const int t = min(t0 + tt, Ta - 1);
      const double2 o = off[(size_t)t * n_layers + l];
      if (kChain) {  // h * (p @ R) + offset, as numpy evaluates it
        xe[tt] = (float)fma(ly.h, fma(px[tt], ly.r00, py[tt] * ly.r10), o.x);
        xc[tt] = (float)fma(ly.h, fma(px[tt], ly.r01, py[tt] * ly.r11), o.y);
      } else {       // the same to float64 rounding, one operation less per axis
        xe[tt] = (float)fma(px[tt], ly.hr00, fma(py[tt], ly.hr10, o.x));
        xc[tt] = (float)fma(px[tt], ly.hr01, fma(py[tt], ly.hr11, o.y));
      }
    }
    gfloat* values = (gfloat*)ly.values;
    gfloat* axis_e = (gfloat*)ly.axis_e;
    gfloat* axis_c = (gfloat*)ly.axis_c;
    bool miss[kT];
    {
    // nodes fetched from the caller's axis arrays (L1-resident)
#pragma unroll
    for (int tt = 0; tt < kT; ++tt) {
      miss[tt] = false;
      const Cell ce = probe_cell<kChain>(
          [=](int i, float& lo, float& hi) { lo = axis_e[i]; hi = axis_e[i + 1]; },
          ly.n_e, xe[tt], ly.e_first, ly.e_inv, ly.e_last, miss[tt]);
      const Cell cc = probe_cell<kChain>(
          [=](int i, float& lo, float& hi) { lo = axis_c[i]; hi = axis_c[i + 1]; },
          nc, xc[tt], ly.c_first, ly.c_inv, ly.c_last, miss[tt]);
      y[tt] = bilinear(values, nc, ce, cc);
    }
    }
#pragma unroll
    for (int tt = 0; tt < kT; ++tt)
      if (__builtin_amdgcn_ballot_w64(miss[tt]) != 0) {
        // some lane's guess was not the cell: full search on the axis arrays
        const Cell ce = find_cell([=](int i) { return axis_e[i]; }, ly.n_e,
                                  xe[tt], ly.e_first, ly.e_inv, ly.e_last);
        const Cell cc = find_cell([=](int i) { return axis_c[i]; }, nc, xc[tt],
                                  ly.c_first, ly.c_inv, ly.c_last);
        const float ys = bilinear(values, nc, ce, cc);
        if (miss[tt]) y[tt] = ys;
      }
#pragma unroll
    for (int tt = 0; tt < kT; ++tt) {
      // layer.pwv_rms * y is a float32 product (jax array), accumulated into
      // the float64 numpy array (atmosphere.py:373).
      pwv[tt] += (double)(ly.pwv_rms * y[tt]);
    }
  }
  // only a line of sight off a screen (jax's NaN fill) makes the sum NaN: one test per step
  // instead of one per layer (pwv0 and the screens are finite)
#pragma unroll
  for (int tt = 0; tt < kT; ++tt)
    if (pwv[tt] != pwv[tt] && t0 + tt < Ta) iflags |= MRX_FLAG_SCREEN_OOB;

  // ---- band emission (band/band.py:264-300) and Mueller weight -----------
#pragma unroll
  for (int tt = 0; tt < kT; ++tt) {
    const int t = t0 + tt;
    const float out = band_loading(tb, tdata, pwv[tt], theta[tt], m00, t < Ta, iflags);
    if (live && t < Ta) {
      const size_t o = (size_t)t * D + d;
      loading[o] = out;
      if (pwv_out) pwv_out[o] = pwv[tt];
    }
  }
  }  // chunk loop
  if (live) myflags |= iflags;
  }  // item loop
  if (myflags) atomicOr(flags, myflags);
}

template <bool kLdsTables, int kT, bool kPipe>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kT == 1 ? (kPipe ? kPxWaves : 8) : 5, kT == 1 ? (kPipe ? kPxWaves : 8) : 5))) void atm_sample_px_kernel(
    const mrx_layer_fast* __restrict__ fast, const mrx_layer_px* __restrict__ lpx, int n_layers,
    const double2* __restrict__ offpx, const mrx_table_dev* __restrict__ tables, int n_tables,
    const float* __restrict__ table_data, int table_floats, const float* __restrict__ az,
    const float* __restrict__ el, int Ta, const float* __restrict__ dxs, const float* __restrict__ dys,
    const int32_t* __restrict__ band, const float* __restrict__ mueller00, int D, double pwv0,
    double* __restrict__ pwv_out, float* __restrict__ loading, uint32_t* __restrict__ flags, int chunk,
    int nby) {
  extern __shared__ __align__(16) float4 lds_px[];
  // one block of D rows, nobody to tell: the body is mrx_sample_px.h's
  PxNoHooks hooks;
  px_sample_items<kLdsTables, kT, kPipe, false>(fast, lpx, n_layers, offpx, tables, n_tables, table_data, table_floats, az, el, Ta,
                                         dxs, dys, band, mueller00, D, pwv0, pwv_out, loading, flags, chunk, nby, D, 1, 0, 1,
                                         (int)blockIdx.x, (int)gridDim.x, lds_px, hooks);
}

// ---- plan construction ------------------------------------------------------

// Fills the axis-derived fields of each layer descriptor and checks whether
// float32(g0 + i*dg) reproduces the axis arrays exactly.
__global__ void plan_finish_layers(mrx_layer_dev* layers, mrx_layer_fast* fast, mrx_layer_px* px, int n_layers) {
  const int l = blockIdx.x;
  mrx_layer_dev& ly = layers[l];
  __shared__ int ok_e, ok_c;
  if (threadIdx.x == 0) {
    ok_e = ly.de != 0.0;
    ok_c = ly.dc != 0.0;
  }
  __syncthreads();
  // A node may sit one float32 ulp off float32(g0 + i*dg): the caller's float64 grid is itself rounded (np.arange /
  // np.linspace: start + i*step), and where that value falls within ~1e-11 of the midpoint of two float32 numbers the
  // two roundings part -- one node in ~1e5 at 4096 nodes 70 km from the origin (BASELINE config 5: two of its 32 axes,
  // which sent the whole plan to the general kernel).  An ulp of a node is ~1e-3 pixel there, the rounding the
  // reference's own float32 coordinate carries (MRX_OPT_AXIS_LITERAL), so such an axis is uniform for this purpose.
  auto off_grid = [](double g0, double dg, int i, float node) {
    const float a = (float)((double)i * dg + g0);
    return !(fabsf(a - node) <= fmaxf(fmaxf(fabsf(a), fabsf(node)), (float)fabs(dg)) * 1.1920929e-7f);  // 2^-23; NaN is off
  };
  int bad_e = 0, bad_c = 0;
  for (int i = threadIdx.x; i < ly.n_e; i += blockDim.x) bad_e |= off_grid(ly.e0, ly.de, i, ly.axis_e[i]);
  for (int i = threadIdx.x; i < ly.n_c; i += blockDim.x) bad_c |= off_grid(ly.c0, ly.dc, i, ly.axis_c[i]);
  if (bad_e) atomicAnd(&ok_e, 0);
  if (bad_c) atomicAnd(&ok_c, 0);
  __syncthreads();
  if (threadIdx.x == 0) {
    ly.uniform_e = ok_e;
    ly.uniform_c = ok_c;
    // inverse step for the cell guess: from the float64 hint when it verified
    // (the difference of two float32 nodes carries a ~1e-4 relative error, which
    // would put the guess in the wrong cell far from the origin), else the mean
    ly.e_first = ly.axis_e[0];
    ly.e_last = ly.axis_e[ly.n_e - 1];
    ly.e_inv = ok_e ? (float)(1.0 / ly.de)
                    : (float)((double)(ly.n_e - 1) /
                              ((double)ly.e_last - (double)ly.e_first));
    ly.c_first = ly.axis_c[0];
    ly.c_last = ly.axis_c[ly.n_c - 1];
    ly.c_inv = ok_c ? (float)(1.0 / ly.dc)
                    : (float)((double)(ly.n_c - 1) /
                              ((double)ly.c_last - (double)ly.c_first));
    mrx_layer_fast& f = fast[l];
    f.values = ly.values;
    f.pe_x = ly.pe_x; f.pe_y = ly.pe_y; f.pc_x = ly.pc_x; f.pc_y = ly.pc_y;
    f.n_e = ly.n_e; f.n_c = ly.n_c;
    f.pwv_rms = ly.pwv_rms;
    // the pixel kernel addresses a screen with a 32-bit byte offset formed by a 24-bit multiply, ie * (4 n_c):
    // sides below 2^22 and screens below 4 GiB (larger ones -- a caller's own 32768^2 -- take the general kernel)
    f.pixel = ok_e && ok_c && ly.de > 0.0 && ly.dc > 0.0 && ly.n_e < (1 << 22) && ly.n_c < (1 << 22) &&
              (unsigned long long)ly.n_e * (unsigned long long)ly.n_c * 4ull < (1ull << 32);
    mrx_layer_px& q = px[l];
    q.values = ly.values;
    q.values1 = ly.values + ly.n_c;
    q.pe_x = (float)ly.pe_x; q.pe_y = (float)ly.pe_y; q.pc_x = (float)ly.pc_x; q.pc_y = (float)ly.pc_y;
    q.nc4 = 4 * ly.n_c;
    q.pwv_rms = ly.pwv_rms;
    q.half_e = 0.5f * (float)(ly.n_e - 1); q.half_c = 0.5f * (float)(ly.n_c - 1);
    q.bytes = f.pixel ? (uint32_t)((unsigned long long)ly.n_e * (unsigned long long)ly.n_c * 4ull) : 0u;
    q.bytes1 = f.pixel ? q.bytes - (uint32_t)q.nc4 : 0u;
  }
}

__global__ void plan_pack_offsets(double2* off, double2* offpx, const mrx_layer_dev* layers,
                                  const double* off_e, const double* off_c, int l, int n_layers,
                                  int n_t) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_t) return;
  const mrx_layer_dev& ly = layers[l];
  off[(size_t)t * n_layers + l] = make_double2(off_e[t], off_c[t]);
  const bool uni = ly.uniform_e && ly.uniform_c && ly.de > 0.0 && ly.dc > 0.0;  // set by plan_finish_layers, earlier on this stream
  offpx[(size_t)t * n_layers + l] =
      uni ? make_double2((off_e[t] - ly.e0) / ly.de, (off_c[t] - ly.c0) / ly.dc) : make_double2(0.0, 0.0);
}

__global__ void plan_pack_table(float* data, mrx_table_dev* tables, int b,
                                const float* values, const float* axis_pwv,
                                const float* axis_el) {
  mrx_table_dev& tb = tables[b];
  const int nv = 2 * tb.n_pwv * tb.n_el;
  for (int i = threadIdx.x; i < nv; i += blockDim.x)
    data[tb.off_values + i] = values[i];
  for (int i = threadIdx.x; i < tb.n_pwv; i += blockDim.x)
    data[tb.off_pwv + i] = axis_pwv[i];
  for (int i = threadIdx.x; i < tb.n_el; i += blockDim.x)
    data[tb.off_el + i] = axis_el[i];
  if (threadIdx.x == 0) {
    tb.p_first = axis_pwv[0];
    tb.p_inv = 1.0f / (axis_pwv[1] - axis_pwv[0]);
    tb.e_first = axis_el[0];
    tb.e_inv = 1.0f / (axis_el[1] - axis_el[0]);
  }
}

void plan_free(mrx_atm_plan* p) {
  if (!p) return;
  if (p->d_layers) (void)hipFree(p->d_layers);
  if (p->d_off) (void)hipFree(p->d_off);
  if (p->d_offpx) (void)hipFree(p->d_offpx);
  if (p->d_fast) (void)hipFree(p->d_fast);
  if (p->d_px) (void)hipFree(p->d_px);
  if (p->d_tables) (void)hipFree(p->d_tables);
  if (p->d_table_data) (void)hipFree(p->d_table_data);
  delete p;
}

}  // namespace

extern "C" {

int mrx_atm_plan_create(mrx_ctx* ctx, const mrx_layer* layers, int n_layers,
                        const mrx_band_table* tables, int n_tables, int n_t,
                        mrx_atm_plan** out) {
  MRX_ENTER(ctx);
  if (!ctx || !out) return MRX_ERR_INVALID;
  *out = nullptr;
  MRX_REQUIRE(ctx, layers != nullptr || n_layers == 0, "layers is null");
  MRX_REQUIRE(ctx, n_layers >= 0 && n_layers <= 1024, "0 <= n_layers <= 1024");  // model="3d": hundreds of layers
  MRX_REQUIRE(ctx, tables != nullptr && n_tables >= 1 && n_tables <= 1024,
              "need 1..1024 band tables");
  MRX_REQUIRE(ctx, n_t >= 1, "n_t must be positive");
  for (int l = 0; l < n_layers; ++l) {
    const mrx_layer& y = layers[l];
    MRX_REQUIRE(ctx, y.d_values && y.d_axis_e && y.d_axis_c && y.d_off_e &&
                         y.d_off_c,
                "layer has a null device pointer");
    MRX_REQUIRE(ctx, y.n_e >= 2 && y.n_c >= 2, "layer grid needs >= 2 nodes");
  }
  std::vector<mrx_table_dev> htab((size_t)n_tables);
  int floats = 0;
  for (int b = 0; b < n_tables; ++b) {
    const mrx_band_table& t = tables[b];
    MRX_REQUIRE(ctx, t.d_values && t.d_axis_pwv && t.d_axis_el,
                "band table has a null device pointer");
    MRX_REQUIRE(ctx, t.n_pwv >= 2 && t.n_el >= 2 && t.n_pwv <= 4096 &&
                         t.n_el <= 4096,
                "band table needs 2..4096 nodes per axis");
    mrx_table_dev& h = htab[(size_t)b];
    h.n_pwv = t.n_pwv;
    h.n_el = t.n_el;
    h.w_t = t.w_t;
    h.t_oob = t.t_oob;
    h.off_values = floats;
    floats += 2 * t.n_pwv * t.n_el;
    h.off_pwv = floats;
    floats += t.n_pwv;
    h.off_el = floats;
    floats += t.n_el;
    h.p_first = h.p_inv = h.e_first = h.e_inv = 0.f;
    h.cubic = t.d_cubic;
    MRX_REQUIRE(ctx, !t.d_cubic || (t.n_pwv >= 4 && t.n_el >= 4), "the cubic lookup needs >= 4 nodes per axis");
  }
  bool any_cubic = false;
  for (int b = 0; b < n_tables; ++b) any_cubic = any_cubic || tables[b].d_cubic != nullptr;
  std::vector<mrx_layer_dev> hlay((size_t)n_layers);
  for (int l = 0; l < n_layers; ++l) {
    const mrx_layer& y = layers[l];
    mrx_layer_dev& h = hlay[(size_t)l];
    h.values = y.d_values;
    h.axis_e = y.d_axis_e;
    h.axis_c = y.d_axis_c;
    h.h = y.h;
    h.r00 = y.r00; h.r10 = y.r10; h.r01 = y.r01; h.r11 = y.r11;
    h.hr00 = y.h * y.r00; h.hr10 = y.h * y.r10; h.hr01 = y.h * y.r01; h.hr11 = y.h * y.r11;
    h.e0 = y.e0; h.de = y.de; h.c0 = y.c0; h.dc = y.dc;
    h.pe_x = y.de != 0.0 ? y.h * y.r00 / y.de : 0.0;
    h.pe_y = y.de != 0.0 ? y.h * y.r10 / y.de : 0.0;
    h.pc_x = y.dc != 0.0 ? y.h * y.r01 / y.dc : 0.0;
    h.pc_y = y.dc != 0.0 ? y.h * y.r11 / y.dc : 0.0;
    h.pwv_rms = y.pwv_rms;
    h.n_e = y.n_e;
    h.n_c = y.n_c;
    h.uniform_e = h.uniform_c = 0;
    h.e_first = h.e_inv = h.e_last = h.c_first = h.c_inv = h.c_last = 0.f;
  }

  mrx_atm_plan* p = new (std::nothrow) mrx_atm_plan();
  if (!p) return mrx_fail(ctx, MRX_ERR_ALLOC, "out of host memory");
  p->n_layers = n_layers;
  p->n_tables = n_tables;
  p->n_t = n_t;
  p->table_floats = floats;
  p->any_cubic = any_cubic;
  hipError_t e = hipSuccess;
  auto ok = [&]() { return e == hipSuccess; };
  if (n_layers > 0) {
    e = hipMalloc(&p->d_layers, sizeof(mrx_layer_dev) * n_layers);
    if (ok()) e = hipMalloc(&p->d_off, sizeof(double2) * (size_t)n_layers * n_t);
    if (ok()) e = hipMalloc(&p->d_offpx, sizeof(double2) * (size_t)n_layers * n_t);
    if (ok()) e = hipMalloc(&p->d_fast, sizeof(mrx_layer_fast) * n_layers);
    if (ok()) e = hipMalloc(&p->d_px, sizeof(mrx_layer_px) * n_layers);
    if (ok())
      e = hipMemcpyAsync(p->d_layers, hlay.data(),
                         sizeof(mrx_layer_dev) * n_layers,
                         hipMemcpyHostToDevice, ctx->stream);
  }
  if (ok()) e = hipMalloc(&p->d_tables, sizeof(mrx_table_dev) * n_tables);
  if (ok()) e = hipMalloc(&p->d_table_data, sizeof(float) * (size_t)floats);
  if (ok())
    e = hipMemcpyAsync(p->d_tables, htab.data(),
                       sizeof(mrx_table_dev) * n_tables, hipMemcpyHostToDevice,
                       ctx->stream);
  if (ok() && n_layers > 0) {
    hipLaunchKernelGGL(plan_finish_layers, dim3(n_layers), dim3(256), 0,
                       ctx->stream, p->d_layers, p->d_fast, p->d_px, n_layers);
    for (int l = 0; l < n_layers; ++l)
      hipLaunchKernelGGL(plan_pack_offsets, dim3(mrx_ceil_div(n_t, 256)),
                         dim3(256), 0, ctx->stream, p->d_off, p->d_offpx, p->d_layers,
                         layers[l].d_off_e, layers[l].d_off_c, l, n_layers,
                         n_t);
  }
  if (ok())
    for (int b = 0; b < n_tables; ++b)
      hipLaunchKernelGGL(plan_pack_table, dim3(1), dim3(256), 0, ctx->stream,
                         p->d_table_data, p->d_tables, b, tables[b].d_values,
                         tables[b].d_axis_pwv, tables[b].d_axis_el);
  std::vector<mrx_layer_fast> hfast((size_t)n_layers);
  if (ok() && n_layers > 0)
    e = hipMemcpyAsync(hfast.data(), p->d_fast, sizeof(mrx_layer_fast) * n_layers, hipMemcpyDeviceToHost, ctx->stream);
  if (ok()) e = hipGetLastError();
  if (ok()) e = hipStreamSynchronize(ctx->stream);  // host vectors go away
  p->all_pixel = n_layers > 0;
  p->screen_bytes = 0;
  for (auto& f : hfast) {
    p->all_pixel = p->all_pixel && f.pixel != 0;
    p->screen_bytes += 4ull * (unsigned long long)f.n_e * (unsigned long long)f.n_c;
  }
  if (!ok()) {
    plan_free(p);
    return mrx_fail(ctx, MRX_ERR_HIP, "plan upload failed: %s",
                    hipGetErrorString(e));
  }
  *out = p;
  return MRX_OK;
}

int mrx_atm_plan_destroy(mrx_ctx* ctx, mrx_atm_plan* plan) {
  MRX_ENTER(ctx);
  if (!ctx || !plan) return MRX_ERR_INVALID;
  plan_free(plan);
  return MRX_OK;
}

int mrx_atm_plan_info(mrx_ctx* ctx, const mrx_atm_plan* plan,
                      int* uniform_axes, int* tables_in_lds) {
  MRX_ENTER(ctx);
  if (!ctx || !plan) return MRX_ERR_INVALID;
  std::vector<mrx_layer_dev> h((size_t)plan->n_layers);
  if (plan->n_layers > 0) {
    MRX_HIP(ctx, hipMemcpyAsync(h.data(), plan->d_layers,
                                sizeof(mrx_layer_dev) * plan->n_layers,
                                hipMemcpyDeviceToHost, ctx->stream));
    MRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  int n = 0;
  for (auto& l : h) n += (l.uniform_e != 0) + (l.uniform_c != 0);
  if (uniform_axes) *uniform_axes = n;
  if (tables_in_lds) *tables_in_lds = plan->table_floats <= kMaxLdsTableFloats;
  return MRX_OK;
}

int mrx_atm_sample(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az,
                   const float* d_el, int Ta, const float* d_dx,
                   const float* d_dy, const int32_t* d_band,
                   const float* d_mueller00, int D, double pwv0, double* d_pwv,
                   float* d_loading, uint32_t* d_flags) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && Ta >= 0, "negative size");
  if (D == 0 || Ta == 0) return MRX_OK;  // empty shard: nothing to do
  MRX_REQUIRE(ctx, plan != nullptr, "plan is null");
  MRX_REQUIRE(ctx, d_az && d_el && d_dx && d_dy && d_band && d_mueller00,
              "null input pointer");
  MRX_REQUIRE(ctx, d_loading && d_flags, "null output pointer");
  MRX_REQUIRE(ctx, plan->n_layers == 0 || Ta == plan->n_t,
              "Ta differs from the plan's n_t (length of the wind offsets)");
  int kt = ctx->options[MRX_OPT_SAMPLE_TIMES];
  if (kt != 1 && kt != 2 && kt != 4) kt = kTimes;
  // time steps per workgroup: as many as keep >= ~24 workgroups per CU in the grid
  // (measured on atlast_10k and act_3k: 16-32 steps per workgroup beat 64 by 5-10 %)
  int chunk = ctx->options[MRX_OPT_SAMPLE_CHUNK];
  if (chunk <= 0) {
    chunk = kMaxChunk;
    const long long want = 24LL * (ctx->n_cu > 0 ? ctx->n_cu : 256);
    while (chunk > kt && (long long)mrx_ceil_div(D, kBlock) * mrx_ceil_div(Ta, chunk) < want)
      chunk /= 2;
  }
  chunk = ((chunk < kt ? kt : chunk > kMaxChunk ? kMaxChunk : chunk) / kt) * kt;
  const int nbx = mrx_ceil_div(D, kBlock);
  const long long items = (long long)nbx * mrx_ceil_div(Ta, chunk);
  MRX_REQUIRE(ctx, items <= 0x7fffffffLL, "too many work items for one launch");
  // a resident grid of 8 workgroups per CU walking the items measured 8 % faster than one
  // workgroup per item (1.15 vs 1.25 ms on atlast_10k): tables staged once, no dispatch tail
  int per_cu = ctx->options[MRX_OPT_SAMPLE_WGS_PER_CU];
  if (per_cu <= 0) per_cu = 8;
  const long long wgs = std::min(items, (long long)per_cu * (ctx->n_cu > 0 ? ctx->n_cu : 256));
  dim3 grid((unsigned)wgs);
  const bool lds = plan->table_floats <= kMaxLdsTableFloats;
  const size_t lds_bytes = lds ? sizeof(float) * (size_t)plan->table_floats : 0;
  const bool chain = ctx->options[MRX_OPT_POINTING_CHAIN] != 0;
  // (an instance of the general kernel compiled without the literal path, for plans whose layers
  // are all uniform, measured SLOWER -- 1.27 vs 1.00 ms -- whatever the occupancy bound: kept out)
  const bool literal = ctx->options[MRX_OPT_AXIS_LITERAL] != 0 || chain;
  if (plan->all_pixel && !literal && !plan->any_cubic) {
    // the pixel-coordinate kernel: anchors of one work item (chunk steps x layers) live in LDS
    if (plan->n_layers * kt > kMaxAnchors) kt = 1;
    while (chunk > kt && chunk * plan->n_layers > kMaxAnchors) chunk /= 2;
    chunk = (chunk / kt) * kt;
    const int nby = mrx_ceil_div(Ta, chunk);
    const long long n_items = (long long)nbx * nby;
    MRX_REQUIRE(ctx, n_items <= 0x7fffffffLL, "too many work items for one launch");
    long long wgs_p = std::min(n_items, (long long)per_cu * (ctx->n_cu > 0 ? ctx->n_cu : 256));
    if (wgs_p >= 8) wgs_p &= ~7LL;  // whole groups of 8: one workgroup per XCD and turn (the kernel's item order)
    const dim3 gridp((unsigned)wgs_p);
    const size_t lds_p = 2 * sizeof(float4) * (size_t)chunk * plan->n_layers + lds_bytes;
    // a small resident grid (beside the TOD writer) takes the software-pipelined layer loop
    const bool pipe = ctx->options[MRX_OPT_SAMPLE_WGS_PER_CU] > 0 && ctx->options[MRX_OPT_SAMPLE_WGS_PER_CU] < 8;
    if (kt == 4) kt = 2;  // (no instance: four interleaved steps spill at 96 registers)
#define MRX_LAUNCH_PX(L, T, P)                                                                       \
  do {                                                                                               \
    MRX_LDS_CAP(ctx, (atm_sample_px_kernel<L, T, P>), lds_p);                                         \
    hipLaunchKernelGGL((atm_sample_px_kernel<L, T, P>), gridp, dim3(kBlock), lds_p, ctx->stream,     \
                       plan->d_fast, plan->d_px, plan->n_layers, plan->d_offpx, plan->d_tables,      \
                       plan->n_tables, plan->d_table_data, plan->table_floats, d_az, d_el, Ta, d_dx, \
                       d_dy, d_band, d_mueller00, D, pwv0, d_pwv, d_loading, d_flags, chunk, nby);   \
  } while (0)
#define MRX_LAUNCH_PX_T(L, P) do { if (kt == 1) MRX_LAUNCH_PX(L, 1, P); else MRX_LAUNCH_PX(L, 2, P); } while (0)
    if (lds) {
      if (pipe) MRX_LAUNCH_PX_T(true, true); else MRX_LAUNCH_PX_T(true, false);
    } else {
      if (pipe) MRX_LAUNCH_PX_T(false, true); else MRX_LAUNCH_PX_T(false, false);
    }
#undef MRX_LAUNCH_PX_T
#undef MRX_LAUNCH_PX
    MRX_CHECK_LAUNCH(ctx);
    return MRX_OK;
  }
#define MRX_LAUNCH_SAMPLE(L, C, T)                                             \
  hipLaunchKernelGGL((atm_sample_kernel<L, C, T>), grid, dim3(kBlock),         \
                     lds_bytes, ctx->stream, plan->d_layers, plan->d_fast, plan->n_layers, \
                     plan->d_off, plan->d_offpx, plan->d_tables, plan->n_tables, \
                     plan->d_table_data, plan->table_floats, d_az, d_el, Ta,   \
                     d_dx, d_dy, d_band, d_mueller00, D, pwv0, d_pwv,          \
                     d_loading, d_flags, literal, chunk, nbx, (int)items)
#define MRX_LAUNCH_SAMPLE_T(L, C)                                              \
  do {                                                                         \
    if (kt == 1) MRX_LAUNCH_SAMPLE(L, C, 1);                                   \
    else if (kt == 2) MRX_LAUNCH_SAMPLE(L, C, 2);                              \
    else MRX_LAUNCH_SAMPLE(L, C, 4);                                           \
  } while (0)
  if (lds) {
    if (chain) MRX_LAUNCH_SAMPLE_T(true, true); else MRX_LAUNCH_SAMPLE_T(true, false);
  } else {
    if (chain) MRX_LAUNCH_SAMPLE_T(false, true); else MRX_LAUNCH_SAMPLE_T(false, false);
  }
#undef MRX_LAUNCH_SAMPLE_T
#undef MRX_LAUNCH_SAMPLE
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

}  // extern "C"
