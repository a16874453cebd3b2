// Map sampling for gfx950: the celestial map's contribution to
// the TOD, sim/map.py:76-172.  For every detector sample the reference
//   1. takes the detector's az/el (coords/transforms.py:10-29, float32),
//   2. rotates it into the map's frame: float32 unit vector times the float64 3x3
//      of that sample, back to float32 angles (coords/coordinates.py:220-230),
//   3. turns the angles into offsets from the map centre (transforms.py:36-53, float32),
//   4. builds a sparse pointing matrix from np.digitize on the eta / xi axes with
//      bilinear (or nearest) weights in float64, times the Stokes row of the
//      detector's Mueller matrix (utils/linalg.py:9-58, map/projection.py:134-179),
//   5. multiplies P @ map by the channel's K_RJ -> pW factor 1e12 k_B Int tau exp(-opacity)
//      looked up at the sample's (zenith pwv, elevation) (band/band.py:235-255),
//      accumulates the channels in float32,
//   6. convolves the result with [0.25, 0.5, 0.25] along time (scipy reflect mode).
// Here all of it is one kernel: no [D, T] pointing, no sparse matrix, one store per
// sample.  Tile = 16 detectors x 1024 samples like the TOD writer; a thread computes 4
// consecutive raw samples per detector and trades edge values with its neighbours
// through LDS for the 3-tap kernel (tile edges: one extra evaluation by the first and
// last thread).  Steps 1-3 are three rotations of a unit vector: by default they are
// composed once per sample in float64 and applied to the detector's vector with six
// multiply-adds (the literal float32 chain, ~15 transcendental calls per sample, stays
// available through MRX_OPT_POINTING_CHAIN and agrees to float32 rounding of the angles).
// Bound by arithmetic (float64 weights and sums, as the reference's sparse product), not
// by memory.
#include <cmath>
#include <type_traits>

#include "mrx_internal.h"
#include "mrx_krj.h"  // the tile (mrx_tile.h: 16 rows x 1024 samples, 4 a thread) and TOD.to("K_RJ") on the sampler's store

namespace {

static_assert(kBlock == 256 && kTileDet == 16 && kSamplesPerThread == 4 && kTileSamples == 1024, "the map kernels are written for the TOD tile");
constexpr int kMaxStokes = 4;
constexpr float kHalfPiF = 1.57079637050628662109375f;
constexpr float kTwoPiF = 6.283185482025146484375f;

// One map axis: node i sits at first + i * step (np.linspace, map/projection.py:122-123;
// step of either sign -- eta is descending after the parity flip).
struct Axis {
  int n;
  double first, inv_step;
  float lo, hi;  // float32 bounds just beyond the axis (one pixel either side): offsets are clamped into them first
  double u_last;  // the largest double below n - 1 (axis_cell, nearest pixel)
  // The sampler's bilinear cell in float32 (axis_cell): u = x inv + c with inv = inv_hi + inv_lo (float32 head and tail of
  // 1 / step) and c = -first / step = c_int + c_frac (an integer and a fraction in [0, 1), both exact float32 numbers or
  // rounded at 6e-8).  x_min / x_max: the float32 numbers just inside the axis' ends; k_lo / k_hi: the cells 0 and n - 2
  // counted from c_int.
  float inv_hi, inv_lo, c_int, c_frac, x_min, x_max, k_lo, k_hi;
};

inline Axis make_axis(int n, double first, double step) {  // (host)
  Axis a{n, first, 1.0 / step, 0.0f, 0.0f, (double)(n - 1) * (1.0 - 1.1102230246251565e-16)};
  const double x0 = first - 1.5 * step, x1 = first + ((double)n + 0.5) * step;
  a.lo = (float)(x0 < x1 ? x0 : x1);
  a.hi = (float)(x0 < x1 ? x1 : x0);
  a.inv_hi = (float)a.inv_step;
  a.inv_lo = (float)(a.inv_step - (double)a.inv_hi);
  const double c = -first * a.inv_step, ci = std::floor(c);
  a.c_int = (float)ci;  // (|c| < 2^24: a map axis has fewer than 2^24 nodes, and the host refuses centres farther off)
  a.c_frac = (float)(c - ci);
  const double last = first + (double)(n - 1) * step, e0 = first < last ? first : last, e1 = first < last ? last : first;
  a.x_min = (float)e0;
  a.x_max = (float)e1;
  if ((double)a.x_min < e0) a.x_min = std::nextafterf(a.x_min, INFINITY);
  if ((double)a.x_max > e1) a.x_max = std::nextafterf(a.x_max, -INFINITY);
  a.k_lo = -a.c_int;
  a.k_hi = (float)(n - 2) - a.c_int;
  return a;
}

struct MapArgs {
  const float* values;  // [C][S][n_eta][n_xi]
  Axis eta, xi;
  int C, S, n_eta, n_xi;
  float cphi;           // map centre longitude (float32, as jax demotes it)
  float rot_re, rot_im; // exp(i (pi/2 - ctheta)) in complex64
  double cos_cphi, sin_cphi, cos_ctheta, sin_ctheta;  // the same centre in float64 (direct mode)
  int bilinear;
  // calibration
  const float* cal;       // [C][n_pwv][n_el] or null
  const float* cal_pwv;   // [n_pwv]
  const float* cal_el;    // [n_el]
  int n_pwv, n_el;
  const double* pwv;      // [Ta][D] coarse zenith-scaled pwv
  int Ta;
  double ta0, dta, inv_dta;
  const double* t;        // [T]
  const double* scalar;   // [C] (no atmosphere)
  // pointing
  const float* az;        // [T]
  const float* el;        // [T]
  const double* transform;  // [T][9] or null
  const float* dx;
  const float* dy;
  const double* stokes_w;  // [D][S]
  int D, T;
  float* out;
  size_t ld;
  int vec_ok;
  uint32_t map_bytes;  // of all planes (below 4 GiB): the sampler's raw buffer
  // The map's row pairs interleaved (round 6): pairs[plane][e][x] = (m[e][x], m[e + 1][x]), e = 0 .. n_eta - 2 -- the four
  // corners of a cell are then 16 contiguous bytes, ONE gather a sample and plane instead of two of 8 bytes.  The
  // sampler waits on the number of gather instructions (a what-if build without them: 7.1 -> 3.9 ms; the texture
  // addresser takes a wave's 64 addresses at the same rate whether they fetch 8 or 16 bytes), not on bytes or arithmetic.
  const float* pairs;
  uint32_t pairs_bytes;
  int coef_offset;     // of the calibration's interval table in the dynamic LDS, in floats; < 0: none (more than kCalFastChannels channels)
  int coef_cap;        // coarse steps per row the table holds
};

// TOD.to("K_RJ") on the sampler's store (mrx_map_sample_krj): the arguments of mrx_tod_to_krj for the rows of this call
struct MapKrj {
  const float* bore_el;     // [T]
  const float* dx;          // [D]
  const float* dy;          // [D]
  const int32_t* band;      // [D]
  const float* scale;       // [D] or null
  const float* cal_axis;    // [n_el]
  const float* cal_values;  // [n_bands][n_el]
  int n_el, n_bands;
  int cells_offset;         // of the cell table in the dynamic LDS, in float4
};

struct DetConst {
  float c_re, c_cr, c_im;  // sin(r)cos(p), cos(r), sin(r)sin(p)
  double w[kMaxStokes];    // Mueller[d, 0, stokes]: float64 for the binning's float64 sums; map sampling rounds them to float32
};

struct SampleConst;
__device__ __forceinline__ int sc_index(const struct MapArgs& g, const SampleConst& sc);

// one axis of utils/linalg.py:25-41: the two pixels and the weight of the upper one.
// With u = (x - first) / step the reference's np.digitize bin is floor(u) + 1 and its weight
// (x - side[b-1]) / (side[b] - side[b-1]) is the fractional part of u; at a node the two
// conventions (bin b with p = 0, bin b - 1 with p = 1) give the same interpolated value, so the
// nodes themselves need not be read.
__device__ __forceinline__ void axis_weights(const Axis& a, float x, bool bilinear, int& i0, int& i1, float& p) {
  // Round 6: the float32 split form of axis_cell_t (below) instead of a float64 position (measured equal in the binning's
  // pass A, which does not wait for its arithmetic -- kept so that the file has ONE rule for cell and weight).  The offset
  // is first clamped INTO the axis (a NaN goes to the low end): beyond the ends the reference's weights are
  // (x + inf)/inf = nan -> 0 and finite/inf = 0, i.e. the edge pixel alone -- here the first cell with p = 0 or the last
  // with p = 1, the same pixels and weights.  k: the cell, kept inside the axis; frac: the position inside it, good to
  // 1e-7 pixel (the float64 form's fraction to float32 rounding) -- so the nearest pixel flips against the float64 form
  // only within 1e-7 pixel of a pixel's edge, and a bilinear weight moves by 1e-7.
  const float xc = __builtin_amdgcn_fmed3f(x, a.x_min, a.x_max);
  const float k = __builtin_amdgcn_fmed3f(floorf(fmaf(xc, a.inv_hi, a.c_frac)), a.k_lo, a.k_hi);
  const float frac = fmaf(xc, a.inv_hi, -k) + fmaf(xc, a.inv_lo, a.c_frac);
  const int cell = (int)(k + a.c_int);  // 0 .. n - 2
  // bilinear: the cell's two nodes and the weight of the upper one (at a node the weight 0 or 1 names it either way);
  // nearest (np.digitize on the midpoints): the node on this side or the far side of the cell's middle
  const int up = frac >= 0.5f ? 1 : 0;
  i0 = bilinear ? cell : cell + up;
  i1 = bilinear ? cell + 1 : cell + up;
  p = bilinear ? __builtin_amdgcn_fmed3f(frac, 0.0f, 1.0f) : 0.0f;
}

// The same for the map SAMPLER, as one cell: i0 in [0, n - 2] and the weight p in [0, 1] of node i0 + 1 -- the value
// (1 - p) v[i0] + p v[i0 + 1] is axis_weights' everywhere: inside the axis the same floor and the same fraction of the
// same float64 u; at a node p = 0; before the first node (0, p = 0) = v[0]; from the last node on (n - 2, p = 1) =
// v[n - 1], where axis_weights says (n - 1, n - 1, 0).  What it saves the kernel -- which is bound by its instruction
// count (173 vector instructions a sample, the SIMDs 85 % busy issuing them) -- is the integer clamps and selects of two
// indices (u is clamped instead, NaN to the low end by fmax's rule; just below n - 1, so that the floor stays <= n - 2)
// and, in sample_value, the special case of the last column for its corner pairs: 26 -> 16 issue slots an axis.
template <bool kBil>
__device__ __forceinline__ void axis_cell_t(const Axis& a, float x, int& i0, float& p) {
  if (kBil) {
    // Round 6: float32, no float64 (nine of the axis' eleven instructions were float64, at half rate or less).  With
    // u = x inv + c split as in Axis: p0 = x inv_hi + c_frac is u - c_int to float32 rounding (3e-5 pixel at pixel 500: good
    // enough to name the cell, not the weight); k = floor(p0) kept inside the axis; then the weight as
    // (x inv_hi - k) + (x inv_lo + c_frac): the first bracket is ONE fused multiply-add whose exact value is below one
    // pixel, so it rounds at 6e-8 of a pixel -- the float64 form's fraction to float32 rounding.  A p0 within rounding of
    // an integer may name the neighbouring cell: the weight then comes out just below 0 or just above 1, which is the same
    // interpolated value to 1e-7 of the corner difference.  The offset is first clamped INTO the axis (a NaN goes to the
    // low end by med3's rule): before the first node (cell 0, p = 0), from the last node on (cell n - 2, p = 1).
    const float xc = __builtin_amdgcn_fmed3f(x, a.x_min, a.x_max);
    const float k = __builtin_amdgcn_fmed3f(floorf(fmaf(xc, a.inv_hi, a.c_frac)), a.k_lo, a.k_hi);
    p = fmaf(xc, a.inv_hi, -k) + fmaf(xc, a.inv_lo, a.c_frac);
    i0 = (int)(k + a.c_int);
  } else {
    // np.digitize on the midpoints: the nearest node, as the cell that holds it with weight 0 or 1
    const double u = ((double)x - a.first) * a.inv_step;
    const int i = (int)fmin(fmax(floor(u + 0.5), 0.0), (double)(a.n - 1));
    i0 = min(i, a.n - 2);
    p = i > a.n - 2 ? 1.0f : 0.0f;
  }
}

__device__ __forceinline__ void axis_cell(const Axis& a, float x, bool bilinear, int& i0, float& p) {
  if (bilinear) {  // workgroup-uniform
    asm volatile("" ::: "memory");  // keep this a branch (see axis_weights)
    axis_cell_t<true>(a, x, i0, p);
  } else {
    asm volatile("" ::: "memory");
    axis_cell_t<false>(a, x, i0, p);
  }
}

// jax RegularGridInterpolator index and weight on a float32 axis (searchsorted left)
struct RgiAxis {
  const float* g;  // nodes (LDS)
  int n;
  float first, last, inv;  // inv: (n - 1) / (last - first), the arithmetic first guess
};

__device__ __forceinline__ RgiAxis make_rgi_axis(const float* g, int n) {
  RgiAxis a{g, n, g[0], g[n - 1], 0.0f};
  a.inv = (float)(n - 1) / (a.last - a.first);
  return a;
}

__device__ __forceinline__ void rgi_axis(const RgiAxis& a, float x, int& i, float& w, bool& oob) {
  const float* g = a.g;
  int k = min(max((int)fminf(fmaxf((x - a.first) * a.inv, -1.0f), 2.0e9f), 0), a.n - 2);
  float lo = g[k], hi = g[k + 1];
  while (k < a.n - 2 && hi < x) {
    ++k;
    lo = hi;
    hi = g[k + 1];
  }
  while (k > 0 && lo >= x) {
    --k;
    hi = lo;
    lo = g[k];
  }
  i = k;
  w = (x - lo) * __builtin_amdgcn_rcpf(hi - lo);  // 1 ulp from the reference's division
  oob = !(x >= a.first && x <= a.last);
}

// What a sample contributes to every detector of the tile.
struct SampleConst {
  float ca, sa;   // cos / sin of (el_bore - pi/2), float32 as the chain computes them
  float az;       // boresight azimuth
  float G[6];     // direct mode: (dz_re, dz_im) = c @ G, c = (sin r cos p, cos r, sin r sin p); composed in float64
  int jj;         // coarse interval of the sample time and the weight within it
  double u;
  int s;          // the (clamped) sample index
};

__device__ __forceinline__ int sc_index(const MapArgs&, const SampleConst& sc) { return sc.s; }

__device__ __forceinline__ void sample_const(const MapArgs& g, int s, bool chain, SampleConst& sc) {
  s = min(max(s, 0), g.T - 1);
  const float a = __fsub_rn(g.el[s], kHalfPiF);
  sc.ca = cosf(a);
  sc.sa = sinf(a);
  sc.az = g.az[s];
  if (g.cal) {
    // zenith-scaled pwv of the sample: linear interpolation of the coarse series
    // (sim/atmosphere.py:30-37)
    const double tt = g.t[s];
    int jj = (int)floor(fmin(fmax((tt - g.ta0) * g.inv_dta, -1.0), 2.0e9));
    jj = min(max(jj, 0), g.Ta - 2);
    sc.jj = jj;
    sc.u = (tt - (g.ta0 + (double)jj * g.dta)) * g.inv_dta;
  }
  if (!chain) {
    // Steps 1-3 are rotations of the unit vector c: xyz = c @ R(az, el), v = xyz @ M(t),
    // and phi_theta_to_offsets only needs two components of v in the frame of the map
    // centre, dz = (sin dphi cos theta, cos dphi cos theta sin ctheta - sin theta cos ctheta).
    // Composed once per sample in float64.
    const double ca = (double)sc.ca, sa = (double)sc.sa;
    const double cb = cos((double)sc.az), sb = sin((double)sc.az);
    // rows of R: xyz = c_re R[0] + c_cr R[1] + c_im R[2]
    const double R[3][3] = {{ca * cb, ca * sb, sa}, {-sa * cb, -sa * sb, ca}, {-sb, cb, 0.0}};
    double P[3][2];  // v -> dz: P = M @ Cmat^T
    double M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (g.transform) {
      const double* m = g.transform + (size_t)s * 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) M[k] = m[k];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double vx = M[j * 3 + 0], vy = M[j * 3 + 1], vz = M[j * 3 + 2];
      P[j][0] = -vx * g.sin_cphi + vy * g.cos_cphi;
      P[j][1] = (vx * g.cos_cphi + vy * g.sin_cphi) * g.sin_ctheta - vz * g.cos_ctheta;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) sc.G[i * 2 + k] = (float)(R[i][0] * P[0][k] + R[i][1] * P[1][k] + R[i][2] * P[2][k]);
  }
}

// Steps 4-5 for a sample at offsets (ox, oy) from the map centre, detector elevation el_d.
struct CalLds {
  RgiAxis pwv, el;   // axes (nodes in LDS)
  const float* tab;  // [C][n_pwv][n_el] (LDS)
  const float* chan; // without the tables: [C] pW per K_RJ of a channel, 1e12 k_B x the caller's scalar, float32 (LDS)
};

// the cell of the calibration tables a (zenith pwv, elevation) pair falls in, with jax's index rule and weights
struct CalCell {
  int ip, ie;
  float wp, we;
  bool oob;
};

__device__ __forceinline__ CalCell cal_cell(const CalLds& cl, float pw, float el_d) {
  CalCell k;
  bool o1, o2;
  rgi_axis(cl.pwv, pw, k.ip, k.wp, o1);
  rgi_axis(cl.el, el_d, k.ie, k.we, o2);
  k.oob = o1 || o2;
  return k;
}

// pW per K_RJ of channel c there (band/band.py:250-255): float32 corner sum in product order, weights built as
// (1 * w_pwv) * w_el; 1e12 k_B as a weak scalar on a float32 array
__device__ __forceinline__ float cal_factor(const MapArgs& g, const CalLds& cl, const CalCell& k, int c) {
  const float* tab = cl.tab + c * g.n_pwv * g.n_el + k.ip * g.n_el + k.ie;
  const float wp0 = __fsub_rn(1.0f, k.wp), we0 = __fsub_rn(1.0f, k.we);
  float v = __fmul_rn(tab[0], __fmul_rn(wp0, we0));
  v = __fadd_rn(v, __fmul_rn(tab[1], __fmul_rn(wp0, k.we)));
  v = __fadd_rn(v, __fmul_rn(tab[g.n_el], __fmul_rn(k.wp, we0)));
  v = __fadd_rn(v, __fmul_rn(tab[g.n_el + 1], __fmul_rn(k.wp, k.we)));
  if (k.oob) v = __builtin_nanf("");
  return __fmul_rn(1.380649e-11f, v);
}

constexpr int kCalFastChannels = 4;  // channels whose factors the interval form below tabulates
// The interval form of the per-sample calibration (round 6): per row and coarse step of the pwv series that the tile
// meets, the factor along the step as a parabola in the step's own coordinate (three floats), in LDS as
// [channel][row][steps][3] (row pitch 3 * cap + 1 floats).  kMaxSteps: the most steps a tile may meet for the form to
// apply (the knots' boresight table is sized for it); the launcher sizes the coefficients by mrx_map_cal.steps_per_tile.
constexpr int kMaxSteps = 64, kDefaultSteps = 33;

// kCal with fixed_cal != nullptr: the per-channel factors of this sample are given (interpolated by the caller)
template <bool kCal, int kS>
__device__ __forceinline__ float sample_value(const MapArgs& g, const CalLds& cl, const Axis& ax_eta, const Axis& ax_xi,
                                              const DetConst& dc, int d, const SampleConst& sc, float ox,
                                              float oy, float el_d, double y0, double y1, const float* fixed_cal = nullptr) {
  int e0, x0;
  float pe, px;
  axis_cell(ax_eta, oy, g.bilinear, e0, pe);
  axis_cell(ax_xi, ox, g.bilinear, x0, px);
  // float32 weights and sums (round 3): the reference's float64 sparse product P @ map is rounded to
  // float32 per channel anyway (map.py:155); a float32 evaluation is within 2e-7 of it
  const float qe = 1.0f - pe;
  CalCell cell{};
  if (kCal && !fixed_cal) cell = cal_cell(cl, (float)fma(sc.u, y1 - y0, y0), el_d);  // (pwv demoted to float32 by the jax interpolator)
  const int plane = g.n_eta * g.n_xi;  // < 2^29: checked by the host (byte offsets in 32 bits)
  // The two corners of a row as ONE 8-byte load (x0, x0 + 1), rows e0 and e0 + 1 (axis_cell: a sample in the last
  // column or row, or beyond, sits in the last cell with upper weight 1).  Round 6: BUFFER loads -- a plane is a raw
  // buffer whose descriptor lives in scalar registers, the cell's byte offset (one 24-bit multiply-add and a shift)
  // is the only vector operand and the second row is the same offset with the row pitch as the scalar offset: the
  // plain loads added the plane's base to a 64-bit vector address twice per plane and sample.
  const float w0a = qe * (1.0f - px), w0b = qe * px, w1a = pe * (1.0f - px), w1b = pe * px;
  const int o0 = (int)(__umul24((unsigned)e0, (unsigned)g.n_xi) + (unsigned)x0) << 2;  // (e0 < n_eta, n_xi < 2^24)
  const int pitch4 = g.n_xi << 2;
  typedef float pair4 __attribute__((ext_vector_type(2)));
  // ONE descriptor for the whole map (four scalar registers for the kernel's life; a descriptor per plane had the
  // compiler keep them all and spill scalars by the hundred), the plane as the load's scalar offset: the host refuses
  // maps of 4 GiB or more.
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.values, 0, (int)g.map_bytes, 0x00020000);
  float acc = 0.0f;
  auto channel = [&](int c, float pw_per_k, bool first_channel) {
    float val = 0.0f;
#pragma unroll
    for (int k = 0; k < kS; ++k) {
      const int soff = (c * kS + k) * (plane << 2);
      const pair4 r0 = __builtin_bit_cast(pair4, __builtin_amdgcn_raw_buffer_load_b64(rs, o0, soff, 0));
      const pair4 r1 = __builtin_bit_cast(pair4, __builtin_amdgcn_raw_buffer_load_b64(rs, o0, soff + pitch4, 0));
      const float v = fmaf(w0a, r0.x, fmaf(w0b, r0.y, fmaf(w1a, r1.x, w1b * r1.y)));
      val = k == 0 ? (float)dc.w[k] * v : fmaf((float)dc.w[k], v, val);  // (0 + x: the sum's first term as it is)
    }
    acc = first_channel ? pw_per_k * val : fmaf(pw_per_k, val, acc);  // float32 accumulator (map.py:155)
  };
  if (kCal && fixed_cal) {
    // (the caller's factors live in registers: a loop over c would pick them with a chain of selects)
#pragma unroll
    for (int c = 0; c < kCalFastChannels; ++c)
      if (c < g.C) channel(c, fixed_cal[c], c == 0);  // (uniform)
  } else {
    for (int c = 0; c < g.C; ++c) channel(c, kCal ? cal_factor(g, cl, cell, c) : cl.chan[c], false);
  }
  return acc;
}

// asin on [-1, 1] by Abramowitz & Stegun 4.4.46 (|error| <= 2e-8, below the float32 spacing of
// an elevation): pi/2 - sqrt(1 - |x|) P7(|x|).  Only the composed path's table lookup uses it.
__device__ __forceinline__ float asin_poly(float x) {
  const float a = fabsf(x);
  float p = -0.0012624911f;
  p = fmaf(p, a, 0.0066700901f);
  p = fmaf(p, a, -0.0170881256f);
  p = fmaf(p, a, 0.0308918810f);
  p = fmaf(p, a, -0.0501743046f);
  p = fmaf(p, a, 0.0889789874f);
  p = fmaf(p, a, -0.2145988016f);
  p = fmaf(p, a, 1.5707963050f);
  const float r = 1.57079632679f - sqrtf(fmaxf(1.0f - a, 0.0f)) * p;
  return copysignf(r, x);
}

// raw (unconvolved) map loading of one detector at one sample, float32.
// kChain: steps 1-3 literally as the reference's float32 chain; otherwise the composed
// rotation of sample_const (same angles to float32 rounding, ~10x less arithmetic).
// Steps 1-3: offsets (ox, oy) of one detector sample from the map centre, and the
// detector's elevation when the calibration needs it.
template <bool kChain, bool kNeedEl>
__device__ __forceinline__ void sample_offsets(const MapArgs& g, const DetConst& dc, const SampleConst& sc,
                                               float& ox, float& oy, float& el_d) {
  const float im = __fadd_rn(__fmul_rn(dc.c_re, sc.sa), __fmul_rn(dc.c_cr, sc.ca));
  el_d = 0.0f;
  if (kChain) {
    // 1. detector az/el (transforms.py:10-29)
    const float re = __fsub_rn(__fmul_rn(dc.c_re, sc.ca), __fmul_rn(dc.c_cr, sc.sa));
    const float az_d = __fadd_rn(atan2f(dc.c_im, re), sc.az);
    el_d = asinf(im);
    // 2. frame rotation (coordinates.py:220-230)
    float phi = az_d, theta = el_d;
    if (g.transform) {
      const float ce = cosf(el_d);
      const double x = (double)__fmul_rn(cosf(az_d), ce), y = (double)__fmul_rn(sinf(az_d), ce), z = (double)sinf(el_d);
      const double* M = g.transform + (size_t)sc_index(g, sc) * 9;
      const float vx = (float)(x * M[0] + y * M[3] + z * M[6]);
      const float vy = (float)(x * M[1] + y * M[4] + z * M[7]);
      const float vz = (float)(x * M[2] + y * M[5] + z * M[8]);
      float ph = fmodf(atan2f(vy, vx), kTwoPiF);
      if (ph < 0.0f) ph = __fadd_rn(ph, kTwoPiF);
      phi = ph;
      const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz)));
      theta = asinf(__fdiv_rn(vz, nrm));
    }
    // 3. offsets from the map centre (transforms.py:36-53)
    const float dphi = __fsub_rn(phi, g.cphi);
    const float ct = cosf(theta), st = sinf(theta);
    const float pr = __fmul_rn(cosf(dphi), ct);
    const float proj_re = __fsub_rn(__fmul_rn(pr, g.rot_re), __fmul_rn(st, g.rot_im));
    const float dz_re = __fmul_rn(sinf(dphi), ct), dz_im = proj_re;
    const float r = sqrtf(__fadd_rn(__fmul_rn(dz_re, dz_re), __fmul_rn(dz_im, dz_im)));
    const float f = __fdiv_rn(asinf(r), r > 0.0f ? r : 1.0f);
    ox = -__fmul_rn(dz_re, f);
    oy = -__fmul_rn(dz_im, f);
  } else {
    // float32 (round 3): the offsets are float32 numbers in the reference (transforms.py:36-53); with the
    // composed rotation rounded to float32 every term below is good to 6e-8 of ~1e-2 rad, i.e. to the
    // spacing of the float32 result itself
    const float dz_re = fmaf(dc.c_re, sc.G[0], fmaf(dc.c_cr, sc.G[2], dc.c_im * sc.G[4]));
    const float dz_im = fmaf(dc.c_re, sc.G[1], fmaf(dc.c_cr, sc.G[3], dc.c_im * sc.G[5]));
    const float r2 = fmaf(dz_re, dz_re, dz_im * dz_im);
    // asin(r)/r: the series to r^8 below 0.1 rad (error < 3e-10), the function beyond
    float f = fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, 35.0f / 1152.0f, 15.0f / 336.0f), 3.0f / 40.0f), 1.0f / 6.0f), 1.0f);
    if (__builtin_amdgcn_ballot_w64(r2 >= 0.01f) != 0) {
      if (r2 >= 0.01f) {
        const float r = sqrtf(r2);
        f = asinf(fminf(r, 1.0f)) / r;
      }
    }
    ox = -dz_re * f;
    oy = -dz_im * f;
    if (kNeedEl) el_d = asin_poly(im);
  }
}

// the same with the sample's per-channel calibration factors given: no detector elevation, no pwv
template <bool kChain, int kS>
__device__ __forceinline__ float raw_sample_fixed(const MapArgs& g, const CalLds& cl, const Axis& ax_eta, const Axis& ax_xi,
                                                  const DetConst& dc, int d, const SampleConst& sc, const float* cal) {
  float ox, oy, el_d;
  sample_offsets<kChain, false>(g, dc, sc, ox, oy, el_d);
  return sample_value<true, kS>(g, cl, ax_eta, ax_xi, dc, d, sc, ox, oy, el_d, 0.0, 0.0, cal);
}

// (jj0, y0, y1): the coarse pwv pair the caller already holds for this detector (the samples
// of a thread nearly always share their coarse interval); reloaded when the sample's differs.
template <bool kChain, bool kCal, int kS>
__device__ __forceinline__ float raw_sample(const MapArgs& g, const CalLds& cl, const Axis& ax_eta, const Axis& ax_xi,
                                            const DetConst& dc, int d, const SampleConst& sc, int jj0, double y0,
                                            double y1) {
  float ox, oy, el_d;
  sample_offsets<kChain, kCal>(g, dc, sc, ox, oy, el_d);
  if (kCal && sc.jj != jj0) {
    y0 = g.pwv[(size_t)sc.jj * g.D + d];
    y1 = g.pwv[(size_t)(sc.jj + 1) * g.D + d];
  }
  return sample_value<kCal, kS>(g, cl, ax_eta, ax_xi, dc, d, sc, ox, oy, el_d, y0, y1);
}

// Round 6: the kN samples of one thread and one detector row TOGETHER (kN = 4, or 5 with the thread's halo sample), for the
// composed rotation with per-channel factors that need no lookup per sample (the caller's scalars, or the stretch form's
// chord).  One sample at a time the kernel was a chain of latencies -- three LDS reads, arithmetic, two gathers from L2,
// arithmetic, per sample and plane, every wave waiting 59 % of its cycles at 38 % of the vector issue rate
// (gpurun_out/r6a_pmc_*: 5.6e9 wave-instructions in 2.8e7 cycles of 1024 SIMDs at two cycles each) -- because the sample's
// code sat behind branches (bilinear or nearest, the far-offset form of asin(r)/r, a guard per channel) that end a basic
// block, and the compiler schedules inside a block.  Here the branches are out of the samples' way: kBil is a template
// parameter, the far-offset form is one vote per row, the channel loop is outside the samples -- so a thread's 2 kN kS
// gathers of a channel are in flight together, behind 3 kN LDS reads issued together.
//   record(slot): the sample's six matrix entries (LDS); slot[q]: the samples' slots; pw(c, f): fills f[q] with pW per K_RJ
//   of channel c at sample q; kUnrollC: at most kCalFastChannels channels, unrolled under a guard; else a loop.
template <bool kBil, int kS, int kN, bool kUnrollC, typename RecordFn, typename Record4Fn, typename PwFn>
__device__ __forceinline__ void row_samples(const MapArgs& g, const Axis& ax_eta, const Axis& ax_xi, const DetConst& dc,
                                            const int (&slot)[kN], RecordFn record, Record4Fn record4, PwFn pw, float (&out)[kN]) {
  float ox[kN], oy[kN], r2[kN];
  bool far = false;
  float G[6][kN];  // slots slot[0] .. slot[0] + 3 are the thread's own four samples: one 16-byte read per entry
  {
    float G4[6][4];
    record4(slot[0], G4);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
      for (int q = 0; q < 4; ++q) G[k][q] = G4[k][q];
    }
    if (kN > 4) {
      const SampleConst h = record(slot[kN - 1]);
#pragma unroll
      for (int k = 0; k < 6; ++k) G[k][kN - 1] = h.G[k];
    }
  }
#pragma unroll
  for (int q = 0; q < kN; ++q) {
    // (sample_offsets' composed form: float32, see there)
    const float dz_re = fmaf(dc.c_re, G[0][q], fmaf(dc.c_cr, G[2][q], dc.c_im * G[4][q]));
    const float dz_im = fmaf(dc.c_re, G[1][q], fmaf(dc.c_cr, G[3][q], dc.c_im * G[5][q]));
    r2[q] = fmaf(dz_re, dz_re, dz_im * dz_im);
    const float f = fmaf(r2[q], fmaf(r2[q], fmaf(r2[q], fmaf(r2[q], 35.0f / 1152.0f, 15.0f / 336.0f), 3.0f / 40.0f), 1.0f / 6.0f), 1.0f);
    far |= r2[q] >= 0.01f;
    ox[q] = -dz_re * f;
    oy[q] = -dz_im * f;
  }
  if (__builtin_amdgcn_ballot_w64(far) != 0) {  // beyond 0.1 rad of the map's centre: asin(r)/r itself instead of its series
#pragma unroll 1
    for (int q = 0; q < kN; ++q) {
      const float x = r2[q];
      if (x >= 0.01f) {
        const float series = fmaf(x, fmaf(x, fmaf(x, fmaf(x, 35.0f / 1152.0f, 15.0f / 336.0f), 3.0f / 40.0f), 1.0f / 6.0f), 1.0f);
        const float r = sqrtf(x);
        const float ratio = (asinf(fminf(r, 1.0f)) / r) / series;
#pragma unroll
        for (int k = 0; k < kN; ++k) {  // (registers, not an indexed array)
          ox[k] = k == q ? ox[k] * ratio : ox[k];
          oy[k] = k == q ? oy[k] * ratio : oy[k];
        }
      }
    }
  }
  int o0[kN];
  float w0a[kN], w0b[kN], w1a[kN], w1b[kN];
#pragma unroll
  for (int q = 0; q < kN; ++q) {
    int e0, x0;
    float pe, px;
    axis_cell_t<kBil>(ax_eta, oy[q], e0, pe);
    axis_cell_t<kBil>(ax_xi, ox[q], x0, px);
    const float qe = 1.0f - pe, qx = 1.0f - px;
    w0a[q] = qe * qx; w0b[q] = qe * px; w1a[q] = pe * qx; w1b[q] = pe * px;
    o0[q] = (int)(__umul24((unsigned)e0, (unsigned)g.n_xi) + (unsigned)x0) << 3;  // (e0 < n_eta, n_xi < 2^24): the cell's 16 bytes in its plane of pairs
  }
  typedef float quad4 __attribute__((ext_vector_type(4)));
  const int plane8 = ((g.n_eta - 1) * g.n_xi) << 3;  // a plane of row pairs, bytes
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g.pairs, 0, (int)g.pairs_bytes, 0x00020000);
#pragma unroll
  for (int q = 0; q < kN; ++q) out[q] = 0.0f;
  float wf[kS];  // the Stokes weights as the float32 numbers the sums take
#pragma unroll
  for (int k = 0; k < kS; ++k) wf[k] = (float)dc.w[k];
  auto channel = [&](int c, bool first_channel) {
    float val[kN];
#pragma unroll
    for (int k = 0; k < kS; ++k) {  // (a plane's kN gathers in flight together; all kS planes' at once would not fit the registers)
      const int soff = (c * kS + k) * plane8;
      quad4 cn[kN];  // (m[e0][x0], m[e0 + 1][x0], m[e0][x0 + 1], m[e0 + 1][x0 + 1])
#pragma unroll
      for (int q = 0; q < kN; ++q) cn[q] = __builtin_bit_cast(quad4, __builtin_amdgcn_raw_buffer_load_b128(rs, o0[q], soff, 0));
#pragma unroll
      for (int q = 0; q < kN; ++q) {
        const float v = fmaf(w0a[q], cn[q].x, fmaf(w0b[q], cn[q].z, fmaf(w1a[q], cn[q].y, w1b[q] * cn[q].w)));
        val[q] = k == 0 ? wf[k] * v : fmaf(wf[k], v, val[q]);
      }
    }
    float f[kN];
    pw(c, f);
#pragma unroll
    for (int q = 0; q < kN; ++q)
      out[q] = first_channel ? f[q] * val[q] : fmaf(f[q], val[q], out[q]);  // float32 accumulator (map.py:155)
  };
  if (kUnrollC) {
#pragma unroll
    for (int c = 0; c < kCalFastChannels; ++c)
      if (c < g.C) channel(c, c == 0);  // (uniform)
  } else {
    for (int c = 0; c < g.C; ++c) channel(c, false);
  }
}

__device__ __forceinline__ DetConst make_det_const(const MapArgs& g, int d) {
  // transforms.py:14-23: r = |offset|, p = atan2(-dx, -dy); c = (sin r cos p, cos r, sin r sin p) = (-dy s, cos r, -dx s) with
  // s = sin(r) / r.  Round 6: the series of mrx_krj.h for a focal-plane offset (exact to a float32 rounding below 0.3 rad)
  // instead of a square root, an arctangent and four sines and cosines -- once per row and tile, but on sixteen lanes of
  // one wave while the workgroup waits at the barrier behind it.
  const float dx = g.dx[d], dy = g.dy[d];
  const float r2 = fmaf(dx, dx, dy * dy);
  DetConst dc;
  if (r2 < 0.09f) {
    const float sc = sinc_small(r2);
    dc.c_re = -dy * sc;
    dc.c_cr = cos_small(r2);
    dc.c_im = -dx * sc;
  } else {
    const float r = sqrtf(r2);
    const float p = atan2f(-dx, -dy);
    const float sr = sinf(r);
    dc.c_re = __fmul_rn(sr, cosf(p));
    dc.c_cr = cosf(r);
    dc.c_im = __fmul_rn(sr, sinf(p));
  }
  for (int k = 0; k < kMaxStokes; ++k) dc.w[k] = k < g.S ? g.stokes_w[(size_t)d * g.S + k] : 0.0;
  return dc;
}

// ---- binning: the transpose of the same pointing matrix (mappers/bin_mapper.py:84-120) ----
struct BinArgs {
  const float* tod;     // [D][ld_tod] signal
  size_t ld_tod;
  const float* weight;  // [D][ld_w] or null (ones)
  size_t ld_w;
  const int32_t* channel;  // [D] map channel of each detector, or null (0)
  double* sum;          // [S][C][n_eta][n_xi]
  double* wgt;
};

// One run of consecutive samples that share their pixels: A[c] = sum W D w_c, B[c] = sum W w_c
// over the 4 corners c (bilinear weights w_c >= 0); flushed with one atomic per corner, Stokes
// plane and product.
struct BinRun {
  int e0, e1, x0, x1;
  double A[4], B[4];
};

__device__ __forceinline__ void flush_run(const MapArgs& g, const BinArgs& b, const DetConst& dc, int chan, const BinRun& r) {
  const int plane = g.n_eta * g.n_xi;
  const int o[4] = {r.e0 * g.n_xi + r.x0, r.e1 * g.n_xi + r.x0, r.e0 * g.n_xi + r.x1, r.e1 * g.n_xi + r.x1};
  const int corners = g.bilinear ? 4 : 1;
  for (int k = 0; k < g.S; ++k) {
    const size_t base = ((size_t)k * g.C + chan) * plane;
    const double m = dc.w[k];
    for (int c = 0; c < corners; ++c) {
      if (r.B[c] == 0.0) continue;  // zero weight: nothing to add (np.abs(P) entries that are 0)
      atomicAdd(b.sum + base + o[c], m * r.A[c]);
      atomicAdd(b.wgt + base + o[c], fabs(m) * r.B[c]);
    }
  }
}

template <bool kChain>
__global__ __launch_bounds__(kBlock) void bin_map_kernel(MapArgs g, BinArgs b) {
  __shared__ DetConst dets[kTileDet];
  const int d0 = blockIdx.y * kTileDet;
  const int sb = blockIdx.x * kTileSamples + threadIdx.x * kSamplesPerThread;
  const int nd = min(kTileDet, g.D - d0);
  if ((int)threadIdx.x < nd) dets[threadIdx.x] = make_det_const(g, d0 + threadIdx.x);
  const Axis ax_eta = g.eta, ax_xi = g.xi;
  SampleConst sc[kSamplesPerThread];
#pragma unroll
  for (int q = 0; q < kSamplesPerThread; ++q) {
    sample_const(g, sb + q, kChain, sc[q]);
    sc[q].s = min(max(sb + q, 0), g.T - 1);
  }
  __syncthreads();
  if (sb >= g.T) return;
  for (int dl = 0; dl < nd; ++dl) {
    const DetConst dc = dets[dl];
    const int d = d0 + dl;
    const int chan = b.channel ? min(max(b.channel[d], 0), g.C - 1) : 0;
    BinRun run;
    bool open = false;
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) {
      if (sb + q >= g.T) break;
      float ox, oy, el_d;
      sample_offsets<kChain, false>(g, dc, sc[q], ox, oy, el_d);
      int e0, e1, x0, x1;
      float pef, pxf;
      axis_weights(ax_eta, oy, g.bilinear, e0, e1, pef);
      axis_weights(ax_xi, ox, g.bilinear, x0, x1, pxf);
      const double pe = (double)pef, px = (double)pxf;
      const double W = b.weight ? (double)b.weight[(size_t)d * b.ld_w + sb + q] : 1.0;
      const double WD = W * (double)b.tod[(size_t)d * b.ld_tod + sb + q];
      const double w[4] = {(1.0 - pe) * (1.0 - px), pe * (1.0 - px), (1.0 - pe) * px, pe * px};
      if (open && (e0 != run.e0 || x0 != run.x0 || e1 != run.e1 || x1 != run.x1)) {
        flush_run(g, b, dc, chan, run);
        open = false;
      }
      if (!open) {
        run.e0 = e0; run.e1 = e1; run.x0 = x0; run.x1 = x1;
#pragma unroll
        for (int c = 0; c < 4; ++c) run.A[c] = run.B[c] = 0.0;
        open = true;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        run.A[c] = fma(WD, w[c], run.A[c]);
        run.B[c] = fma(W, w[c], run.B[c]);
      }
    }
    if (open) flush_run(g, b, dc, chan, run);
  }
}

// ---- bucketed binning: no global atomics on scattered addresses ---------------------------
// Scattered float atomics execute at the memory side at a tenth of the contiguous rate
// (MI355X_MICROARCH.md, Global float atomics), and the focal plane is sparser than the map, so
// privatising tiles in LDS merges next to nothing.  Instead the samples are routed
// to the map: pass A computes every sample's pixel(s) once, sorts its tile's contributions by
// map region (64 x 32 pixels of one plane) with an LDS counting sort and writes them, region by
// region, into the tile's slot of the work buffer plus one word per (region, tile) that says
// where; pass B gives every region to a few workgroups that read the region's segments,
// accumulate in LDS (float64 ds_add) and add the finished block to the map: atomics again, but
// on whole rows of 64 pixels.  The result equals mrx_bin_map's to float64 rounding (the order of
// the sums differs).  Nearest pixel: a tile is 16 detectors x 1024 samples, one contribution per
// sample.  Bilinear: 8 detectors x 256 samples, up to four contributions per sample (corners of
// zero weight are dropped); the float32 offsets of the tile's samples stay in LDS between the
// two sweeps so that the weights are formed once more without the pointing.
constexpr int kBinBx = 64, kBinBy = 32;
constexpr int kBinRegionPx = kBinBx * kBinBy;  // 2048 pixels: 11 bits
constexpr int kBinMaxRegions = 2048;
constexpr uint32_t kBinNone = 0xffffffffu;

template <bool kBil>
struct BinTile {
#ifndef MRX_BIN_DET
#define MRX_BIN_DET 16
#define MRX_BIN_SPT 4
#endif
  static constexpr int kDet = kBil ? 8 : MRX_BIN_DET;     // detectors per tile (< 32: five bits of an entry)
  static constexpr int kSpt = kBil ? 1 : MRX_BIN_SPT;     // samples per thread
  static constexpr int kCorners = kBil ? 4 : 1;
  static constexpr int kSamples = kBlock * kSpt;
  static constexpr int kEntries = kDet * kSamples * kCorners;  // 16 384 / 8 192 contributions per tile at most
};

// A routed contribution.  Bilinear: pixel, signal, sample weight x corner weight in float64 (16 bytes).
// Nearest pixel (the mapper's default; corner weight 1): the sample weight as the float32 it is (12 bytes), or
// nothing at all when the caller passed no weights (8 bytes: pass A writes and pass B reads half the bytes).
template <int kBytes>
struct BinEntryT;
template <>
struct BinEntryT<16> {
  uint32_t local;  // pixel inside the region, (eta & 31) << 6 | (xi & 63), | detector within the tile << 11
  float d;         // signal
  double ww;       // sample weight x corner weight
  __device__ __forceinline__ double weight() const { return ww; }
  __device__ __forceinline__ void set_weight(double w) { ww = w; }
};
template <>
struct BinEntryT<12> {
  uint32_t local;
  float d;
  float w;
  __device__ __forceinline__ double weight() const { return (double)w; }
  __device__ __forceinline__ void set_weight(double x) { w = (float)x; }
};
template <>
struct BinEntryT<8> {
  uint32_t local;
  float d;
  __device__ __forceinline__ double weight() const { return 1.0; }
  __device__ __forceinline__ void set_weight(double) {}
};
static_assert(sizeof(BinEntryT<16>) == 16 && sizeof(BinEntryT<12>) == 12 && sizeof(BinEntryT<8>) == 8, "entry sizes");
constexpr int bin_entry_bytes(bool bilinear, bool weights) { return bilinear ? 16 : weights ? 12 : 8; }

struct BucketArgs {
  int nbx, nby, R;     // regions per row / per column of a plane, in all (channels x nby x nbx)
  int s0, s1;          // samples [s0, s1) of this time chunk
  int tiles_x;         // sample tiles of the chunk; tile = blockIdx.y * tiles_x + blockIdx.x
  int n_tiles;
  int tile_det, tile_entries;  // BinTile<>::kDet, kEntries of the form in use
  uint32_t* tab;       // [R][n_tiles]: (first entry of the region in the tile's slot) << 16 | count
  uint32_t* totals;    // [R]: contributions per region in this chunk (pass A adds, bin_order_kernel reads)
  const int* order;    // [R]: regions by falling total (pass B takes the heavy ones first: its tail is then made of light ones)
  void* entries;       // [n_tiles][tile_entries] BinEntryT<entry_bytes>
};

// the contributions of one sample: pixel (region << 11 | local) and weight per corner
template <bool kBil>
__device__ __forceinline__ void bin_corners(const MapArgs& g, const BucketArgs& k, const Axis& ax_eta, const Axis& ax_xi,
                                            float ox, float oy, int chan, uint32_t (&word)[BinTile<kBil>::kCorners],
                                            double (&wc)[BinTile<kBil>::kCorners]) {
  int e0, e1, x0, x1;
  float pef, pxf;
  axis_weights(ax_eta, oy, kBil, e0, e1, pef);
  axis_weights(ax_xi, ox, kBil, x0, x1, pxf);
  const double pe = (double)pef, px = (double)pxf;  // the products below in float64, as bin_map_kernel forms them
  auto pixel = [&](int e, int x) {
    const uint32_t r = (uint32_t)((chan * k.nby + (e >> 5)) * k.nbx + (x >> 6));
    return (r << 11) | (uint32_t)(((e & 31) << 6) | (x & 63));
  };
  if constexpr (kBil) {
    // the order and the weights of bin_map_kernel: (e0,x0), (e1,x0), (e0,x1), (e1,x1)
    word[0] = pixel(e0, x0); wc[0] = (1.0 - pe) * (1.0 - px);
    word[1] = pixel(e1, x0); wc[1] = pe * (1.0 - px);
    word[2] = pixel(e0, x1); wc[2] = (1.0 - pe) * px;
    word[3] = pixel(e1, x1); wc[3] = pe * px;
  } else {
    word[0] = pixel(e0, x0);
    wc[0] = 1.0;
  }
}

template <bool kChain, bool kBil, bool kW>
// (168 registers instead of 174: measured 21.3 -> 19.7 ms)
#ifndef MRX_BIN_WAVES
#define MRX_BIN_WAVES 3
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MRX_BIN_WAVES, MRX_BIN_WAVES))) void bin_bucket_kernel(MapArgs g, BinArgs b, BucketArgs k) {
  using Tile = BinTile<kBil>;
  using Entry = BinEntryT<bin_entry_bytes(kBil, kW)>;
  constexpr int kDet = Tile::kDet, kSpt = Tile::kSpt, kCorners = Tile::kCorners;
  __shared__ DetConst dets[kDet];
  __shared__ uint32_t part[kBlock];
  extern __shared__ uint32_t bucket_lds[];
  // A contribution's sort word, region << 11 | pixel-in-region, is 22 bits: kept as its low 16 bits and its high byte
  // in two arrays (0xff in the high byte: no contribution), 48 KB a tile instead of 64 -- with the histogram doubling
  // as the cursors that lets THREE workgroups share a CU's LDS instead of two (the kernel waits on its pointing
  // arithmetic and its LDS atomics: occupancy is what it lacks; pass A 16.9 -> 13.1 ms onto 1024^2).  A thread's four
  // words of one detector are neighbours in both arrays: one 8-byte and one 4-byte LDS access each way.
  constexpr int kGroup = kSpt * kCorners;  // sort words per thread and detector: 4, or 2
  static_assert(kGroup == 4 || kGroup == 2, "two or four sort words per thread and detector");
  using Lo = std::conditional_t<kGroup == 4, uint2, uint32_t>;     // kGroup x uint16
  using Hi = std::conditional_t<kGroup == 4, uint32_t, uint16_t>;  // kGroup x uint8
  Lo* lo16 = reinterpret_cast<Lo*>(bucket_lds);                   // [kEntries / kGroup], by (detector, thread)
  Hi* hi8 = reinterpret_cast<Hi*>(bucket_lds + Tile::kEntries / 2);  // [kEntries / kGroup]
  uint32_t* hist = bucket_lds + Tile::kEntries / 2 + Tile::kEntries / 4;  // [R]: counts, then (after the scan) cursors
  float2* oxy = reinterpret_cast<float2*>(hist + k.R);  // bilinear: [kDet][kSamples] offsets of the samples
  const int d0 = blockIdx.y * kDet;
  const int sb = k.s0 + blockIdx.x * Tile::kSamples + threadIdx.x * kSpt;
  const int nd = min(kDet, g.D - d0);
  const int tile = blockIdx.y * k.tiles_x + blockIdx.x;
  if ((int)threadIdx.x < nd) dets[threadIdx.x] = make_det_const(g, d0 + threadIdx.x);
  for (int i = threadIdx.x; i < k.R; i += kBlock) hist[i] = 0u;
  const Axis ax_eta = g.eta, ax_xi = g.xi;
  SampleConst sc[kSpt];
#pragma unroll
  for (int q = 0; q < kSpt; ++q) {
    sample_const(g, sb + q, kChain, sc[q]);
    sc[q].s = min(max(sb + q, 0), g.T - 1);
  }
  __syncthreads();
  // sweep 1: the pixels of every sample, the tile's histogram over regions
  for (int dl = 0; dl < nd; ++dl) {
    const DetConst dc = dets[dl];
    const int d = d0 + dl;
    const int chan = b.channel ? min(max(b.channel[d], 0), g.C - 1) : 0;
    uint32_t grp[kGroup];
#pragma unroll
    for (int q = 0; q < kSpt; ++q) {
      uint32_t word[kCorners];
      double wc[kCorners];
#pragma unroll
      for (int c = 0; c < kCorners; ++c) word[c] = kBinNone;
      if (sb + q < k.s1) {
        float ox, oy, el_d;
        sample_offsets<kChain, false>(g, dc, sc[q], ox, oy, el_d);
        if (kBil) oxy[(dl * kBlock + threadIdx.x) * kSpt + q] = make_float2(ox, oy);
        bin_corners<kBil>(g, k, ax_eta, ax_xi, ox, oy, chan, word, wc);
#pragma unroll
        for (int c = 0; c < kCorners; ++c) {
          if (wc[c] == 0.0) word[c] = kBinNone;  // a corner of zero weight adds nothing (np.abs(P) entries that are 0)
          else atomicAdd(&hist[word[c] >> 11], 1u);
        }
      }
#pragma unroll
      for (int c = 0; c < kCorners; ++c) grp[q * kCorners + c] = word[c];
    }
    // (kBinNone >> 16 = 0xffff: its high byte reads 0xff, which no region's does -- R <= 2048 leaves 6 bits there)
    if constexpr (kGroup == 4) {
      lo16[dl * kBlock + threadIdx.x] = make_uint2((grp[0] & 0xffffu) | (grp[1] << 16), (grp[2] & 0xffffu) | (grp[3] << 16));
      hi8[dl * kBlock + threadIdx.x] = ((grp[0] >> 16) & 0xffu) | (((grp[1] >> 16) & 0xffu) << 8) | (((grp[2] >> 16) & 0xffu) << 16) |
                                       (((grp[3] >> 16) & 0xffu) << 24);
    } else {
      lo16[dl * kBlock + threadIdx.x] = (grp[0] & 0xffffu) | (grp[1] << 16);
      hi8[dl * kBlock + threadIdx.x] = (uint16_t)(((grp[0] >> 16) & 0xffu) | (((grp[1] >> 16) & 0xffu) << 8));
    }
  }
  __syncthreads();
  // exclusive scan of the histogram: where each region's contributions start in the tile's slot
  const int per = (k.R + kBlock - 1) / kBlock;
  const int r_lo = threadIdx.x * per, r_hi = min(r_lo + per, k.R);
  uint32_t mine = 0;
  for (int r = r_lo; r < r_hi; ++r) mine += hist[r];
  part[threadIdx.x] = mine;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {
    const uint32_t add = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - mine;
  for (int r = r_lo; r < r_hi; ++r) {
    const uint32_t c = hist[r];
    hist[r] = run;  // the region's cursor from here on
    if (c) {
      k.tab[(size_t)r * k.n_tiles + tile] = (run << 16) | c;
      atomicAdd(&k.totals[r], c);
    }
    run += c;
  }
  __syncthreads();
  // sweep 2: every contribution to its place
  Entry* slot = reinterpret_cast<Entry*>(k.entries) + (size_t)tile * Tile::kEntries;
  for (int dl = 0; dl < nd; ++dl) {
    const int d = d0 + dl;
    const int chan = b.channel ? min(max(b.channel[d], 0), g.C - 1) : 0;
    uint32_t grp[kGroup];
    if constexpr (kGroup == 4) {
      const uint2 l2 = lo16[dl * kBlock + threadIdx.x];
      const uint32_t h4 = hi8[dl * kBlock + threadIdx.x];
      grp[0] = (l2.x & 0xffffu) | ((h4 & 0xffu) << 16);
      grp[1] = (l2.x >> 16) | (((h4 >> 8) & 0xffu) << 16);
      grp[2] = (l2.y & 0xffffu) | (((h4 >> 16) & 0xffu) << 16);
      grp[3] = (l2.y >> 16) | ((h4 >> 24) << 16);
    } else {
      const uint32_t l2 = lo16[dl * kBlock + threadIdx.x];
      const uint32_t h2 = hi8[dl * kBlock + threadIdx.x];
      grp[0] = (l2 & 0xffffu) | ((h2 & 0xffu) << 16);
      grp[1] = (l2 >> 16) | ((h2 >> 8) << 16);
    }
#pragma unroll
    for (int q = 0; q < kSpt; ++q) {
      if (sb + q >= k.s1) continue;
      uint32_t word[kCorners];
      double wc[kCorners];
      if (kBil) {
        const float2 o = oxy[(dl * kBlock + threadIdx.x) * kSpt + q];
        bin_corners<kBil>(g, k, ax_eta, ax_xi, o.x, o.y, chan, word, wc);
      } else {
        wc[0] = 1.0;
      }
      const double W = (kW || kBil) && b.weight ? (double)b.weight[(size_t)d * b.ld_w + sb + q] : 1.0;
      const float D = b.tod[(size_t)d * b.ld_tod + sb + q];
#pragma unroll
      for (int c = 0; c < kCorners; ++c) {
        const uint32_t wd = grp[q * kCorners + c];
        if ((wd >> 16) == 0xffu) continue;
        const uint32_t pos = atomicAdd(&hist[wd >> 11], 1u);
        Entry en;
        en.local = (wd & (uint32_t)(kBinRegionPx - 1)) | ((uint32_t)dl << 11);
        en.d = D;
        en.set_weight(W * wc[c]);
        slot[pos] = en;
      }
    }
  }
}

// between the passes: the regions by falling number of contributions (rank by counting: R <= 2048), so that pass B's
// grid meets the regions under the scan's centre -- several times the mean -- first and ends on light ones
__global__ __launch_bounds__(1024) void bin_order_kernel(const uint32_t* __restrict__ totals, int R, int* __restrict__ order) {
  __shared__ uint32_t t[kBinMaxRegions];
  for (int i = threadIdx.x; i < R; i += blockDim.x) t[i] = totals[i];
  __syncthreads();
  for (int i = threadIdx.x; i < R; i += blockDim.x) {
    const uint32_t mine = t[i];
    int rank = 0;
    for (int j = 0; j < R; ++j) rank += (t[j] > mine) || (t[j] == mine && j < i);
    order[rank] = i;
  }
}

// pass B: block (region, split): the region's segments of the split's tiles into LDS, then the
// block of 64 x 32 pixels into the map
template <int kEntryBytes>
__global__ __launch_bounds__(kBlock) void bin_accumulate_kernel(MapArgs g, BinArgs b, BucketArgs k, int splits) {
  using Entry = BinEntryT<kEntryBytes>;
  const Entry* entries = reinterpret_cast<const Entry*>(k.entries);
  extern __shared__ double bin_acc[];  // [S][2][kBinRegionPx]: sum, weight
  __shared__ uint32_t seg_cnt[kBlock];   // a batch's non-empty segments: count,
  __shared__ uint32_t seg_base[kBlock];  // index of the first entry in the work buffer minus its place in the batch's list,
  __shared__ int seg_d0[kBlock];         // first detector of the tile,
  __shared__ int seg_end[kBlock];        // inclusive scan of the counts
  __shared__ int n_seg;
  // workgroup b runs on XCD b mod 8: with the region straight from blockIdx.x and a power-of-two
  // number of regions per map row, an XCD would own whole columns of the map -- and the columns
  // under the scan's centre hold several times the samples of those at its rim (measured: 20 ms
  // against 9 ms for this kernel).  Rotating by the split spreads every column over the XCDs.
  const int r = k.order[(blockIdx.x + blockIdx.y) % (unsigned)k.R];
  for (int i = threadIdx.x; i < g.S * 2 * kBinRegionPx; i += kBlock) bin_acc[i] = 0.0;
  const int per = (k.n_tiles + splits - 1) / splits;
  const int t0 = blockIdx.y * per, t1 = min(k.n_tiles, t0 + per);
  const uint32_t* row = k.tab + (size_t)r * k.n_tiles;
  bool any = false;
  for (int base = t0; base < t1; base += kBlock) {
    if (threadIdx.x == 0) n_seg = 0;
    __syncthreads();
    const int tile = base + threadIdx.x;
    const uint32_t v = tile < t1 ? row[tile] : 0u;
    if (v & 0xffffu) {
      const int at = atomicAdd(&n_seg, 1);
      seg_cnt[at] = v & 0xffffu;
      seg_base[at] = (uint32_t)tile * (uint32_t)k.tile_entries + (v >> 16);  // < 2^32: the host bounds the chunk
      seg_d0[at] = (tile / k.tiles_x) * k.tile_det;
    }
    __syncthreads();
    const int n = n_seg;
    if (n == 0) continue;  // uniform
    any = true;
    // the batch's segments as one list of entries: an inclusive scan of the counts, then every
    // thread strides through the list (its segment index only ever moves forward) -- all lanes
    // busy and the loads of successive entries independent, instead of one wave per segment
    // waiting out the memory latency of each
    seg_end[threadIdx.x] = (int)threadIdx.x < n ? (int)seg_cnt[threadIdx.x] : 0;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
      const int add = (int)threadIdx.x >= off ? seg_end[threadIdx.x - off] : 0;
      __syncthreads();
      seg_end[threadIdx.x] += add;
      __syncthreads();
    }
    if ((int)threadIdx.x < n) seg_base[threadIdx.x] -= (uint32_t)(seg_end[threadIdx.x] - (int)seg_cnt[threadIdx.x]);
    __syncthreads();
    const int total = seg_end[n - 1];
    // kPer entries per thread and trip, kBlock apart: every load instruction reads 1 KiB of
    // consecutive entries (lanes on consecutive entries; 16-byte loads at a 64-byte lane stride
    // ran 40 % slower), and the loads of a trip are issued together -- the kernel is bound by
    // memory latency, not by bytes
    constexpr int kPer = 4;
    int cur = 0;
    auto fetch = [&](int j0, Entry (&en)[kPer], int (&det)[kPer]) {
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int j = j0 + i * kBlock;
        en[i].local = kBinNone;
        det[i] = 0;
        if (j < total) {
          while (j >= seg_end[cur]) ++cur;
          en[i] = entries[seg_base[cur] + (uint32_t)j];
          det[i] = seg_d0[cur] + (int)(en[i].local >> 11);
        }
      }
    };
    auto weights = [&](const Entry (&en)[kPer], const int (&det)[kPer], int s, double (&m)[kPer]) {
#pragma unroll
      for (int i = 0; i < kPer; ++i) m[i] = en[i].local != kBinNone ? g.stokes_w[(size_t)det[i] * g.S + s] : 0.0;
    };
    auto add = [&](const Entry (&en)[kPer], int s, const double (&m)[kPer]) {
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        if (m[i] == 0.0) continue;  // zero weight: nothing to add (np.abs(P) entries that are 0); padding
        const uint32_t px = en[i].local & (uint32_t)(kBinRegionPx - 1);
        const double ww = en[i].weight();
        atomicAdd(&bin_acc[(s * 2) * kBinRegionPx + px], m[i] * (ww * (double)en[i].d));
        atomicAdd(&bin_acc[(s * 2 + 1) * kBinRegionPx + px], fabs(m[i]) * ww);
      }
    };
    // Two trips in flight.  An entry needs a second, dependent load (its detector's Stokes weight); loads return
    // in order, so the weights of this trip are requested FIRST and the next trip's entries after them: waiting
    // for the weights then leaves the entries in flight (14.1 -> 12.6 ms at 10 000 x 240 000; the other order,
    // or one trip at a time, gains nothing).
    auto trip = [&](const Entry (&en)[kPer], const int (&det)[kPer], bool more, int j_next, Entry (&en_n)[kPer], int (&det_n)[kPer]) {
      double m[kPer];
      weights(en, det, 0, m);
      __builtin_amdgcn_sched_barrier(0);
      if (more) fetch(j_next, en_n, det_n);
      __builtin_amdgcn_sched_barrier(0);
      add(en, 0, m);
      for (int s = 1; s < g.S; ++s) {
        weights(en, det, s, m);
        add(en, s, m);
      }
    };
    Entry ea[kPer], eb[kPer];
    int da[kPer], db[kPer];
    constexpr int kTrip = kBlock * kPer;
    int j0 = threadIdx.x;
    if (j0 < total) fetch(j0, ea, da);
    while (j0 < total) {
      const bool more_b = j0 + kTrip < total;
      trip(ea, da, more_b, j0 + kTrip, eb, db);
      if (!more_b) break;
      const bool more_a = j0 + 2 * kTrip < total;
      trip(eb, db, more_a, j0 + 2 * kTrip, ea, da);
      if (!more_a) break;
      j0 += 2 * kTrip;
    }
    __syncthreads();
  }
  if (!any) return;  // uniform: n_seg is read by every thread between barriers
  const int per_plane = k.nby * k.nbx;
  const int chan = r / per_plane, rem = r - chan * per_plane;
  const int by = rem / k.nbx, bx = rem - by * k.nbx;
  const size_t plane = (size_t)g.n_eta * g.n_xi;
  for (int s = 0; s < g.S; ++s) {
    const size_t base = ((size_t)s * g.C + chan) * plane;
    for (int i = threadIdx.x; i < kBinRegionPx; i += kBlock) {
      const double wv = bin_acc[(s * 2 + 1) * kBinRegionPx + i];
      const int e = (by << 5) + (i >> 6), x = (bx << 6) + (i & 63);
      if (wv == 0.0 || e >= g.n_eta || x >= g.n_xi) continue;
      atomicAdd(b.sum + base + (size_t)e * g.n_xi + x, bin_acc[(s * 2) * kBinRegionPx + i]);
      atomicAdd(b.wgt + base + (size_t)e * g.n_xi + x, wv);
    }
  }
}

// kKrj: the field leaves in K_RJ -- every row's four values times the row's scale, divided by den_band(el_det) exactly as
// tod_krj_kernel divides a finished field (mrx_krj.h: the same per-tile elevation model from the same 1024 samples, the same
// lookup), instead of a second pass that reads and writes the field again (3.9 of 18.3 ms at 10 000 x 240 000).
// pairs[plane][e][x] = (m[e][x], m[e + 1][x]) (MapArgs::pairs), one thread per pair
__global__ __launch_bounds__(kBlock) void map_pairs_kernel(const float* __restrict__ m, float2* __restrict__ pairs, int planes, int n_eta, int n_xi) {
  const size_t per = (size_t)(n_eta - 1) * n_xi, n = per * planes;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const size_t p = i / per, r = i - p * per;
    const float* src = m + p * (size_t)n_eta * n_xi + r;
    pairs[i] = make_float2(src[0], src[n_xi]);
  }
}

#ifndef MRX_MAP_CAL_WAVES
#define MRX_MAP_CAL_WAVES 4
#endif
template <bool kChain, bool kCal, int kS, bool kKrj = false>
// without the per-sample atmospheric calibration the kernel fits 168 registers (three waves per
// SIMD: 16.8 -> 14.8 ms at 10 000 x 240 000); with it the cap costs spills (26.8 -> 34.4 ms)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kChain ? (kCal ? 2 : 3) : (kCal ? MRX_MAP_CAL_WAVES : kS == 1 ? 5 : 4), kChain ? (kCal ? 2 : 3) : 8))) void map_sample_kernel(MapArgs g, int groups, MapKrj kj) {
  __shared__ DetConst dets[kTileDet];
  __shared__ float2 edge[2][kBlock];  // (first, last) raw value of every thread, double-buffered
  extern __shared__ __align__(16) float cal_lds[];   // calibration axes and tables (a few KB); K_RJ: the cell table behind them
  __shared__ CalDet cdet[kKrj ? kTileDet : 1];
  __shared__ float red[12];
  CalLds cl{};
  if (kCal) {
    for (int i = threadIdx.x; i < g.n_pwv; i += kBlock) cal_lds[i] = g.cal_pwv[i];
    for (int i = threadIdx.x; i < g.n_el; i += kBlock) cal_lds[g.n_pwv + i] = g.cal_el[i];
    for (int i = threadIdx.x; i < g.C * g.n_pwv * g.n_el; i += kBlock) cal_lds[g.n_pwv + g.n_el + i] = g.cal[i];
  } else {
    // (once per workgroup: per sample and channel this was a float64 product and a conversion)
    for (int i = threadIdx.x; i < g.C; i += kBlock) cal_lds[i] = (float)(1.380649e-11 * g.scalar[i]);
    cl.chan = cal_lds;
  }
  const int s_tile = blockIdx.x * kTileSamples;
  const int sb = s_tile + threadIdx.x * kSamplesPerThread;
  const Axis ax_eta = g.eta, ax_xi = g.xi;
  // the per-sample part (float64 composition of the three rotations, ~500 instruction slots a sample)
  // is shared by the `groups` x 16 detector rows this workgroup walks.  Without the literal chain and
  // the per-sample calibration a sample's record is its six float32 matrix entries: it then lives in LDS
  // (slot 0 / 1025: the halo samples of the 3-tap kernel) and the row loop fetches it with three 8-byte
  // reads, instead of five records held in 65 registers per thread (168 -> 63 registers).
  constexpr bool kLdsSc = !kChain;
  // The records as six planes (round 6; an array of 24-byte records before): entry k of slot i at sc_lds[k][i + 3], so that a
  // thread's four samples' entries -- slots 1 + 4 tid .. 4 + 4 tid -- are ONE aligned 16-byte read per plane, consecutive
  // lanes at consecutive addresses.  As records, a thread's slots lay 96 bytes apart: its three 8-byte reads a sample met
  // the banks four lanes at a time.
  constexpr int kRec = 6, kRecPitch = (kTileSamples + 2 + 3 + 3) & ~3;
  __shared__ __align__(16) float sc_lds[kLdsSc ? kRec * kRecPitch : 8];
  // the interval form of the calibration (see kMaxSteps): (cos, sin) of (boresight elevation - pi/2) at the start, the middle
  // and the end of every coarse step the tile meets; a row with a knot off the tables (NaN)
  __shared__ float2 knot_cs[kLdsSc && kCal ? 2 * kMaxSteps + 1 : 1];
  __shared__ int bad_row[kLdsSc && kCal ? kTileDet : 1];
  SampleConst sc[kSamplesPerThread], sc_halo;
  const bool first = threadIdx.x == 0, last = threadIdx.x == kBlock - 1;
  if (kLdsSc) {
    for (int i = threadIdx.x; i < kTileSamples + 2; i += kBlock) {
      SampleConst one;
      sample_const(g, s_tile - 1 + i, false, one);
#pragma unroll
      for (int k = 0; k < 6; ++k) sc_lds[k * kRecPitch + i + 3] = one.G[k];
    }
  } else {
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) {
      sample_const(g, sb + q, kChain, sc[q]);
      sc[q].s = min(max(sb + q, 0), g.T - 1);
    }
    if (first || last) {
      const int sh = first ? s_tile - 1 : s_tile + kTileSamples;
      sample_const(g, sh, kChain, sc_halo);
      sc_halo.s = min(max(sh, 0), g.T - 1);
    }
  }
  float4* const cal_cells = reinterpret_cast<float4*>(cal_lds) + kj.cells_offset;
  KrjSamples ks{};
  if constexpr (kKrj) ks = krj_prologue(cal_cells, red, kj.bore_el, g.T, sb, kj.cal_axis, kj.cal_values, kj.n_el, kj.n_bands);  // (ends with a barrier)
  __syncthreads();
  if (kCal) {
    cl.pwv = make_rgi_axis(cal_lds, g.n_pwv);
    cl.el = make_rgi_axis(cal_lds + g.n_pwv, g.n_el);
    cl.tab = cal_lds + g.n_pwv + g.n_el;
  }
  // slot of a sample's record: the thread's q-th sample, or its halo sample
  auto record = [&](int slot) {
    SampleConst one{};
#pragma unroll
    for (int k = 0; k < 6; ++k) one.G[k] = sc_lds[k * kRecPitch + slot + 3];
    return one;
  };
  // the four samples of a thread at once: G[k][q], slot0 = 1 + 4 tid
  auto record4 = [&](int slot0, float (&G)[6][4]) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(sc_lds + k * kRecPitch + slot0 + 3);
      G[k][0] = v.x; G[k][1] = v.y; G[k][2] = v.z; G[k][3] = v.w;
    }
  };
  // the calibration's part of a record, recomputed (the per-sample fallback; sample_const's arithmetic)
  auto with_cal = [&](SampleConst one, int slot) {
    const int sidx = min(max(s_tile - 1 + slot, 0), g.T - 1);
    const float a = __fsub_rn(g.el[sidx], kHalfPiF);
    one.ca = cosf(a);
    one.sa = sinf(a);
    const double tt = g.t[sidx];
    int jj = (int)floor(fmin(fmax((tt - g.ta0) * g.inv_dta, -1.0), 2.0e9));
    jj = min(max(jj, 0), g.Ta - 2);
    one.jj = jj;
    one.u = (double)(float)((tt - (g.ta0 + (double)jj * g.dta)) * g.inv_dta);
    return one;
  };
  const bool full = (sb + kSamplesPerThread <= g.T) && g.vec_ok;
  // The interval form of the per-sample calibration (round 6).  The factor -- pW per K_RJ at the sample's zenith pwv and
  // the detector's elevation, per channel -- along a row: the pwv is piecewise LINEAR between the coarse samples (0.1 s;
  // sim/atmosphere.py:30-37), the elevation a smooth scan.  Inside one coarse step the factor is therefore smooth, and its
  // kinks sit at the steps' ends.  So it is looked up ONCE per row at the start, the middle and the end of every coarse
  // step the tile meets (16 rows x ~26 steps x 3 lookups a group, five a thread, all of a thread's in flight together),
  // kept in LDS as the parabola through the three in the step's own coordinate, and a sample evaluates its step's
  // parabola at its place in the step: the kinks are where the reference has them, the elevation's curvature inside a step
  // is the parabola's (what is left is the third difference over 0.1 s: below 1e-8 of the factor), and a kink of the
  // TABLES inside a step is missed by less than 1e-7 of the factor.  Rounds 3-5 looked the factor up per thread and row (at
  // its first sample; a lane shuffle for the chord to the next thread's): a chain of LDS reads, two global loads and a
  // barrier per ROW -- a third of the kernel's instructions and most of its waiting.  The form applies to whole tiles
  // that meet at most coef_cap steps (the launcher's table; mrx_map_cal.steps_per_tile) of at least eight samples; a row
  // with a knot off the tables (NaN: jax's fill) takes the per-sample form, which marks exactly the samples that are off.
  bool coef_ok = false;
  int j_lo = 0, n_int = 0, rb = 0;
  float v0 = 0.0f, dvt = 0.0f, lim = 1.0f;
  float* const coef = cal_lds + (g.coef_offset >= 0 ? g.coef_offset : 0);
  const int coef_pitch = 3 * g.coef_cap + 1;
  if constexpr (kCal && kLdsSc) {
    auto step_of = [&](int sidx, double& u) {  // sample_const's interval and weight
      const double tt = g.t[min(max(sidx, 0), g.T - 1)];
      int jj = (int)floor(fmin(fmax((tt - g.ta0) * g.inv_dta, -1.0), 2.0e9));
      jj = min(max(jj, 0), g.Ta - 2);
      u = (tt - (g.ta0 + (double)jj * g.dta)) * g.inv_dta;
      return jj;
    };
    double u_lo, u_hi;
    j_lo = step_of(s_tile - 1, u_lo);
    n_int = step_of(s_tile + kTileSamples, u_hi) - j_lo + 1;
    coef_ok = g.coef_offset >= 0 && s_tile + kTileSamples <= g.T && n_int <= min(g.coef_cap, kMaxSteps) && n_int * 8 <= kTileSamples;  // (uniform)
    if (coef_ok) {
      // the thread's samples in units of a step, counted from the start of the step of its earliest sample (the first
      // thread's halo): v0 at its first sample, dvt a sample (the samples are evenly spaced over a thread's 10 ms)
      double u_b, u_0, u_3;
      const int jb = step_of(first ? s_tile - 1 : sb, u_b), j0 = step_of(sb, u_0), j3 = step_of(sb + kSamplesPerThread - 1, u_3);
      v0 = (float)((double)(j0 - jb) + u_0);
      dvt = (float)(((double)(j3 - jb) + u_3 - ((double)(j0 - jb) + u_0)) * (1.0 / (kSamplesPerThread - 1)));
      lim = jb >= g.Ta - 2 ? 3.0e38f : 1.0f;  // (past the last coarse sample the series is extrapolated: still the last step)
      rb = jb - j_lo;
      // the boresight at the steps' starts, middles and ends: linear between the two samples around the knot's time
      const int s_a = max(s_tile - 1, 0), s_b = min(s_tile + kTileSamples, g.T - 1);
      const double t_a = g.t[s_a], inv_dt = (double)(s_b - s_a) / (g.t[s_b] - t_a);
      for (int m = threadIdx.x; m < 2 * n_int + 1; m += kBlock) {
        const double tau = g.ta0 + ((double)j_lo + 0.5 * (double)m) * g.dta;
        int sidx = min(max(s_a + (int)floor(fmin(fmax((tau - t_a) * inv_dt, -2.0e9), 2.0e9)), 0), g.T - 2);
        for (int it = 0; it < 8 && sidx < g.T - 2 && g.t[sidx + 1] < tau; ++it) ++sidx;  // (unevenly spaced samples)
        for (int it = 0; it < 8 && sidx > 0 && g.t[sidx] > tau; ++it) --sidx;
        const double wt = (tau - g.t[sidx]) / (g.t[sidx + 1] - g.t[sidx]);
        const float e0 = g.el[sidx], e1 = g.el[sidx + 1];
        const float a = __fsub_rn((float)((double)e0 + wt * ((double)e1 - (double)e0)), kHalfPiF);
        knot_cs[m] = make_float2(cosf(a), sinf(a));
      }
    }
  }
  for (int grp = 0; grp < groups; ++grp) {
  const int d0 = (blockIdx.y * groups + grp) * kTileDet;
  if (d0 >= g.D) break;
  const int nd = min(kTileDet, g.D - d0);
  __syncthreads();  // the previous group is done with dets[], cdet[] and edge[]
  if ((int)threadIdx.x < nd) dets[threadIdx.x] = make_det_const(g, d0 + threadIdx.x);
  if (kCal && kLdsSc && (int)threadIdx.x < kTileDet) bad_row[threadIdx.x] = 0;
  if constexpr (kKrj) krj_stage_rows(cdet, red, kj.dx, kj.dy, kj.band, kj.scale, kj.n_bands, d0, nd, cal_cells, kj.n_el);
  __syncthreads();
  // The interval form of the per-sample calibration: this group's rows (see the tile's part above)
  if constexpr (kCal && kLdsSc) {
    if (coef_ok) {
      for (int i = threadIdx.x; i < kTileDet * n_int; i += kBlock) {
        const int dl = i & (kTileDet - 1), r = i / kTileDet;
        if (dl >= nd) continue;
        const int j = j_lo + r, d = d0 + dl;
        const double y0 = g.pwv[(size_t)j * g.D + d], y1 = g.pwv[(size_t)(j + 1) * g.D + d];
        float F[3][kCalFastChannels];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // the step's start, middle and end
          const float2 cs = knot_cs[2 * r + k];
          const float im = __fadd_rn(__fmul_rn(dets[dl].c_re, cs.y), __fmul_rn(dets[dl].c_cr, cs.x));
          const CalCell cell = cal_cell(cl, (float)fma(0.5 * (double)k, y1 - y0, y0), asin_poly(im));  // (pwv demoted to float32 by the jax interpolator)
#pragma unroll
          for (int c = 0; c < kCalFastChannels; ++c) {
            F[k][c] = 0.0f;
            if (c < g.C) {  // (uniform)
              F[k][c] = cal_factor(g, cl, cell, c);
              bad |= !(F[k][c] == F[k][c]);
            }
          }
        }
#pragma unroll
        for (int c = 0; c < kCalFastChannels; ++c)
          if (c < g.C) {  // the parabola through (0, F0), (1/2, F1), (1, F2): F0 + w (4 F1 - 3 F0 - F2) + w^2 (2 F0 + 2 F2 - 4 F1)
            float* rec = coef + (c * kTileDet + dl) * coef_pitch + 3 * r;
            rec[0] = F[0][c];
            rec[1] = fmaf(4.0f, F[1][c], fmaf(-3.0f, F[0][c], -F[2][c]));
            rec[2] = fmaf(-4.0f, F[1][c], 2.0f * (F[0][c] + F[2][c]));
          }
        if (bad) bad_row[dl] = 1;
      }
      __syncthreads();
    }
  }
  for (int dl = 0; dl < nd; ++dl) {
    const DetConst dc = dets[dl];
    const int d = d0 + dl;
    float r[kSamplesPerThread];
    float halo = 0.0f;
    // Round 6: the thread's four samples TOGETHER (row_samples: their LDS reads, then their gathers, in flight at once);
    // the tile's halo samples ride along as a fifth sample of waves 0 and 3 (in a branch of their own the first and the
    // last thread's waves ran a whole sample's chain of latencies alone, once per row)
    auto batched = [&](auto unroll_c, auto pw) {
      const int s0 = 1 + threadIdx.x * kSamplesPerThread;
      const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
      auto run = [&](auto bil) {
        if (wave == 0 || wave == kBlock / 64 - 1) {
          const int slot[5] = {s0, s0 + 1, s0 + 2, s0 + 3, first ? 0 : last ? kTileSamples + 1 : s0 + 3};
          float v[5];
          row_samples<decltype(bil)::value, kS, 5, decltype(unroll_c)::value>(g, ax_eta, ax_xi, dc, slot, record, record4, pw, v);
          r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3];
          halo = v[4];
        } else {
          const int slot[4] = {s0, s0 + 1, s0 + 2, s0 + 3};
          row_samples<decltype(bil)::value, kS, 4, decltype(unroll_c)::value>(g, ax_eta, ax_xi, dc, slot, record, record4, pw, r);
        }
      };
      if (g.bilinear) run(std::true_type{}); else run(std::false_type{});  // (uniform)
    };
    bool done = false;
    if constexpr (kCal && kLdsSc) {
      if (coef_ok && __builtin_amdgcn_readfirstlane(bad_row[dl]) == 0) {  // (uniform)
        batched(std::true_type{}, [&](int c, auto& f) {
          const float* cf = coef + (c * kTileDet + dl) * coef_pitch + 3 * rb;
          constexpr int n = sizeof(f) / sizeof(f[0]);
#pragma unroll
          for (int q = 0; q < n; ++q) {
            // the sample's place in its coarse step: v counts from the start of the thread's first step; past 1 it is the
            // next step's (one record on).  (The halo sample, q = 4: one sample before the first thread's first, or one
            // after the last thread's last.)
            const float v = fmaf(q == 4 ? (first ? -1.0f : 4.0f) : (float)q, dvt, v0);
            const bool next = v >= lim;
            const float w = next ? v - 1.0f : v;
            const float* rec = next ? cf + 3 : cf;
            f[q] = fmaf(w, fmaf(w, rec[2], rec[1]), rec[0]);
          }
        });
        done = true;
      }
    }
    if constexpr (!kCal && kLdsSc) {
      batched(std::false_type{}, [&](int c, auto& f) {
        const float v = cl.chan[c];
        constexpr int n = sizeof(f) / sizeof(f[0]);
#pragma unroll
        for (int q = 0; q < n; ++q) f[q] = v;
      });
      done = true;  // (nothing left for the per-sample form below)
    }
    if (!done) {
      // the per-sample form: the literal chain, more than kCalFastChannels channels, a ragged last tile, a row off the tables
      int jj0 = -1;  // (the coarse pwv pair the caller holds: none -- raw_sample loads the sample's own)
      double y0 = 0.0, y1 = 0.0;
      if (kCal && !kLdsSc) {
        jj0 = sc[0].jj;
        y0 = g.pwv[(size_t)jj0 * g.D + d];
        y1 = g.pwv[(size_t)(jj0 + 1) * g.D + d];
      }
      auto rec_of = [&](int slot) { return (kCal && kLdsSc) ? with_cal(record(slot), slot) : record(slot); };
      if constexpr (kLdsSc) {
        // (one sample at a time, not unrolled: this form is the exception, and four samples of it side by side set the
        // whole kernel's register count -- 138 against the batched form's own need)
#pragma unroll 1
        for (int q = 0; q < kSamplesPerThread; ++q) {
          const float v = raw_sample<kChain, kCal, kS>(g, cl, ax_eta, ax_xi, dc, d, rec_of(1 + threadIdx.x * kSamplesPerThread + q), jj0, y0, y1);
          r[0] = q == 0 ? v : r[0]; r[1] = q == 1 ? v : r[1]; r[2] = q == 2 ? v : r[2]; r[3] = q == 3 ? v : r[3];
        }
      } else {
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q)
          r[q] = raw_sample<kChain, kCal, kS>(g, cl, ax_eta, ax_xi, dc, d, sc[q], jj0, y0, y1);
      }
    if (first || last)
      halo = raw_sample<kChain, kCal, kS>(g, cl, ax_eta, ax_xi, dc, d, kLdsSc ? rec_of(first ? 0 : kTileSamples + 1) : sc_halo, jj0, y0, y1);
    }
    edge[dl & 1][threadIdx.x] = make_float2(r[0], r[kSamplesPerThread - 1]);
    __syncthreads();
    const float left = first ? halo : edge[dl & 1][threadIdx.x - 1].y;
    const float right = last ? halo : edge[dl & 1][threadIdx.x + 1].x;
    // scipy.ndimage.convolve1d, symmetric 3-tap kernel (map.py:170): 0.5 r + 0.25 (prev + next); the
    // float32 form differs from scipy's float64 accumulation by one rounding of the float32 result
    float o[kSamplesPerThread];
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) {
      const float prev = q == 0 ? left : r[q - 1];
      const float next = q == kSamplesPerThread - 1 ? right : r[q + 1];
      o[q] = fmaf(r[q], 0.5f, (prev + next) * 0.25f);
    }
    if constexpr (kKrj) {
      const CalDet c = cdet[dl];
      float sv[kSamplesPerThread];
#pragma unroll
      for (int q = 0; q < kSamplesPerThread; ++q) sv[q] = c.scale * o[q];
      const float4* C = cal_cells + c.band * (kj.n_el - 1);
      krj_row<false>(c, C, kj.n_el, ks, sv, o, kj.bore_el, sb, g.T, cal_cells, kj.cal_axis);
    }
    float* dst = g.out + (size_t)d * g.ld + sb;
    if (full) {
      const vfloat4 v = {o[0], o[1], o[2], o[3]};
      __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(dst));
    } else {
#pragma unroll
      for (int q = 0; q < kSamplesPerThread; ++q)
        if (sb + q < g.T) dst[q] = o[q];
    }
  }
  }
}

}  // namespace

// mrx_map_sample (krj == nullptr) and mrx_map_sample_krj
static int map_sample(mrx_ctx* ctx, const mrx_sky_map* map, const mrx_map_cal* cal,
                      const float* d_az, const float* d_el, int T, const double* d_transform,
                      const float* d_dx, const float* d_dy, const double* d_stokes_w, int D,
                      float* d_out, size_t ld_out, const MapKrj* krj) {
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, map && cal && d_az && d_el && d_dx && d_dy && d_stokes_w && d_out, "null pointer");
  MRX_REQUIRE(ctx, map->d_values, "null map pointer");
  MRX_REQUIRE(ctx, map->deta != 0.0 && map->dxi != 0.0, "map axes need a non-zero step");
  MRX_REQUIRE(ctx, map->n_channels >= 1 && map->n_stokes >= 1 && map->n_stokes <= kMaxStokes &&
                       map->n_eta >= 2 && map->n_xi >= 2,
              "need n_channels >= 1, 1 <= n_stokes <= 4, n_eta >= 2, n_xi >= 2");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  if (cal->d_table) {
    MRX_REQUIRE(ctx, cal->d_axis_pwv && cal->d_axis_el && cal->d_pwv && cal->d_t, "null calibration pointer");
    MRX_REQUIRE(ctx, cal->n_pwv >= 2 && cal->n_el >= 2 && cal->Ta >= 2 && cal->dta > 0.0,
                "calibration needs n_pwv >= 2, n_el >= 2, Ta >= 2, dta > 0");
  } else {
    MRX_REQUIRE(ctx, cal->d_scalar, "need a per-channel scalar calibration without a table");
  }
  MapArgs g{};
  g.values = map->d_values;
  g.eta = make_axis(map->n_eta, map->eta0, map->deta);
  g.xi = make_axis(map->n_xi, map->xi0, map->dxi);
  g.C = map->n_channels;
  g.S = map->n_stokes;
  g.n_eta = map->n_eta;
  g.n_xi = map->n_xi;
  g.cphi = (float)map->center_phi;
  // exp(1j * (pi/2 - ctheta)) as jax evaluates it: the python float demoted to float32,
  // then a complex64 exponential
  const float ang = (float)(1.5707963267948966 - map->center_theta);
  g.rot_re = (float)cos((double)ang);
  g.rot_im = (float)sin((double)ang);
  g.cos_cphi = cos(map->center_phi);
  g.sin_cphi = sin(map->center_phi);
  g.cos_ctheta = cos(map->center_theta);
  g.sin_ctheta = sin(map->center_theta);
  g.bilinear = map->bilinear;
  {
    // (the sampler reads the map as ONE raw buffer: 32-bit offsets)
    const unsigned long long bytes = 4ull * (unsigned long long)map->n_channels * map->n_stokes * map->n_eta * map->n_xi;
    MRX_REQUIRE(ctx, bytes < (1ull << 32), "the map (all channels and Stokes planes) must be smaller than 4 GiB");
    g.map_bytes = (uint32_t)bytes;
  }
  g.cal = cal->d_table;
  g.cal_pwv = cal->d_axis_pwv;
  g.cal_el = cal->d_axis_el;
  g.n_pwv = cal->n_pwv;
  g.n_el = cal->n_el;
  g.pwv = cal->d_pwv;
  g.Ta = cal->Ta;
  g.ta0 = cal->ta0;
  g.dta = cal->dta;
  g.inv_dta = cal->dta > 0.0 ? 1.0 / cal->dta : 0.0;
  g.t = cal->d_t;
  g.scalar = cal->d_scalar;
  g.az = d_az;
  g.el = d_el;
  g.transform = d_transform;
  g.dx = d_dx;
  g.dy = d_dy;
  g.stokes_w = d_stokes_w;
  g.D = D;
  g.T = T;
  g.out = d_out;
  g.ld = ld_out;
  g.vec_ok = (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0);
  // detector tiles per workgroup (they share the per-sample constants): MRX_MAP_GROUPS while that leaves >= 16
  // workgroups per CU in the grid
#ifndef MRX_MAP_GROUPS
#define MRX_MAP_GROUPS 4
#endif
  int groups = MRX_MAP_GROUPS;
  while (groups > 1 && (long long)mrx_ceil_div(T, kTileSamples) * mrx_ceil_div(D, kTileDet * groups) < 16LL * 256) groups /= 2;
  dim3 grid(mrx_ceil_div(T, kTileSamples), mrx_ceil_div(D, kTileDet * groups));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  size_t lds = 0;
  g.coef_offset = -1;
  g.coef_cap = 0;
  if (cal->d_table) {
    lds = sizeof(float) * ((size_t)cal->n_pwv + cal->n_el + (size_t)map->n_channels * cal->n_pwv * cal->n_el);
    MRX_REQUIRE(ctx, lds <= 48 * 1024, "calibration tables of all channels must fit in 48 KiB");
    if (map->n_channels <= kCalFastChannels && T >= 2) {  // the interval table of the per-sample calibration behind them
      g.coef_cap = cal->steps_per_tile > 0 ? std::min(cal->steps_per_tile, kMaxSteps) : kDefaultSteps;
      g.coef_offset = (int)(lds / sizeof(float));
      lds += sizeof(float) * (size_t)map->n_channels * kTileDet * (3 * g.coef_cap + 1);
    }
  } else {
    lds = sizeof(float) * (size_t)map->n_channels;  // the channels' scalar factors
    MRX_REQUIRE(ctx, lds <= 48 * 1024, "too many channels");
  }
  MapKrj kj{};
  if (krj) {  // the cell table of TOD.to("K_RJ") behind the sampler's own tables, on a 16-byte boundary
    kj = *krj;
    lds = (lds + 15) / 16 * 16;
    kj.cells_offset = (int)(lds / 16);
    lds += sizeof(float4) * (size_t)(kj.n_el - 1) * kj.n_bands;
    MRX_REQUIRE(ctx, lds <= 60 * 1024, "the calibration tables (sampling and K_RJ) must fit in 60 KiB");
  }
  MRX_REQUIRE(ctx, (long long)map->n_eta * map->n_xi < (1LL << 29), "a map plane must hold fewer than 2^29 pixels");
  MRX_REQUIRE(ctx, map->n_eta < (1 << 23) && map->n_xi < (1 << 23) && std::fabs(map->eta0 / map->deta) < 8388608.0 &&
                       std::fabs(map->xi0 / map->dxi) < 8388608.0,
              "a map axis must have fewer than 2^23 nodes and start within 2^23 pixels of the map's centre");
  const bool chain = ctx->options[MRX_OPT_POINTING_CHAIN] != 0, has_cal = cal->d_table != nullptr;
  g.pairs = nullptr;
  g.pairs_bytes = 0;
  if (!chain) {
    const int planes = map->n_channels * map->n_stokes;
    const unsigned long long bytes = 8ull * (unsigned long long)planes * (map->n_eta - 1) * map->n_xi;
    MRX_REQUIRE(ctx, bytes < (1ull << 32), "the map (all channels and Stokes planes) must be smaller than 2 GiB (its row pairs than 4 GiB)");
    if (!ctx->map_pairs_read) MRX_HIP(ctx, hipEventCreateWithFlags(&ctx->map_pairs_read, hipEventDisableTiming));
    else MRX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->map_pairs_read, 0));  // (the sampler that last read the copy)
    if (ctx->map_pairs_cap < bytes) {  // (grown on demand: rare)
      if (ctx->d_map_pairs) {
        MRX_HIP(ctx, hipDeviceSynchronize());
        (void)hipFree(ctx->d_map_pairs);
        ctx->d_map_pairs = nullptr;
        ctx->map_pairs_cap = 0;
      }
      MRX_HIP(ctx, hipMalloc(&ctx->d_map_pairs, (size_t)bytes));
      ctx->map_pairs_cap = (size_t)bytes;
    }
    const unsigned long long n_pairs = bytes / 8;
    const unsigned blocks = (unsigned)std::min<unsigned long long>((n_pairs + kBlock - 1) / kBlock, 65536ull);
    hipLaunchKernelGGL(map_pairs_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, map->d_values, reinterpret_cast<float2*>(ctx->d_map_pairs),
                       planes, map->n_eta, map->n_xi);
    g.pairs = ctx->d_map_pairs;
    g.pairs_bytes = (uint32_t)bytes;
  }
  if (krj && chain)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "mrx_map_sample_krj: not with MRX_OPT_POINTING_CHAIN (sample in pW, then mrx_tod_to_krj)");
  // (the kernel's static LDS -- sample records, edge exchange -- is up to 46 KiB: with large calibration
  // tables the sum passes the 64 KiB a launch gets by default)
#define MRX_LAUNCH_MAP(CH, CA, S, K)                                                                           \
  do {                                                                                                         \
    if (lds + 48 * 1024 > 64 * 1024)                                                                           \
      MRX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(map_sample_kernel<CH, CA, S, K>),        \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
    hipLaunchKernelGGL((map_sample_kernel<CH, CA, S, K>), grid, dim3(kBlock), lds, ctx->stream, g, groups, kj); \
  } while (0)
#define MRX_LAUNCH_MAP_S(CH, CA, K)                    \
  switch (map->n_stokes) {                             \
    case 1: MRX_LAUNCH_MAP(CH, CA, 1, K); break;       \
    case 2: MRX_LAUNCH_MAP(CH, CA, 2, K); break;       \
    case 3: MRX_LAUNCH_MAP(CH, CA, 3, K); break;       \
    default: MRX_LAUNCH_MAP(CH, CA, 4, K); break;      \
  }
  if (chain) {
    if (has_cal) { MRX_LAUNCH_MAP_S(true, true, false); } else { MRX_LAUNCH_MAP_S(true, false, false); }
  } else if (krj) {
    if (has_cal) { MRX_LAUNCH_MAP_S(false, true, true); } else { MRX_LAUNCH_MAP_S(false, false, true); }
  } else {
    if (has_cal) { MRX_LAUNCH_MAP_S(false, true, false); } else { MRX_LAUNCH_MAP_S(false, false, false); }
  }
#undef MRX_LAUNCH_MAP_S
#undef MRX_LAUNCH_MAP
  MRX_CHECK_LAUNCH(ctx);
  if (g.pairs) MRX_HIP(ctx, hipEventRecord(ctx->map_pairs_read, ctx->stream));
  return MRX_OK;
}

extern "C" {

int mrx_map_sample(mrx_ctx* ctx, const mrx_sky_map* map, const mrx_map_cal* cal,
                   const float* d_az, const float* d_el, int T, const double* d_transform,
                   const float* d_dx, const float* d_dy, const double* d_stokes_w, int D,
                   float* d_out, size_t ld_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  return map_sample(ctx, map, cal, d_az, d_el, T, d_transform, d_dx, d_dy, d_stokes_w, D, d_out, ld_out, nullptr);
}

int mrx_map_sample_krj(mrx_ctx* ctx, const mrx_sky_map* map, const mrx_map_cal* cal,
                       const float* d_az, const float* d_el, int T, const double* d_transform,
                       const float* d_dx, const float* d_dy, const double* d_stokes_w, int D,
                       const float* d_scale, const float* d_bore_el, const float* d_krj_dx, const float* d_krj_dy,
                       const int32_t* d_band, const float* d_cal_axis_el, const float* d_cal_values, int n_el, int n_bands,
                       float* d_out, size_t ld_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, d_bore_el && d_krj_dx && d_krj_dy && d_band && d_cal_axis_el && d_cal_values, "null K_RJ pointer");
  MRX_REQUIRE(ctx, n_el >= 2 && n_bands >= 1, "need n_el >= 2 and n_bands >= 1");
  MapKrj kj{};
  kj.bore_el = d_bore_el;
  kj.dx = d_krj_dx;
  kj.dy = d_krj_dy;
  kj.band = d_band;
  kj.scale = d_scale;
  kj.cal_axis = d_cal_axis_el;
  kj.cal_values = d_cal_values;
  kj.n_el = n_el;
  kj.n_bands = n_bands;
  return map_sample(ctx, map, cal, d_az, d_el, T, d_transform, d_dx, d_dy, d_stokes_w, D, d_out, ld_out, &kj);
}

static int bin_map_args(mrx_ctx* ctx, const mrx_sky_map* map, const float* d_tod, size_t ld_tod,
                        const float* d_weight, size_t ld_weight, const float* d_az, const float* d_el, int T,
                        const double* d_transform, const float* d_dx, const float* d_dy,
                        const double* d_stokes_w, const int32_t* d_channel, int D, double* d_sum,
                        double* d_wgt, MapArgs& g, BinArgs& b) {
  MRX_REQUIRE(ctx, map && d_tod && d_az && d_el && d_dx && d_dy && d_stokes_w && d_sum && d_wgt, "null pointer");
  MRX_REQUIRE(ctx, map->n_channels >= 1 && map->n_stokes >= 1 && map->n_stokes <= kMaxStokes &&
                       map->n_eta >= 2 && map->n_xi >= 2,
              "need n_channels >= 1, 1 <= n_stokes <= 4, n_eta >= 2, n_xi >= 2");
  MRX_REQUIRE(ctx, map->deta != 0.0 && map->dxi != 0.0, "map axes need a non-zero step");
  MRX_REQUIRE(ctx, (long long)map->n_eta * map->n_xi < (1LL << 31), "a map plane must hold fewer than 2^31 pixels");
  MRX_REQUIRE(ctx, ld_tod >= (size_t)T && (!d_weight || ld_weight >= (size_t)T), "leading dimension smaller than T");
  g = MapArgs{};
  g.eta = make_axis(map->n_eta, map->eta0, map->deta);
  g.xi = make_axis(map->n_xi, map->xi0, map->dxi);
  g.C = map->n_channels;
  g.S = map->n_stokes;
  g.n_eta = map->n_eta;
  g.n_xi = map->n_xi;
  g.cphi = (float)map->center_phi;
  const float ang = (float)(1.5707963267948966 - map->center_theta);
  g.rot_re = (float)cos((double)ang);
  g.rot_im = (float)sin((double)ang);
  g.cos_cphi = cos(map->center_phi);
  g.sin_cphi = sin(map->center_phi);
  g.cos_ctheta = cos(map->center_theta);
  g.sin_ctheta = sin(map->center_theta);
  g.bilinear = map->bilinear;
  g.az = d_az;
  g.el = d_el;
  g.transform = d_transform;
  g.dx = d_dx;
  g.dy = d_dy;
  g.stokes_w = d_stokes_w;
  g.D = D;
  g.T = T;
  b = BinArgs{d_tod, ld_tod, d_weight, ld_weight, d_channel, d_sum, d_wgt};
  return MRX_OK;
}

int mrx_bin_map(mrx_ctx* ctx, const mrx_sky_map* map, const float* d_tod, size_t ld_tod,
                const float* d_weight, size_t ld_weight, const float* d_az, const float* d_el, int T,
                const double* d_transform, const float* d_dx, const float* d_dy,
                const double* d_stokes_w, const int32_t* d_channel, int D, double* d_sum,
                double* d_wgt) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MapArgs g;
  BinArgs b;
  const int rc = bin_map_args(ctx, map, d_tod, ld_tod, d_weight, ld_weight, d_az, d_el, T, d_transform, d_dx, d_dy,
                              d_stokes_w, d_channel, D, d_sum, d_wgt, g, b);
  if (rc != MRX_OK) return rc;
  dim3 grid(mrx_ceil_div(T, kTileSamples), mrx_ceil_div(D, kTileDet));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  if (ctx->options[MRX_OPT_POINTING_CHAIN])
    hipLaunchKernelGGL(bin_map_kernel<true>, grid, dim3(kBlock), 0, ctx->stream, g, b);
  else
    hipLaunchKernelGGL(bin_map_kernel<false>, grid, dim3(kBlock), 0, ctx->stream, g, b);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

// regions of a map for the bucketed form, or 0 when it does not apply
static int bin_regions(const mrx_sky_map* map, int* nbx, int* nby) {
  if (!map || map->n_eta < 2 || map->n_xi < 2 || map->n_channels < 1) return 0;
  *nbx = mrx_ceil_div(map->n_xi, kBinBx);
  *nby = mrx_ceil_div(map->n_eta, kBinBy);
  const long long R = (long long)map->n_channels * *nbx * *nby;
  return R <= kBinMaxRegions ? (int)R : 0;
}

// tile geometry of the form a map takes
struct BinGeometry {
  int tile_det, tile_samples, tile_entries;
  size_t lds_a;
};

static BinGeometry bin_geometry(bool bilinear, int R) {
  BinGeometry q;
  if (bilinear) {
    q.tile_det = BinTile<true>::kDet;
    q.tile_samples = BinTile<true>::kSamples;
    q.tile_entries = BinTile<true>::kEntries;
  } else {
    q.tile_det = BinTile<false>::kDet;
    q.tile_samples = BinTile<false>::kSamples;
    q.tile_entries = BinTile<false>::kEntries;
  }
  // sort words at 3 bytes each (two arrays), one word per region (count, then cursor), the bilinear form's offsets
  q.lds_a = (size_t)q.tile_entries * 3 + (size_t)R * sizeof(uint32_t) +
            (bilinear ? (size_t)q.tile_det * q.tile_samples * sizeof(float2) : 0);
  return q;
}

int mrx_bin_map_work_bytes(const mrx_sky_map* map, int D, int T, size_t* min_bytes, size_t* full_bytes) {
  int nbx, nby;
  const int R = bin_regions(map, &nbx, &nby);
  if (!R || D < 1 || T < 1 || !min_bytes || !full_bytes) return R ? MRX_ERR_INVALID : MRX_ERR_UNSUPPORTED;
  const BinGeometry q = bin_geometry(map->bilinear != 0, R);
  // one column of tiles (all detectors x one tile of samples): its slots and its words of the table
  // (sized for the largest entry: the nearest-pixel forms need 12 or 8 of the 16 bytes)
  const size_t col = (size_t)mrx_ceil_div(D, q.tile_det) * ((size_t)q.tile_entries * 16 + (size_t)R * sizeof(uint32_t));
  *min_bytes = col;
  *full_bytes = col * (size_t)mrx_ceil_div(T, q.tile_samples);
  return MRX_OK;
}

int mrx_bin_map_bucketed(mrx_ctx* ctx, const mrx_sky_map* map, const float* d_tod, size_t ld_tod,
                         const float* d_weight, size_t ld_weight, const float* d_az, const float* d_el, int T,
                         const double* d_transform, const float* d_dx, const float* d_dy,
                         const double* d_stokes_w, const int32_t* d_channel, int D, double* d_sum,
                         double* d_wgt, void* d_work, size_t work_bytes) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  BucketArgs k{};
  k.R = bin_regions(map, &k.nbx, &k.nby);
  if (!k.R)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "mrx_bin_map_bucketed: maps of at most %d regions of %d x %d pixels (use mrx_bin_map)",
                    kBinMaxRegions, kBinBx, kBinBy);
  MapArgs g;
  BinArgs b;
  const int rc = bin_map_args(ctx, map, d_tod, ld_tod, d_weight, ld_weight, d_az, d_el, T, d_transform, d_dx, d_dy,
                              d_stokes_w, d_channel, D, d_sum, d_wgt, g, b);
  if (rc != MRX_OK) return rc;
  const bool bil = map->bilinear != 0;
  const BinGeometry q = bin_geometry(bil, k.R);
  k.tile_det = q.tile_det;
  k.tile_entries = q.tile_entries;
  const int tiles_y = mrx_ceil_div(D, q.tile_det);
  MRX_REQUIRE(ctx, tiles_y <= 65535, "D too large for one launch");
  const int entry_bytes = bin_entry_bytes(bil, d_weight != nullptr);
  const size_t col = (size_t)tiles_y * ((size_t)q.tile_entries * entry_bytes + (size_t)k.R * sizeof(uint32_t));
  MRX_REQUIRE(ctx, d_work && (reinterpret_cast<uintptr_t>(d_work) & 15u) == 0 && work_bytes >= col,
              "work buffer: 16-byte aligned, at least mrx_bin_map_work_bytes' minimum");
  const int cols_total = mrx_ceil_div(T, q.tile_samples);
  const size_t usable = work_bytes - 16;  // the table behind the entries is moved up to a 16-byte boundary
  int cols = (int)(usable / col < (size_t)cols_total ? usable / col : (size_t)cols_total);
  if (cols < 1) cols = 1;  // (work_bytes >= the 16-byte-entry minimum: one column of 8- or 12-byte entries fits)
  // pass B indexes the entries of a chunk with 32 bits
  while ((long long)cols * tiles_y * q.tile_entries > (1LL << 32) - 1) cols = (cols + 1) / 2;
  const size_t lds_b = (size_t)g.S * 2 * kBinRegionPx * sizeof(double);
  const bool chain = ctx->options[MRX_OPT_POINTING_CHAIN] != 0;
  typedef void (*BucketKernel)(MapArgs, BinArgs, BucketArgs);
  typedef void (*AccKernel)(MapArgs, BinArgs, BucketArgs, int);
  const bool wts = d_weight != nullptr;
  const BucketKernel pass_a = bil ? (chain ? bin_bucket_kernel<true, true, true> : bin_bucket_kernel<false, true, true>)
                              : wts ? (chain ? bin_bucket_kernel<true, false, true> : bin_bucket_kernel<false, false, true>)
                                    : (chain ? bin_bucket_kernel<true, false, false> : bin_bucket_kernel<false, false, false>);
  const AccKernel pass_b = bil ? bin_accumulate_kernel<16> : wts ? bin_accumulate_kernel<12> : bin_accumulate_kernel<8>;
  MRX_LDS_CAP(ctx, pass_a, q.lds_a);
  MRX_LDS_CAP(ctx, pass_b, lds_b);
  // enough workgroups per region to fill the chip: the regions under the scan hold most samples
  // (round 4, onto 1024^2, regions in dispatch order: 8192 items 28.4 ms, 16384 26.8, 32768 25.6, 65536 25.3, 131072 26.2 for
  //  the call; heaviest regions first: 23.6 / 22.5 / 22.6 / 22.3 / 23.1 from 16384 to 262144)
  if (!ctx->d_bin_order) MRX_HIP(ctx, hipMalloc(&ctx->d_bin_order, sizeof(uint32_t) * 2 * kBinMaxRegions));
  k.totals = ctx->d_bin_order;
  k.order = reinterpret_cast<const int*>(ctx->d_bin_order + kBinMaxRegions);
  int splits = 65536 / k.R;
  splits = splits < 1 ? 1 : splits;
  for (int c0 = 0; c0 < cols_total; c0 += cols) {
    const int nc = cols_total - c0 < cols ? cols_total - c0 : cols;
    k.tiles_x = nc;
    k.n_tiles = nc * tiles_y;
    k.s0 = c0 * q.tile_samples;
    k.s1 = (long long)(c0 + nc) * q.tile_samples < (long long)T ? (c0 + nc) * q.tile_samples : T;
    k.entries = d_work;
    // (the table behind the entries, 16-byte aligned whatever the entry size)
    k.tab = reinterpret_cast<uint32_t*>(static_cast<char*>(d_work) + (((size_t)k.n_tiles * q.tile_entries * entry_bytes + 15) & ~(size_t)15));
    MRX_HIP(ctx, hipMemsetAsync(k.tab, 0, (size_t)k.R * k.n_tiles * sizeof(uint32_t), ctx->stream));
    MRX_HIP(ctx, hipMemsetAsync(k.totals, 0, (size_t)k.R * sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(pass_a, dim3(nc, tiles_y), dim3(kBlock), q.lds_a, ctx->stream, g, b, k);
    hipLaunchKernelGGL(bin_order_kernel, dim3(1), dim3(1024), 0, ctx->stream, k.totals, k.R, const_cast<int*>(k.order));
    const int sp = splits < k.n_tiles ? splits : k.n_tiles;
    hipLaunchKernelGGL(pass_b, dim3(k.R, sp), dim3(kBlock), lds_b, ctx->stream, g, b, k, sp);
    MRX_CHECK_LAUNCH(ctx);
  }
  return MRX_OK;
}

}  // extern "C"
