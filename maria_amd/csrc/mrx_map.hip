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
// last thread).  The kernel is bound by the float32 trigonometry of steps 1-3
// (~15 transcendental calls per sample), not by memory.
#include "mrx_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTileDet = 16;
constexpr int kSamplesPerThread = 4;
constexpr int kTileSamples = kBlock * kSamplesPerThread;
constexpr int kMaxStokes = 4;
constexpr float kHalfPiF = 1.57079637050628662109375f;
constexpr float kTwoPiF = 6.283185482025146484375f;

typedef float vfloat4 __attribute__((ext_vector_type(4)));

struct MapArgs {
  const float* values;  // [C][S][n_eta][n_xi]
  const double* eta;    // [n_eta]
  const double* xi;     // [n_xi]
  int C, S, n_eta, n_xi;
  float cphi;           // map centre longitude (float32, as jax demotes it)
  float rot_re, rot_im; // exp(i (pi/2 - ctheta)) in complex64
  int bilinear;
  // calibration
  const float* cal;       // [C][n_pwv][n_el] or null
  const float* cal_pwv;   // [n_pwv]
  const float* cal_el;    // [n_el]
  int n_pwv, n_el;
  const double* pwv;      // [Ta][D] coarse zenith-scaled pwv
  int Ta;
  double ta0, dta;
  const double* t;        // [T]
  const double* scalar;   // [C] (no atmosphere)
  // pointing
  const float* az;        // [T]
  const float* el;        // [T]
  const double* transform;  // [T][9] or null
  const float* dx;
  const float* dy;
  const float* stokes_w;  // [D][S]
  int D, T;
  float* out;
  size_t ld;
  int vec_ok;
};

struct DetConst {
  float c_re, c_cr, c_im;  // sin(r)cos(p), cos(r), sin(r)sin(p)
  float w[kMaxStokes];
};

// np.digitize(x, side) for a monotonic axis (either direction), right = False:
// ascending: side[b-1] <= x < side[b]; descending: side[b-1] > x >= side[b].
__device__ __forceinline__ int digitize(const double* side, int n, double x, double first, double inv_step, bool ascending) {
  int b = (int)fmin(fmax((x - first) * inv_step + 1.0, 0.0), (double)n);
  if (ascending) {
    while (b < n && side[b] <= x) ++b;
    while (b > 0 && side[b - 1] > x) --b;
  } else {
    while (b < n && side[b] > x) ++b;
    while (b > 0 && side[b - 1] <= x) --b;
  }
  return b;
}

// one axis of utils/linalg.py:25-41: the two pixels and the weight of the upper one
__device__ __forceinline__ void axis_weights(const double* side, int n, double x, bool bilinear, int& i0, int& i1, double& p) {
  const bool ascending = side[n - 1] >= side[0];
  if (bilinear) {
    const double inv = (double)(n - 1) / (side[n - 1] - side[0]);
    const int b = digitize(side, n, x, side[0], inv, ascending);
    if (b == 0 || b == n) {
      p = 0.0;  // (x + inf)/inf = nan -> 0; finite/inf = 0
    } else {
      p = (x - side[b - 1]) / (side[b] - side[b - 1]);
      p = p > 0.0 ? p : 0.0;
    }
    i0 = min(max(b - 1, 0), n - 1);
    i1 = min(b, n - 1);
  } else {
    // np.digitize on the midpoints
    const double inv = (double)(n - 1) / (side[n - 1] - side[0]);
    int b = (int)fmin(fmax((x - side[0]) * inv + 0.5, 0.0), (double)(n - 1));
    if (ascending) {
      while (b < n - 1 && 0.5 * (side[b] + side[b + 1]) <= x) ++b;
      while (b > 0 && 0.5 * (side[b - 1] + side[b]) > x) --b;
    } else {
      while (b < n - 1 && 0.5 * (side[b] + side[b + 1]) > x) ++b;
      while (b > 0 && 0.5 * (side[b - 1] + side[b]) <= x) --b;
    }
    i0 = i1 = b;
    p = 0.0;
  }
}

// jax RegularGridInterpolator index and weight on a float32 axis (searchsorted left)
__device__ __forceinline__ void rgi_axis(const float* g, int n, float x, int& i, float& w, bool& oob) {
  const float inv = (float)(n - 1) / (g[n - 1] - g[0]);
  int k = min(max((int)fminf(fmaxf((x - g[0]) * inv, -1.0f), 2.0e9f), 0), n - 2);
  while (k < n - 2 && g[k + 1] < x) ++k;
  while (k > 0 && g[k] >= x) --k;
  i = k;
  w = __fdiv_rn(__fsub_rn(x, g[k]), __fsub_rn(g[k + 1], g[k]));
  oob = !(x >= g[0] && x <= g[n - 1]);
}

// raw (unconvolved) map loading of detector `dc` (row d) at sample s, float32
__device__ __forceinline__ float raw_sample(const MapArgs& g, const DetConst& dc, int d, int s) {
  s = min(max(s, 0), g.T - 1);
  // 1. detector az/el (transforms.py:10-29)
  const float a = __fsub_rn(g.el[s], kHalfPiF);
  const float ca = cosf(a), sa = sinf(a);
  const float re = __fsub_rn(__fmul_rn(dc.c_re, ca), __fmul_rn(dc.c_cr, sa));
  const float im = __fadd_rn(__fmul_rn(dc.c_re, sa), __fmul_rn(dc.c_cr, ca));
  const float az_d = __fadd_rn(atan2f(dc.c_im, re), g.az[s]);
  const float el_d = asinf(im);
  // 2. frame rotation (coordinates.py:220-230)
  float phi = az_d, theta = el_d;
  if (g.transform) {
    const float ce = cosf(el_d);
    const double x = (double)__fmul_rn(cosf(az_d), ce), y = (double)__fmul_rn(sinf(az_d), ce), z = (double)sinf(el_d);
    const double* M = g.transform + (size_t)s * 9;
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
  const float ox = -__fmul_rn(dz_re, f), oy = -__fmul_rn(dz_im, f);
  // 4. pointing-matrix row
  int e0, e1, x0, x1;
  double pe, px;
  axis_weights(g.eta, g.n_eta, (double)oy, g.bilinear, e0, e1, pe);
  axis_weights(g.xi, g.n_xi, (double)ox, g.bilinear, x0, x1, px);
  const double w00 = (1.0 - pe) * (1.0 - px), w10 = pe * (1.0 - px), w01 = (1.0 - pe) * px, w11 = pe * px;
  // 5. channels
  float cal_w_p = 0.f, cal_w_e = 0.f;
  int ip = 0, ie = 0;
  bool oob = false;
  if (g.cal) {
    // zenith-scaled pwv of the sample: linear interpolation of the coarse series
    // (sim/atmosphere.py:30-37), demoted to float32 by the jax interpolator
    const double tt = g.t[s];
    const double inv_dta = 1.0 / g.dta;
    int jj = (int)floor(fmin(fmax((tt - g.ta0) * inv_dta, -1.0), 2.0e9));
    jj = min(max(jj, 0), g.Ta - 2);
    const double u = (tt - (g.ta0 + (double)jj * g.dta)) * inv_dta;
    const double y0 = g.pwv[(size_t)jj * g.D + d], y1 = g.pwv[(size_t)(jj + 1) * g.D + d];
    const float pw = (float)(y0 + u * (y1 - y0));
    bool o1, o2;
    rgi_axis(g.cal_pwv, g.n_pwv, pw, ip, cal_w_p, o1);
    rgi_axis(g.cal_el, g.n_el, el_d, ie, cal_w_e, o2);
    oob = o1 || o2;
  }
  const size_t plane = (size_t)g.n_eta * g.n_xi;
  float acc = 0.0f;
  for (int c = 0; c < g.C; ++c) {
    double val = 0.0;
    for (int k = 0; k < g.S; ++k) {
      const float* m = g.values + ((size_t)c * g.S + k) * plane;
      const double v = w00 * (double)m[(size_t)e0 * g.n_xi + x0] + w10 * (double)m[(size_t)e1 * g.n_xi + x0] +
                       w01 * (double)m[(size_t)e0 * g.n_xi + x1] + w11 * (double)m[(size_t)e1 * g.n_xi + x1];
      val += (double)dc.w[k] * v;
    }
    double pw_per_k;
    if (g.cal) {
      const float* tab = g.cal + (size_t)c * g.n_pwv * g.n_el;
      // float32 corner sum in product order, weights built as (1 * w_pwv) * w_el
      const float wp0 = __fsub_rn(1.0f, cal_w_p), we0 = __fsub_rn(1.0f, cal_w_e);
      float v = __fmul_rn(tab[(size_t)ip * g.n_el + ie], __fmul_rn(wp0, we0));
      v = __fadd_rn(v, __fmul_rn(tab[(size_t)ip * g.n_el + ie + 1], __fmul_rn(wp0, cal_w_e)));
      v = __fadd_rn(v, __fmul_rn(tab[(size_t)(ip + 1) * g.n_el + ie], __fmul_rn(cal_w_p, we0)));
      v = __fadd_rn(v, __fmul_rn(tab[(size_t)(ip + 1) * g.n_el + ie + 1], __fmul_rn(cal_w_p, cal_w_e)));
      if (oob) v = __builtin_nanf("");
      pw_per_k = (double)__fmul_rn(1.380649e-11f, v);  // 1e12 k_B as a weak scalar on a float32 array
    } else {
      pw_per_k = 1.380649e-11 * g.scalar[c];
    }
    acc = (float)((double)acc + pw_per_k * val);  // float32 accumulator (map.py:155)
  }
  return acc;
}

__global__ __launch_bounds__(kBlock) void map_sample_kernel(MapArgs g) {
  __shared__ DetConst dets[kTileDet];
  __shared__ float2 edge[2][kBlock];  // (first, last) raw value of every thread, double-buffered
  const int d0 = blockIdx.y * kTileDet;
  const int s_tile = blockIdx.x * kTileSamples;
  const int sb = s_tile + threadIdx.x * kSamplesPerThread;
  const int nd = min(kTileDet, g.D - d0);
  if ((int)threadIdx.x < nd) {
    const int d = d0 + threadIdx.x;
    const float dx = g.dx[d], dy = g.dy[d];
    const float r = sqrtf(dx * dx + dy * dy);
    const float p = atan2f(-dx, -dy);
    const float sr = sinf(r);
    DetConst dc;
    dc.c_re = __fmul_rn(sr, cosf(p));
    dc.c_cr = cosf(r);
    dc.c_im = __fmul_rn(sr, sinf(p));
    for (int k = 0; k < kMaxStokes; ++k) dc.w[k] = k < g.S ? g.stokes_w[(size_t)d * g.S + k] : 0.0f;
    dets[threadIdx.x] = dc;
  }
  __syncthreads();
  const bool full = (sb + kSamplesPerThread <= g.T) && g.vec_ok;
  for (int dl = 0; dl < nd; ++dl) {
    const DetConst dc = dets[dl];
    const int d = d0 + dl;
    float r[kSamplesPerThread];
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) r[q] = raw_sample(g, dc, d, sb + q);
    float halo = 0.0f;
    if (threadIdx.x == 0) halo = raw_sample(g, dc, d, s_tile - 1);
    if (threadIdx.x == kBlock - 1) halo = raw_sample(g, dc, d, s_tile + kTileSamples);
    edge[dl & 1][threadIdx.x] = make_float2(r[0], r[kSamplesPerThread - 1]);
    __syncthreads();
    const float left = threadIdx.x == 0 ? halo : edge[dl & 1][threadIdx.x - 1].y;
    const float right = threadIdx.x == kBlock - 1 ? halo : edge[dl & 1][threadIdx.x + 1].x;
    // scipy.ndimage.convolve1d, symmetric 3-tap kernel, double accumulation (map.py:170)
    float o[kSamplesPerThread];
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) {
      const double prev = (double)(q == 0 ? left : r[q - 1]);
      const double next = (double)(q == kSamplesPerThread - 1 ? right : r[q + 1]);
      o[q] = (float)((double)r[q] * 0.5 + (prev + next) * 0.25);
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

}  // namespace

extern "C" {

int mrx_map_sample(mrx_ctx* ctx, const mrx_sky_map* map, const mrx_map_cal* cal,
                   const float* d_az, const float* d_el, int T, const double* d_transform,
                   const float* d_dx, const float* d_dy, const float* d_stokes_w, int D,
                   float* d_out, size_t ld_out) {
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, map && cal && d_az && d_el && d_dx && d_dy && d_stokes_w && d_out, "null pointer");
  MRX_REQUIRE(ctx, map->d_values && map->d_eta && map->d_xi, "null map pointer");
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
  g.eta = map->d_eta;
  g.xi = map->d_xi;
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
  g.bilinear = map->bilinear;
  g.cal = cal->d_table;
  g.cal_pwv = cal->d_axis_pwv;
  g.cal_el = cal->d_axis_el;
  g.n_pwv = cal->n_pwv;
  g.n_el = cal->n_el;
  g.pwv = cal->d_pwv;
  g.Ta = cal->Ta;
  g.ta0 = cal->ta0;
  g.dta = cal->dta;
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
  dim3 grid(mrx_ceil_div(T, kTileSamples), mrx_ceil_div(D, kTileDet));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  hipLaunchKernelGGL(map_sample_kernel, grid, dim3(kBlock), 0, ctx->stream, g);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

}  // extern "C"
