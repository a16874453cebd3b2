// Turbulent screen generator for gfx950: Philox-4x32-10 normals on the Hermitian
// half of k space x sqrt(von Karman / Matern PSD) -> 2-D complex-to-real inverse FFT,
// with the Gaussian beam smoothing folded into the two FFT passes.
//
// Replaces the reference's autoregressive generator
// (atmosphere/process.py:111-209) by a spectral one with the same target
// covariance, Matern(nu, r0) (functions/__init__.py:30-39): in two dimensions
// its spectrum is PSD(k) ~ (k0^2 + |k|^2)^-(nu+1), k0 = sqrt(2 nu)/r0.
//
// A real field needs only the columns kx = 0 .. nx/2 of its spectrum H
// (H[-ky][-kx] = conj H[ky][kx]); the two passes (ny rows, nx columns, powers of two):
//   1. one workgroup per (kx, layer): draw H[.][kx] straight into LDS (interior
//      columns: independent complex normals / sqrt 2; the columns kx = 0 and nx/2 are
//      Hermitian in ky), inverse FFT along y, optional Gaussian along y (linear in y,
//      so it commutes with the x transform still to come)   -> G[kx][y]
//   2. one workgroup per (block of 2^LJ rows, layer): gathers the rows' half spectra
//      (the only transposed access: 8 << LJ bytes per kx, neighbouring row blocks on the
//      same XCD so that they share the fetched lines in its L2), folds them into an
//      nx/2-point complex sequence, one batched inverse FFT gives (even, odd) samples,
//      optional Gaussian along x on the real row in LDS, scale to unit variance
//                                                           -> out[y][x]
// Per pixel: 4 B written + 4 B read (G) + 4 B written (screen) against 53 B for the
// full-spectrum form with its transpose and the two separate stencil passes.
// The random number of spectrum cell (kx, ky) depends only on (seed, stream,
// kx, ky), so every GPU regenerates bit-identical screens with no broadcast.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#include "mrx_internal.h"

#include "mrx_spectral.h"

namespace {

using namespace mrx_dev;

constexpr int kMaxBatch = 16;         // layers per launch
// G[kx][.] rows are ny + kPitchPad complex apart: pass 2 walks kx at fixed y, and a power-of-two
// stride of 8 ny bytes would put every access of that walk on the same memory channel
constexpr int kPitchPad = 16;
#ifndef MRX_SCREEN_IMAGES
#define MRX_SCREEN_IMAGES 1  // exchange images of the register transforms in the two generator passes (fft_regs: 2 = the round-3 form, A/B)
#endif
constexpr int kScreenImages = MRX_SCREEN_IMAGES;
constexpr int kMaxFusedRadius = 128;  // Gaussian taps kept in LDS by the fused passes

// sqrt of the von Karman / Matern spectrum: (k0^2 + |k|^2)^expo via exp2/log2
__device__ __forceinline__ float spectrum_amp(double k2, float expo) {
  return __builtin_amdgcn_exp2f(expo * __builtin_amdgcn_logf((float)k2));
}

__device__ __forceinline__ double wavenumber(int i, int n, double d) {
  const int s = i < (n + 1) / 2 ? i : i - n;  // numpy.fft.fftfreq ordering
  return 6.283185307179586476925 * (double)s / ((double)n * d);
}

struct ScreenLayerArgs {
  float2* work;           // [nx/2 + 1][ny]
  float* out;             // [out_ny][ld_out]
  const double* psd_sum;  // device scalar: sum of the PSD over the grid
  const float* taps_y;    // [kMaxFusedRadius + 1] normalised Gaussian taps, zero beyond the radius
  const float* taps_x;
  const float* amp;       // [nx/2 + 1][ny/2 + 1] amplitudes of mrx_screen_amplitudes (even in ky), or null: the power law
  const float* resp_y;    // [ny/2 + 1] transfer function of the beam's taps along y on the periodic domain, or null
  const float* resp_x;    // [nx/2 + 1] (periodic_beam: the beam as a factor of the spectrum; ry = rx = 0 then)
  int from_work;          // 1: the half spectrum of this plane already sits in `work` (3-D generator,
                          // written by screen3d_fft_h); pass 1 transforms it in place instead of drawing
  double dy, dx, k0sq;
  unsigned long long ld_out;
  float expo;
  int ry, rx;                    // stencil radii, 0 = no smoothing along that axis
  int out_ny, out_nx;            // the written top-left block of the periodic domain
  uint32_t stream;
};

struct ScreenBatchArgs {
  ScreenLayerArgs l[kMaxBatch];
};

// scipy.ndimage "reflect" (d c b a | a b c d | d c b a), any distance
__device__ __forceinline__ int reflect_index(int i, int n) {
  if (i < 0) i = -i - 1;
  if (i >= n) i = 2 * n - 1 - i;
  if ((unsigned)i >= (unsigned)n) {  // a radius beyond the line's length: several reflections
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    i = i < n ? i : period - 1 - i;
  }
  return i;
}

// Gaussian stencil along a line held in LDS, scipy.ndimage semantics (reflect at the ends of
// the n written samples).  Radii up to kFastRadius (every beam the reference's grid rule allows:
// res >= fwhm/10, atmosphere/extrusion.py:56-60, gives sigma <= 4.25 pixels, radius 17) take
// the register-window form: a thread owns a few consecutive outputs, pulls their whole
// neighbourhood with 16-byte LDS reads and runs the symmetric taps (wave-uniform, from
// scalar loads) over registers -- 3x fewer LDS bytes than one read per tap.  Threads whose
// window would cross an end, and wider beams, take the plain loop.
constexpr int kFastRadius = 20;

// the first kFastRadius + 1 taps as wave-uniform values (SGPRs)
__device__ __forceinline__ void uniform_taps(const float* __restrict__ taps, float (&tp)[kFastRadius + 1]) {
#pragma unroll
  for (int k = 0; k <= kFastRadius; ++k)
    tp[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, taps[k])));
}

__device__ __forceinline__ float smooth_real_at(const float* row, int x, int n, int r,
                                                const float* __restrict__ taps) {
  float acc = taps[0] * row[x];
  for (int k = 1; k <= r; ++k)
    acc = fmaf(taps[k], row[reflect_index(x - k, n)] + row[reflect_index(x + k, n)], acc);
  return acc;
}

__device__ __forceinline__ float2 smooth_complex_at(const float2* col, int y, int n, int r,
                                                    const float* __restrict__ taps) {
  float2 acc = col[y];
  acc.x *= taps[0];
  acc.y *= taps[0];
  for (int k = 1; k <= r; ++k) {
    const float2 a = col[reflect_index(y - k, n)], b = col[reflect_index(y + k, n)];
    acc.x = fmaf(taps[k], a.x + b.x, acc.x);
    acc.y = fmaf(taps[k], a.y + b.y, acc.y);
  }
  return acc;
}

// Epilogue of pass 1 for one column: the transformed column `res` (LDS, natural order) through the
// Gaussian along y (scipy.ndimage.gaussian_filter along axis 0 of the written block, reflect at its
// edges) into G[kx][.]; `nt` threads (t = 0 .. nt - 1) work on the column.
__device__ __forceinline__ void smooth_column_store(const ScreenLayerArgs& L, const float2* res, float2* dst, int t, int nt) {
  if (L.ry > 0) {
    const int r = L.ry, n = L.out_ny;
    const float* __restrict__ taps = L.taps_y;
    if (r <= kFastRadius) {
      constexpr int R = kFastRadius;
      float tp[R + 1];
      uniform_taps(taps, tp);
      for (int y0 = 2 * t; y0 < n; y0 += 2 * nt) {
        float2 win[2 + 2 * R];
        if (y0 >= R && y0 + 2 + R <= n) {
          const float4* wp = reinterpret_cast<const float4*>(res + (y0 - R));
#pragma unroll
          for (int j = 0; j < 1 + R; ++j) {
            const float4 q = wp[j];
            win[2 * j] = make_float2(q.x, q.y);
            win[2 * j + 1] = make_float2(q.z, q.w);
          }
        } else {  // the window crosses an end of the column: gather it through the reflection
#pragma unroll
          for (int j = 0; j < 2 + 2 * R; ++j) win[j] = res[reflect_index(y0 - R + j, n)];
        }
        float2 acc[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[v] = make_float2(tp[0] * win[R + v].x, tp[0] * win[R + v].y);
#pragma unroll
        for (int k = 1; k <= R; ++k) {
          const float w = tp[k];
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            acc[v].x = fmaf(w, win[R + v - k].x + win[R + v + k].x, acc[v].x);
            acc[v].y = fmaf(w, win[R + v - k].y + win[R + v + k].y, acc[v].y);
          }
        }
        if (y0 + 2 <= n) {
          *reinterpret_cast<float4*>(dst + y0) = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y);
        } else {
          dst[y0] = acc[0];
        }
      }
    } else {
      for (int y = t; y < n; y += nt) dst[y] = smooth_complex_at(res, y, n, r, taps);
    }
  } else {
    for (int y = t; y < L.out_ny; y += nt) dst[y] = res[y];
  }
}

// pass 1 for one column by the whole workgroup: spectrum column kx = ix of one layer, all ky, into
// LDS; Stockham inverse FFT along y; Gaussian along y.  lds2: 2 ny + ny / 4 float2.
__device__ __forceinline__ void half_spectrum_column_stockham(const ScreenLayerArgs& L, float2* lds2, int ix, int ny, int nx,
                                                              int log2ny, uint32_t key0, uint32_t key1) {
  float2* data = lds2;
  float2* tw = lds2 + 2 * ny;
  const double kx = wavenumber(ix, nx, L.dx);
  const bool edge = ix == 0 || 2 * ix == nx;
  fill_twiddles(tw, ny);
  // one Philox call feeds two cells: words (x, y) -> ky index iy, (z, w) -> iy + ny/2
  const int half = ny >> 1;
  constexpr float kRoot = 0.70710678118654752f;
  if (L.from_work) {
    // 3-D generator: the column of this height plane's half spectrum was produced by the
    // transform along h.  Its cells are independent complex Gaussians; on the two self-mirrored
    // columns the Hermitian symmetry in ky is imposed here: S' = (S[ky] + conj S[-ky]) / sqrt 2
    // keeps the (real) covariance along h and makes S'[-ky] = conj S'[ky]; the self-conjugate
    // cells become sqrt 2 Re S.
    const float2* src = L.work + (size_t)ix * (ny + kPitchPad);
    const float hx = L.resp_x ? L.resp_x[ix] : 1.0f;
    for (int iy = threadIdx.x; iy < half; iy += kBlock) {
      float2 a = src[iy], b = src[iy + half];
      if (L.resp_y) {  // the beam as a factor of the spectrum (even in ky: cell iy + ny/2 mirrors ny/2 - iy)
        const float h0 = hx * L.resp_y[iy], h1 = hx * L.resp_y[half - iy];
        a.x *= h0; a.y *= h0; b.x *= h1; b.y *= h1;
      }
      if (!edge) {
        data[iy] = a;
        data[iy + half] = b;
      } else if (iy == 0) {
        data[0] = make_float2(1.41421356f * a.x, 0.0f);
        data[half] = make_float2(1.41421356f * b.x, 0.0f);
      } else {
        float2 m = src[ny - iy];
        if (L.resp_y) { m.x *= hx * L.resp_y[iy]; m.y *= hx * L.resp_y[iy]; }
        const float2 h = make_float2(kRoot * (a.x + m.x), kRoot * (a.y - m.y));
        data[iy] = h;
        data[ny - iy] = make_float2(h.x, -h.y);
      }
    }
  } else
  for (int iy = threadIdx.x; iy < half; iy += kBlock) {
    const U4 rnd = philox4x32_10(U4{(uint32_t)ix, (uint32_t)iy, L.stream, 0u}, key0, key1);
    float amp0, amp1;
    if (L.amp) {  // tabulated, even in ky: cell iy + ny/2 is the mirror of ny/2 - iy
      const float* A = L.amp + (size_t)ix * (half + 1);
      amp0 = A[iy];
      amp1 = A[half - iy];
    } else {
      const double ky0 = wavenumber(iy, ny, L.dy), ky1 = wavenumber(iy + half, ny, L.dy);
      amp0 = spectrum_amp(L.k0sq + kx * kx + ky0 * ky0, L.expo);
      amp1 = spectrum_amp(L.k0sq + kx * kx + ky1 * ky1, L.expo);
    }
    if (L.resp_y) {
      const float hx = L.resp_x[ix];
      amp0 *= hx * L.resp_y[iy];
      amp1 *= hx * L.resp_y[half - iy];
    }
    const float2 g0 = box_muller(rnd.x, rnd.y), g1 = box_muller(rnd.z, rnd.w);
    if (!edge) {
      data[iy] = make_float2(kRoot * amp0 * g0.x, kRoot * amp0 * g0.y);
      data[iy + half] = make_float2(kRoot * amp1 * g1.x, kRoot * amp1 * g1.y);
    } else if (iy == 0) {
      // the four self-conjugate cells are real
      data[0] = make_float2(amp0 * g0.x, 0.0f);
      data[half] = make_float2(amp1 * g1.x, 0.0f);
    } else {
      // kx = 0 and kx = nx/2 are their own mirror columns: Hermitian in ky
      const float2 h = make_float2(kRoot * amp0 * g0.x, kRoot * amp0 * g0.y);
      data[iy] = h;
      data[ny - iy] = make_float2(h.x, -h.y);
    }
  }
  __syncthreads();
  const float2* res = fft_lds_inverse(data, data + ny, tw, ny, log2ny);
  smooth_column_store(L, res, L.work + (size_t)ix * (ny + kPitchPad), threadIdx.x, kBlock);
}

// pass 1, every column (0 .. nx/2) through the Stockham transform: one workgroup per (kx, layer)
// (col_step: 1, or nx/2 with a grid of two -- the two self-mirrored columns beside the register form, in a launch of their
// own so that the register form's workgroups ask for their own LDS only)
__global__ __launch_bounds__(kBlock) void screen_half_spectrum_fft_y(
    const ScreenBatchArgs args, int ny, int nx, int log2ny, uint32_t key0, uint32_t key1, int col_step) {
  extern __shared__ __align__(16) float2 lds2[];
  half_spectrum_column_stockham(args.l[blockIdx.y], lds2, blockIdx.x * col_step, ny, nx, log2ny, key0, key1);
}

// pass 1 with the transform in registers (fft_regs: ny = 256 RB, 16 RB threads a column, 16 / RB
// columns a workgroup) for the interior columns 0 < kx < nx/2: a thread draws its 16 cells
// ky = t + T b straight into the registers the first radix-16 pass reads -- the same Philox
// cells, the same values as the Stockham kernel to rounding.  The two self-mirrored columns
// (Hermitian in ky: a cell's mirror belongs to another thread) take the Stockham form in the
// launch's last two workgroups (the LDS of the register form covers its 2 ny + ny/4 values).
template <int RB>
__global__ __launch_bounds__(kBlock) void screen_half_spectrum_regs(const ScreenBatchArgs args, int nx, uint32_t key0,
                                                                    uint32_t key1, int edges_inside) {
  extern __shared__ __align__(16) float2 lds2[];
  constexpr int T = 16 * RB, ny = 256 * RB, half = ny / 2, kCols = kBlock / T;
  const ScreenLayerArgs& L = args.l[blockIdx.y];
  // (edges_inside: the launch's last two workgroups take kx = 0 and kx = nx/2 in the Stockham form -- where that form's
  // LDS is no more than the register form's; otherwise they come in a launch of their own: mrx_screen_generate_batch)
  if (edges_inside && blockIdx.x + 2 >= gridDim.x) {  // (uniform)
    half_spectrum_column_stockham(L, lds2, blockIdx.x + 1 == gridDim.x ? nx / 2 : 0, ny, nx, 8 + (RB == 4 ? 2 : RB == 8 ? 3 : 4),
                                  key0, key1);
    return;
  }
  const int which = threadIdx.x / T, t = threadIdx.x % T;
  const int ix = 1 + blockIdx.x * kCols + which;
  const bool live = ix < nx / 2;  // uniform over the column's waves
  float2* ex1 = lds2 + (size_t)which * kScreenImages * kFft4096Pitch * T;
  float2* ex2 = ex1 + (kScreenImages - 1) * kFft4096Pitch * T;
  float2 v[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) v[b] = make_float2(0.f, 0.f);
  if (live) {
    if (L.from_work) {
      const float2* src = L.work + (size_t)ix * (ny + kPitchPad);
#pragma unroll
      for (int b = 0; b < 16; ++b) v[b] = src[t + T * b];
      if (L.resp_y) {  // the beam as a factor of the spectrum: |ky| = iy for b < 8, ny/2 - iy for the mirrored half
        const float hx = L.resp_x[ix];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const float h0 = hx * L.resp_y[t + T * b], h1 = hx * L.resp_y[half - (t + T * b)];
          v[b].x *= h0; v[b].y *= h0; v[b + 8].x *= h1; v[b + 8].y *= h1;
        }
      }
    } else {
      constexpr float kRoot = 0.70710678118654752f;
      const double kx = wavenumber(ix, nx, L.dx);
      const float hx = L.resp_x ? L.resp_x[ix] : 1.0f;
      auto draw = [&](auto tabulated) {
        float a0[8], a1[8];
        if constexpr (decltype(tabulated)::value) {  // all 16 loads in flight before the first draw
          const float* A = L.amp + (size_t)ix * (half + 1);
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            a0[b] = A[t + T * b];
            a1[b] = A[half - (t + T * b)];
          }
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {  // one Philox call feeds two cells: ky index iy and iy + ny/2 (T * 8 = ny/2)
          const int iy = t + T * b;
          const U4 rnd = philox4x32_10(U4{(uint32_t)ix, (uint32_t)iy, L.stream, 0u}, key0, key1);
          float amp0, amp1;
          if constexpr (decltype(tabulated)::value) {
            amp0 = a0[b];
            amp1 = a1[b];
          } else {
            const double ky0 = wavenumber(iy, ny, L.dy), ky1 = wavenumber(iy + half, ny, L.dy);
            amp0 = spectrum_amp(L.k0sq + kx * kx + ky0 * ky0, L.expo);
            amp1 = spectrum_amp(L.k0sq + kx * kx + ky1 * ky1, L.expo);
          }
          if (L.resp_y) {  // (uniform)
            amp0 *= hx * L.resp_y[iy];
            amp1 *= hx * L.resp_y[half - iy];
          }
          const float2 g0 = box_muller(rnd.x, rnd.y), g1 = box_muller(rnd.z, rnd.w);
          v[b] = make_float2(kRoot * amp0 * g0.x, kRoot * amp0 * g0.y);
          v[b + 8] = make_float2(kRoot * amp1 * g1.x, kRoot * amp1 * g1.y);
        }
      };
      if (L.amp) draw(std::true_type{}); else draw(std::false_type{});
    }
  }
  fft_regs<RB, kScreenImages == 1>(v, ex1, ex2, t);
  // the column in natural order for the stencil: ex1 is free again (its last readers are behind
  // the transform's second barrier) and holds 17 T >= ny values
#pragma unroll
  for (int f = 0; f < 16; ++f) ex1[t + T * f] = v[dft16_pos(f)];
  __syncthreads();
  if (live) smooth_column_store(L, ex1, L.work + (size_t)ix * (ny + kPitchPad), t, T);
}

// Epilogue of pass 2 for one real row held in LDS (`row`, natural order): Gaussian along x (scipy.ndimage semantics on
// the written block) or nothing (periodic_beam / no beam), the unit-variance scale, 16-byte stores; `nt` threads
// (t = 0 .. nt - 1) work on the row.
__device__ __forceinline__ void finish_row(const ScreenLayerArgs& L, const float* row, int y, int t, int nt) {
  const float norm = (float)(1.0 / sqrt(*L.psd_sum));
  const int r = L.rx, nxo = L.out_nx;
  const float* __restrict__ taps = L.taps_x;
  const bool vec_ok = (L.ld_out & 3) == 0 && (reinterpret_cast<uintptr_t>(L.out) & 15) == 0;
  float* dst = L.out + (size_t)y * L.ld_out;
  if (r > kFastRadius) {
    for (int x = t; x < nxo; x += nt) dst[x] = norm * smooth_real_at(row, x, nxo, r, taps);
    return;
  }
  float tp[kFastRadius + 1] = {0.0f};
  if (r > 0) uniform_taps(taps, tp);
  for (int x0 = 4 * t; x0 < nxo; x0 += 4 * nt) {
    float o[4];
    constexpr int R = kFastRadius;
    if (r == 0) {
      const float4 q = *reinterpret_cast<const float4*>(row + x0);
      o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
    } else {
      float win[4 + 2 * R];
      if (x0 >= R && x0 + 4 + R <= nxo) {
        const float4* wp = reinterpret_cast<const float4*>(row + (x0 - R));
#pragma unroll
        for (int j = 0; j < 1 + R / 2; ++j) {
          const float4 q = wp[j];
          win[4 * j] = q.x; win[4 * j + 1] = q.y; win[4 * j + 2] = q.z; win[4 * j + 3] = q.w;
        }
      } else {  // the window crosses an end of the row: gather it through the reflection
#pragma unroll
        for (int j = 0; j < 4 + 2 * R; ++j) win[j] = row[reflect_index(x0 - R + j, nxo)];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) o[v] = tp[0] * win[R + v];
#pragma unroll
      for (int k = 1; k <= R; ++k) {
        const float w = tp[k];
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v] = fmaf(w, win[R + v - k] + win[R + v + k], o[v]);
      }
    }
    if (vec_ok && x0 + 4 <= nxo) {
      *reinterpret_cast<float4*>(dst + x0) = make_float4(o[0] * norm, o[1] * norm, o[2] * norm, o[3] * norm);
    } else {
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (x0 + v < nxo) dst[x0 + v] = o[v] * norm;
    }
  }
}

// pass 2: 2^LJ rows of one layer: fold the half spectra, batched inverse FFT of length
// nx/2, Gaussian along x, unit-variance scale
template <int LJ>
__global__ __launch_bounds__(kBlock) void screen_c2r_x(const ScreenBatchArgs args, int ny, int nx,
                                                       int log2n2) {
  extern __shared__ __align__(16) float2 lds2[];
  constexpr int B = 1 << LJ;
  const ScreenLayerArgs& L = args.l[blockIdx.y];
  const int n2 = nx >> 1;
  const int cells = n2 << LJ;
  float2* imgA = lds2;
  float2* imgB = lds2 + cells;
  float2* tw_full = lds2 + 2 * cells;  // exp(2 pi i m / nx), m < nx/4
  float2* tw_half = tw_full + nx / 4;  // exp(2 pi i m / n2), m < n2/4
  float2* nyq = tw_half + n2 / 4;      // [B]

  // row blocks that share the 128-byte lines of G (16 consecutive y) run on one XCD:
  // workgroups b and b + 8 share an XCD, so XCD x takes a contiguous range of blocks
  const int nblocks = gridDim.x;
  int yb = blockIdx.x;
  if ((nblocks & 7) == 0) yb = (blockIdx.x & 7) * (nblocks >> 3) + (blockIdx.x >> 3);
  const int y0 = yb << LJ;
  const size_t pitch = (size_t)ny + kPitchPad;

  fill_twiddles(tw_full, nx);
  fill_twiddles(tw_half, n2);
  for (int idx = threadIdx.x; idx < cells; idx += kBlock) {
    const int k = idx >> LJ, y = y0 + (idx & (B - 1));
    imgA[idx] = y < L.out_ny ? L.work[(size_t)k * pitch + y] : make_float2(0.f, 0.f);
  }
  if ((int)threadIdx.x < B) {
    const int y = y0 + threadIdx.x;
    nyq[threadIdx.x] = y < L.out_ny ? L.work[(size_t)n2 * pitch + y] : make_float2(0.f, 0.f);
  }
  __syncthreads();
  // x[2n] + i x[2n+1] = IFFT_{n2}(Z),  Z[k] = (X[k] + conj X[n2-k]) + i w^k (X[k] - conj X[n2-k])
  for (int idx = threadIdx.x; idx < cells; idx += kBlock) {
    const int k = idx >> LJ, b = idx & (B - 1);
    float2 xk = imgA[idx], xm;
    if (k == 0) {
      xk.y = 0.0f;  // G[0][y] and G[nx/2][y] are real up to rounding
      xm = make_float2(nyq[b].x, 0.0f);
    } else {
      xm = imgA[((n2 - k) << LJ) | b];
    }
    const float2 e = make_float2(xk.x + xm.x, xk.y - xm.y);
    const float2 o = cmul(make_float2(xk.x - xm.x, xk.y + xm.y), tw_at(tw_full, k, nx >> 2));
    imgB[idx] = make_float2(e.x - o.y, e.y + o.x);
  }
  __syncthreads();
  float2* res = fft_lds_inverse_batched(imgB, imgA, tw_half, n2, log2n2, LJ);
  // de-interleave into contiguous real rows in the other image (pitch nx floats)
  float* rows = reinterpret_cast<float*>(res == imgA ? imgB : imgA);
  for (int idx = threadIdx.x; idx < cells; idx += kBlock) {
    const int n = idx >> LJ, b = idx & (B - 1);
    *reinterpret_cast<float2*>(rows + (size_t)b * nx + 2 * n) = res[idx];
  }
  __syncthreads();
  for (int b = 0; b < B; ++b) {
    const int y = y0 + b;
    if (y >= L.out_ny) break;
    finish_row(L, rows + (size_t)b * nx, y, threadIdx.x, kBlock);
  }
}

// pass 2 with the row transforms in registers (fft_regs: nx = 512 RB, 16 RB threads a row, 16 / RB rows a workgroup):
// a thread gathers its 16 cells X[k], k = t + T b, of the row's half spectrum, one LDS exchange brings it the mirrored
// cells X[n2 - k], the fold Z[k] = (X[k] + conj X[n2 - k]) + i w^k (X[k] - conj X[n2 - k]) happens in registers (w^k
// from one v_sin / v_cos pair per thread times the sixteen constant 32nd roots), and the n2-point transform runs
// through the two conflict-free exchanges of fft_regs instead of four or five Stockham passes through LDS, whose
// accesses conflicted on a third of their cycles (profiles/r03_kernel_pmc.txt): 15 LDS accesses per cell -> 8.
// Same values as screen_c2r_x to rounding (tests/test_gpu_screens.py::test_register_transforms_match_the_stockham_ones).
template <int RB>
__global__ __launch_bounds__(kBlock) void screen_c2r_regs(const ScreenBatchArgs args, int ny) {
  extern __shared__ __align__(16) float2 lds2[];
  constexpr int T = 16 * RB, n2 = 256 * RB, nx = 2 * n2, G = kBlock / T;
  const ScreenLayerArgs& L = args.l[blockIdx.y];
  const int which = threadIdx.x / T, t = threadIdx.x % T;
  float2* ex1 = lds2 + (size_t)which * kScreenImages * kFft4096Pitch * T;
  float2* ex2 = ex1 + (kScreenImages - 1) * kFft4096Pitch * T;
  // row blocks that share the 128-byte lines of G (16 consecutive y) run on one XCD (see screen_c2r_x)
  const int nblocks = gridDim.x;
  int yb = blockIdx.x;
  if ((nblocks & 7) == 0) yb = (blockIdx.x & 7) * (nblocks >> 3) + (blockIdx.x >> 3);
  const int y = yb * G + which;
  const bool live = y < L.out_ny;  // uniform over the row's waves
  const size_t pitch = (size_t)ny + kPitchPad;
  float2 x[16];
  const float nyq = live ? L.work[(size_t)n2 * pitch + y].x : 0.0f;  // G[nx/2][y] is real up to rounding
  if constexpr (G >= 2) {
    // (rows of 2048 samples and more; at 2048 it cost 8 % while the workgroup held two exchange images, and gains 5 %
    // now that it holds one: four workgroups a CU)
    // The gather is what bounds this pass (the texture addresser 81 % busy at 4096^2, 64 lines a load instruction,
    // profiles/r06_50k_kernel_pmc.txt) and its cost is per instruction and address, not per byte: two neighbouring rows'
    // thread groups share the work -- each lane fetches 16 bytes, ITS cell k of both rows, for every other b -- and hand
    // each other's halves over in the exchange image, where the mirrored cells are read from anyway: half the gathers.
    const int pw = which & 1, y0 = y - pw;  // (G is even and so is y0: 16-byte aligned with the even pitch)
    float2* const ex_even = ex1 - (size_t)pw * kScreenImages * kFft4096Pitch * T;
    float2* const ex_odd = ex_even + kScreenImages * kFft4096Pitch * T;
    const bool pair_live = y0 < L.out_ny;
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
      const int b = 2 * bb + pw;
      const float4 q = pair_live ? *reinterpret_cast<const float4*>(L.work + (size_t)(t + T * b) * pitch + y0) : make_float4(0.f, 0.f, 0.f, 0.f);
      ex_even[t + T * b] = make_float2(q.x, q.y);
      ex_odd[t + T * b] = make_float2(q.z, q.w);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) x[b] = live ? ex1[t + T * b] : make_float2(0.f, 0.f);
  } else {
#pragma unroll
    for (int b = 0; b < 16; ++b) x[b] = live ? L.work[(size_t)(t + T * b) * pitch + y] : make_float2(0.f, 0.f);
#pragma unroll
    for (int b = 0; b < 16; ++b) ex1[t + T * b] = x[b];
    __syncthreads();
  }
  // w^k = exp(2 pi i k / nx), k = t + T b: exp(2 pi i t / nx) exp(2 pi i b / 32)
  const float2 wt = make_float2(__builtin_amdgcn_cosf((float)t * (1.0f / (float)nx)), __builtin_amdgcn_sinf((float)t * (1.0f / (float)nx)));
  constexpr float kC32[16] = {1.0f, 0.98078528040323044f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                              0.55557023301960222f, 0.38268343236508977f, 0.19509032201612827f, 0.0f, -0.19509032201612827f,
                              -0.38268343236508977f, -0.55557023301960222f, -0.70710678118654752f, -0.83146961230254524f,
                              -0.92387953251128674f, -0.98078528040323044f};
  constexpr float kS32[16] = {0.0f, 0.19509032201612827f, 0.38268343236508977f, 0.55557023301960222f, 0.70710678118654752f,
                              0.83146961230254524f, 0.92387953251128674f, 0.98078528040323044f, 1.0f, 0.98078528040323044f,
                              0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960222f,
                              0.38268343236508977f, 0.19509032201612827f};
  float2 v[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const int k = t + T * b;
    float2 xk = x[b], xm;
    if (k == 0) {
      xk.y = 0.0f;
      xm = make_float2(nyq, 0.0f);
    } else {
      xm = ex1[n2 - k];
    }
    const float2 w = cmul(wt, make_float2(kC32[b], kS32[b]));
    const float2 e = make_float2(xk.x + xm.x, xk.y - xm.y);
    const float2 o = cmul(make_float2(xk.x - xm.x, xk.y + xm.y), w);
    v[b] = make_float2(e.x - o.y, e.y + o.x);
  }
  __syncthreads();  // the mirrored cells are read: fft_regs may write ex1
  fft_regs<RB, kScreenImages == 1>(v, ex1, ex2, t);
  // x[2n] + i x[2n+1] = y[n]: the row in natural order (ex1 is free again behind the transform's second barrier)
#pragma unroll
  for (int f = 0; f < 16; ++f) ex1[t + T * f] = v[dft16_pos(f)];
  __syncthreads();
  if (live) finish_row(L, reinterpret_cast<const float*>(ex1), y, t, T);
}

// ---- 3-D generator (model="3d": one process of many layers, vertically correlated) ----
// PSD(k) ~ (k0^2 + |k|^2)^-(nu + 3/2) in three dimensions.  Pass 0: one workgroup per (tile of J
// consecutive ky, kx): draws the nh cells along kz of each (ky, kx) straight into LDS
// (interleaved batch of J sequences), inverse FFT along h, and writes the half spectra of the
// requested height planes -- linear interpolation between the two FFT planes around each plane's
// height, rescaled to keep the variance -- as plane[p][kx][ky], the layout pass 1 reads.
constexpr uint32_t kTag3d = 0x33440000u;  // counter word 3 of the 3-D draws

struct Screen3dPlane {
  float2* work;   // [nx/2 + 1][ny + kPitchPad] half spectrum of this plane
  int h0;         // FFT plane below the layer's height
  float w, scale; // weight of plane h0 + 1; variance correction of the interpolation
};

struct Screen3dArgs {
  const Screen3dPlane* planes;  // device array [n_planes]
  const float* amp;             // [nx/2 + 1][nh/2 + 1][ny/2 + 1] amplitudes (even in kz and ky), or null: the power law
  int n_planes;
  double dh, dy, dx, k0sq;
  float expo;
  uint32_t stream;
};

__global__ __launch_bounds__(kBlock) void screen3d_fft_h(const Screen3dArgs g, int nh, int ny, int nx,
                                                         int log2nh, int lj, uint32_t key0, uint32_t key1) {
  extern __shared__ __align__(16) float2 lds2[];
  const int J = 1 << lj;
  const int cells = nh << lj;
  float2* data = lds2;
  float2* tw = lds2 + 2 * cells;
  const int ix = blockIdx.y;       // 0 .. nx/2
  const int iy0 = blockIdx.x << lj;
  const double kx = wavenumber(ix, nx, g.dx);
  fill_twiddles(tw, nh);
  const int half = nh >> 1;
  constexpr float kRoot = 0.70710678118654752f;
  // one Philox call feeds the cells kz = iz and iz + nh/2 of one (ky, kx)
  for (int e = threadIdx.x; e < (half << lj); e += kBlock) {
    const int iz = e >> lj, b = e & (J - 1);
    const int iy = iy0 + b;
    const U4 rnd = philox4x32_10(U4{(uint32_t)ix, (uint32_t)iy, (g.stream << 16) | (uint32_t)iz, kTag3d}, key0, key1);
    float amp0, amp1;
    if (g.amp) {
      const int my = (ny >> 1) + 1, fy = iy <= (ny >> 1) ? iy : ny - iy;
      const float* A = g.amp + (size_t)ix * (half + 1) * my + fy;
      amp0 = kRoot * A[(size_t)iz * my];
      amp1 = kRoot * A[(size_t)(half - iz) * my];
    } else {
      const double ky = wavenumber(iy, ny, g.dy);
      const double kz0 = wavenumber(iz, nh, g.dh), kz1 = wavenumber(iz + half, nh, g.dh);
      const double kk = g.k0sq + kx * kx + ky * ky;
      amp0 = kRoot * spectrum_amp(kk + kz0 * kz0, g.expo);
      amp1 = kRoot * spectrum_amp(kk + kz1 * kz1, g.expo);
    }
    const float2 g0 = box_muller(rnd.x, rnd.y), g1 = box_muller(rnd.z, rnd.w);
    data[(iz << lj) | b] = make_float2(amp0 * g0.x, amp0 * g0.y);
    data[((iz + half) << lj) | b] = make_float2(amp1 * g1.x, amp1 * g1.y);
  }
  __syncthreads();
  const float2* res = fft_lds_inverse_batched(data, data + cells, tw, nh, log2nh, lj);
  for (int e = threadIdx.x; e < (g.n_planes << lj); e += kBlock) {
    const int p = e >> lj, b = e & (J - 1);
    const Screen3dPlane pl = g.planes[p];
    const float2 a = res[(pl.h0 << lj) | b], c = res[(min(pl.h0 + 1, nh - 1) << lj) | b];
    const float wa = (1.0f - pl.w) * pl.scale, wc = pl.w * pl.scale;
    pl.work[(size_t)ix * (ny + kPitchPad) + iy0 + b] = make_float2(wa * a.x + wc * c.x, wa * a.y + wc * c.y);
  }
}

__global__ __launch_bounds__(kBlock) void psd_sum3d_kernel(double* __restrict__ sum, int nh, int ny, int nx, double dh,
                                                           double dy, double dx, double k0sq, float expo) {
  __shared__ double part[kBlock / 64];
  double acc = 0.0;
  const size_t n = (size_t)nh * ny * nx;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / ((size_t)nx * ny));
    const double kx = wavenumber(ix, nx, dx), ky = wavenumber(iy, ny, dy), kz = wavenumber(iz, nh, dh);
    const float amp = spectrum_amp(k0sq + kx * kx + ky * ky + kz * kz, expo);
    acc += (double)amp * (double)amp;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) s += part[w];
    atomicAdd(sum, s);
  }
}

// sum over the grid of amp^2 (what Var[real part] equals), float64
__global__ __launch_bounds__(kBlock) void psd_sum_kernel(
    double* __restrict__ sum, int ny, int nx, double dy, double dx,
    double k0sq, float expo) {
  __shared__ double part[kBlock / 64];
  double acc = 0.0;
  const size_t n = (size_t)ny * nx;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock) {
    const int iy = (int)(i / nx), ix = (int)(i % nx);
    const double kx = wavenumber(ix, nx, dx), ky = wavenumber(iy, ny, dy);
    // the same float32 amplitude the generator uses
    const float amp = spectrum_amp(k0sq + kx * kx + ky * ky, expo);
    acc += (double)amp * (double)amp;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) s += part[w];
    atomicAdd(sum, s);
  }
}

// ---- covariance-matched amplitudes (circulant embedding) ----
// The power law above is the continuous transform of the Matern covariance cut at the grid's Nyquist
// wavenumber: for rough fields (nu = 1/3 in three dimensions) much of the small-scale variance lies
// beyond it and the structure function at one pixel comes out several times low.  The eigenvalues of
// the covariance itself, sampled on the periodic grid,
//     lambda[k] = sum_d rho_per(d) cos(2 pi k d / N)   per axis, separable,
// carry the aliased power: a field drawn with amplitudes sqrt(lambda) has EXACTLY the covariance
// rho_per between any two pixels.  rho is the exact Matern correlation (functions/__init__.py:30-39)
// from a dense log-log table the host supplies (it needs a Bessel function).  The reference's own
// approximate_normalized_matern (:42-74, a 1024-node log-log table good to 1e-5) will not do here: at
// that accuracy the function is no longer positive definite on the grid, and the eigenvalues clipped
// at zero come back as 50 % too much structure at one pixel.  float64 throughout: the eigenvalues
// span ten decades.  Real and even along every axis, so only the non-negative half of each axis is
// stored and transformed -- by direct cosine sums (a set-up step, run once per geometry).
struct RadialTable {
  const double* log_cov;  // [n] log rho at the nodes
  const double* log_sf;   // [n] log (1 - rho)
  int n;
  double log_first, inv_dlog;  // uniform grid in log(r / r0)
  double r_min, r_max;
};

// four-point Lagrange interpolation on the uniform grid (exact for cubics: with 8192 nodes over nine
// decades the correlation is good to ~1e-11 of its value out to 30 outer scales)
__device__ __forceinline__ double lagrange4(const double* __restrict__ f, int i, double u) {
  const double a = u + 1.0, b = u, c = u - 1.0, d = u - 2.0;
  return f[i - 1] * (-b * c * d * (1.0 / 6.0)) + f[i] * (a * c * d * 0.5) + f[i + 1] * (-a * b * d * 0.5) +
         f[i + 2] * (a * b * c * (1.0 / 6.0));
}

__device__ double radial_correlation(const RadialTable& t, double r_eff) {
  if (r_eff == 0.0) return 1.0;
  const double re = fmax(fabs(r_eff), t.r_min);
  if (!(re < t.r_max)) return 0.0;
  const double u = (log(re) - t.log_first) * t.inv_dlog;
  const int i = min(max((int)u, 1), t.n - 3);
  const double w = u - (double)i;
  const double sf = exp(lagrange4(t.log_sf, i, w));
  const double cov = exp(lagrange4(t.log_cov, i, w));
  const double tt = 1.0 / (1.0 + re * re);  // the complement where it is small, the correlation where it is
  return tt * (1.0 - sf) + (1.0 - tt) * cov;
}

// R[iz][iy][ix] = sum over the periodic images n of rho(|offset (iz, iy, ix) + n L|), 0 <= i <= n/2 per
// axis.  Summing the images (instead of wrapping the distance) keeps the periodic covariance positive
// definite -- its eigenvalues are the continuous spectrum sampled on the reciprocal lattice plus the
// grid's aliases, all >= 0 -- whereas the wrapped distance has a crease at half a period whose
// negative eigenvalues, once clipped, come back as spurious small-scale power.  Images farther than
// x_cut outer scales (rho < 1e-10) are skipped.
struct ImageSum {
  double Lz, Ly, Lx;  // periods (m)
  int iz, iy, ix;     // images -i .. i per axis
  double x_cut;       // in units of r0
};

__global__ __launch_bounds__(kBlock) void cov_grid_kernel(double* __restrict__ R, int mz, int my, int mx, double dh,
                                                          double dy, double dx, double inv_r0, RadialTable t, ImageSum g) {
  const size_t n = (size_t)mz * my * mx;
  const double cut2 = g.x_cut * g.x_cut;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const int ix = (int)(i % mx), iy = (int)((i / mx) % my), iz = (int)(i / ((size_t)mx * my));
    const double z0 = iz * dh, y0 = iy * dy, x0 = ix * dx;
    double acc = 0.0;
    for (int nz = -g.iz; nz <= g.iz; ++nz) {
      const double z = (z0 + nz * g.Lz) * inv_r0;
      if (z * z >= cut2) continue;
      for (int ny_ = -g.iy; ny_ <= g.iy; ++ny_) {
        const double y = (y0 + ny_ * g.Ly) * inv_r0;
        const double zy = z * z + y * y;
        if (zy >= cut2) continue;
        for (int nx_ = -g.ix; nx_ <= g.ix; ++nx_) {
          const double x = (x0 + nx_ * g.Lx) * inv_r0;
          const double r2 = zy + x * x;
          if (r2 < cut2) acc += radial_correlation(t, sqrt(r2));
        }
      }
    }
    R[i] = acc;
  }
}

// one line per workgroup: the n-point transform of the even sequence held as its first n/2 + 1 values
__global__ __launch_bounds__(kBlock) void even_dft_axis_kernel(double* __restrict__ R, int n, int n_inner,
                                                               size_t outer_stride, size_t inner_stride, size_t stride) {
  extern __shared__ __align__(16) double lds_d[];
  const int m = n / 2 + 1;
  double* line = lds_d;
  double* ctab = lds_d + m;
  const size_t base = (size_t)(blockIdx.x / n_inner) * outer_stride + (size_t)(blockIdx.x % n_inner) * inner_stride;
  for (int d = threadIdx.x; d < m; d += kBlock) line[d] = R[base + (size_t)d * stride];
  for (int j = threadIdx.x; j < n; j += kBlock) ctab[j] = cospi(2.0 * (double)j / (double)n);
  __syncthreads();
  for (int k = threadIdx.x; k < m; k += kBlock) {
    double acc0 = 0.0, acc1 = 0.0;
    int idx = 0;
    int d = 1;
    for (; d + 1 < n / 2; d += 2) {
      idx += k;
      if (idx >= n) idx -= n;
      acc0 = fma(line[d], ctab[idx], acc0);
      idx += k;
      if (idx >= n) idx -= n;
      acc1 = fma(line[d + 1], ctab[idx], acc1);
    }
    for (; d < n / 2; ++d) {
      idx += k;
      if (idx >= n) idx -= n;
      acc0 = fma(line[d], ctab[idx], acc0);
    }
    const double ends = line[0] + ((k & 1) ? -line[n / 2] : line[n / 2]);
    R[base + (size_t)k * stride] = ends + 2.0 * (acc0 + acc1);
  }
}

// amp[kx][kz][ky] = sqrt(max(lambda, 0)) in float32, and the sum of amp^2 over the FULL grid
// divided by the zero-lag value of the image sum (rho0[0], kept from before the transform): the screens are
// normalised so that their STRUCTURE FUNCTION is Matern's -- their variance is then 1 + the images' share
__global__ __launch_bounds__(kBlock) void amp_table_kernel(const double* __restrict__ R, int mz, int my, int mx,
                                                           float* __restrict__ amp, double* __restrict__ sum,
                                                           const double* __restrict__ rho0) {
  __shared__ double part[kBlock / 64];
  double acc = 0.0;
  const size_t n = (size_t)mz * my * mx;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const int ky = (int)(i % my), kz = (int)((i / my) % mz), kx = (int)(i / ((size_t)my * mz));
    const double lam = R[((size_t)kz * my + ky) * mx + kx];
    const float a = (float)sqrt(fmax(lam, 0.0));
    amp[i] = a;
    // a cell of the half axes stands for itself and its mirror, but for 0 and n/2
    const int mult = ((ky == 0 || ky == my - 1) ? 1 : 2) * ((kx == 0 || kx == mx - 1) ? 1 : 2) *
                     ((mz == 1 || kz == 0 || kz == mz - 1) ? 1 : 2);
    acc += (double)mult * (double)a * (double)a;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) s += part[w];
    atomicAdd(sum, s / rho0[0]);
  }
}

__global__ void keep_first_kernel(const double* __restrict__ R, double* __restrict__ rho0) { rho0[0] = R[0]; }

constexpr size_t kAmpHeader = 4;  // floats before the amplitudes: the float64 sum, then padding to 16 bytes

__global__ void philox_normal_kernel(float* __restrict__ out, size_t n,
                                     uint32_t key0, uint32_t key1,
                                     uint32_t stream) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U4 rnd = philox4x32_10(
      U4{(uint32_t)i, (uint32_t)(i >> 32), stream, 0u}, key0, key1);
  out[i] = box_muller(rnd.x, rnd.y).x;
}

// float32 Gaussian taps of scipy.ndimage._filters._gaussian_kernel1d (order 0, truncate 4),
// normalised in float64, zero-padded to kMaxFusedRadius + 1 entries; cached per sigma
int get_ftaps(mrx_ctx* ctx, double sigma, int radius, const float** d_out) {
  for (auto& slot : ctx->ftaps)
    if (slot.d_taps && slot.sigma == sigma && slot.radius == radius) {
      slot.pinned = ctx->ftaps_batch;
      *d_out = slot.d_taps;
      return MRX_OK;
    }
  float w[kMaxFusedRadius + 1] = {0.0f};
  double e[kMaxFusedRadius + 1], sum = 0.0;
  for (int k = 0; k <= radius; ++k) {
    e[k] = std::exp(-0.5 / (sigma * sigma) * (double)k * (double)k);
    sum += k ? 2.0 * e[k] : e[k];
  }
  for (int k = 0; k <= radius; ++k) w[k] = (float)(e[k] / sum);
  // round-robin eviction that skips the slots the batch under assembly already points at (a batch
  // takes at most 2 kMaxBatch = 32 of the 128 slots, so the walk always ends)
  int pick = ctx->ftaps_next;
  for (int n = 0; n < mrx_ctx::kFTapSlots && ctx->ftaps[pick].d_taps && ctx->ftaps[pick].pinned == ctx->ftaps_batch; ++n)
    pick = (pick + 1) % mrx_ctx::kFTapSlots;
  auto& slot = ctx->ftaps[pick];
  ctx->ftaps_next = (pick + 1) % mrx_ctx::kFTapSlots;
  slot.pinned = ctx->ftaps_batch;
  if (slot.d_taps)  // a kernel in flight may still read the evicted taps
    MRX_HIP(ctx, hipDeviceSynchronize());
  else
    MRX_HIP(ctx, hipMalloc(&slot.d_taps, sizeof(w)));
  MRX_HIP(ctx, hipMemcpyAsync(slot.d_taps, w, sizeof(w), hipMemcpyHostToDevice, ctx->stream));
  MRX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // w goes out of scope
  slot.sigma = sigma;
  slot.radius = radius;
  *d_out = slot.d_taps;
  return MRX_OK;
}

// Transfer function of the same taps on a periodic axis of n nodes: H[k] = w_0 + 2 sum_j w_j cos(2 pi j k / n),
// k = 0 .. n/2 (real and even: the taps are symmetric), float64 on the host; cached per (sigma, n).
int get_fresp(mrx_ctx* ctx, double sigma, int radius, int n, const float** d_out) {
  for (auto& slot : ctx->fresp)
    if (slot.d_resp && slot.sigma == sigma && slot.n == n && slot.radius == radius) {
      slot.pinned = ctx->ftaps_batch;
      *d_out = slot.d_resp;
      return MRX_OK;
    }
  std::vector<double> w((size_t)radius + 1);
  double sum = 0.0;
  for (int k = 0; k <= radius; ++k) {
    w[(size_t)k] = std::exp(-0.5 / (sigma * sigma) * (double)k * (double)k);
    sum += k ? 2.0 * w[(size_t)k] : w[(size_t)k];
  }
  std::vector<float> h((size_t)n / 2 + 1);
  for (int k = 0; k <= n / 2; ++k) {
    double acc = w[0];
    for (int j = 1; j <= radius; ++j)
      acc += 2.0 * w[(size_t)j] * std::cos(6.283185307179586476925 * (double)(((long long)j * k) % n) / (double)n);
    h[(size_t)k] = (float)(acc / sum);
  }
  int pick = ctx->fresp_next;
  for (int m = 0; m < mrx_ctx::kFRespSlots && ctx->fresp[pick].d_resp && ctx->fresp[pick].pinned == ctx->ftaps_batch; ++m)
    pick = (pick + 1) % mrx_ctx::kFRespSlots;
  auto& slot = ctx->fresp[pick];
  ctx->fresp_next = (pick + 1) % mrx_ctx::kFRespSlots;
  slot.pinned = ctx->ftaps_batch;
  if (slot.d_resp) {  // a kernel in flight may still read the evicted table
    MRX_HIP(ctx, hipDeviceSynchronize());
    (void)hipFree(slot.d_resp);
    slot.d_resp = nullptr;
  }
  MRX_HIP(ctx, hipMalloc(&slot.d_resp, h.size() * sizeof(float)));
  MRX_HIP(ctx, hipMemcpyAsync(slot.d_resp, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  MRX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // h goes out of scope
  slot.sigma = sigma;
  slot.n = n;
  slot.radius = radius;
  *d_out = slot.d_resp;
  return MRX_OK;
}

int ilog2_exact(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return (1 << l) == n ? l : -1;
}

// The normalisation depends on the grid and spectrum only, not on the draw: it
// is reduced once per (ny, nx, dy, dx, r0, nu) and kept in a device slot.
int psd_sum_slot(mrx_ctx* ctx, int ny, int nx, double dy, double dx, double r0,
                 double nu, const double** d_sum, int nh = 0, double dh = 0.0) {
  if (!ctx->d_reduce) {
    MRX_HIP(ctx, hipMalloc(&ctx->d_reduce, mrx_ctx::kPsdSlots * sizeof(double)));
    ctx->reduce_cap = mrx_ctx::kPsdSlots;
  }
  for (int i = 0; i < mrx_ctx::kPsdSlots; ++i) {
    const auto& k = ctx->psd[i];
    if (k.valid && k.ny == ny && k.nx == nx && k.dy == dy && k.dx == dx &&
        k.r0 == r0 && k.nu == nu && k.nh == nh && k.dh == dh) {
      *d_sum = ctx->d_reduce + i;
      return MRX_OK;
    }
  }
  const int slot = ctx->psd_next;
  ctx->psd_next = (ctx->psd_next + 1) % mrx_ctx::kPsdSlots;
  double* dst = ctx->d_reduce + slot;
  const double k0sq = 2.0 * nu / (r0 * r0);
  // sqrt of the PSD: (nu + 1)/2 in two dimensions, (nu + 3/2)/2 in three
  const float expo = (float)(nh > 0 ? -(nu + 1.5) / 2.0 : -(nu + 1.0) / 2.0);
  MRX_HIP(ctx, hipMemsetAsync(dst, 0, sizeof(double), ctx->stream));
  const size_t n = (size_t)ny * nx * (nh > 0 ? nh : 1);
  const int blocks = (int)((n + kBlock - 1) / kBlock < 2048
                               ? (n + kBlock - 1) / kBlock
                               : 2048);
  if (nh > 0)
    hipLaunchKernelGGL(psd_sum3d_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, dst, nh, ny, nx, dh, dy, dx,
                       k0sq, expo);
  else
    hipLaunchKernelGGL(psd_sum_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream,
                       dst, ny, nx, dy, dx, k0sq, expo);
  MRX_CHECK_LAUNCH(ctx);
  ctx->psd[slot] = {true, ny, nx, dy, dx, r0, nu, nh, dh};
  *d_sum = dst;
  return MRX_OK;
}

}  // namespace

extern "C" {

int mrx_screen_psd_sum(mrx_ctx* ctx, int ny, int nx, double dy, double dx,
                       double r0, double nu, double* host_sum) {
  MRX_ENTER(ctx);
  if (!ctx || !host_sum) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, ny > 0 && nx > 0 && dy > 0 && dx > 0 && r0 > 0 && nu > 0,
              "sizes, steps, r0 and nu must be positive");
  const double* d_sum = nullptr;
  int rc = psd_sum_slot(ctx, ny, nx, dy, dx, r0, nu, &d_sum);
  if (rc != MRX_OK) return rc;
  MRX_HIP(ctx, hipMemcpyAsync(host_sum, d_sum, sizeof(double),
                              hipMemcpyDeviceToHost, ctx->stream));
  MRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MRX_OK;
}

int mrx_screen_work_floats(int ny, int nx, int n_screens, size_t* floats) {
  if (!floats || ny <= 0 || nx <= 0 || n_screens < 0) return MRX_ERR_INVALID;
  *floats = 2 * (size_t)n_screens * ((size_t)nx / 2 + 1) * ((size_t)ny + kPitchPad);
  return MRX_OK;
}

// Passes 1 and 2 for n_screens screens that share one FFT domain, kMaxBatch per launch.
// spectra != nullptr: plane i's half spectrum is already in its slot of d_work (3-D generator)
// and d_sum3d is the normalisation; otherwise every screen draws its own 2-D spectrum.
static int run_screen_passes(mrx_ctx* ctx, uint64_t seed, int ny, int nx, const mrx_screen_desc* screens,
                             int n_screens, float* d_work, bool from_work, const double* d_sum3d) {
  const int ly = ilog2_exact(ny), lx = ilog2_exact(nx);
  size_t need = 0;
  mrx_screen_work_floats(ny, nx, n_screens, &need);
  const size_t per_screen = need / (size_t)n_screens;
  const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
  const int n2 = nx / 2;

  // rows per workgroup of the second pass: about 1024 complex cells per workgroup keep 6-7
  // workgroups on a CU; measured on 2048^2 x 8 layers: 1 row 145 us, 2 rows 165 us, 4 rows 245 us
  const size_t fixed2 = (size_t)(nx / 4 + n2 / 4 + 4) * sizeof(float2);
  int lj = 2;
  while (lj > 0 && (n2 << lj) > 1024) --lj;
  const size_t lds2 = 2 * ((size_t)n2 << lj) * sizeof(float2) + fixed2;
  const size_t lds1 = (size_t)(2 * ny + ny / 4) * sizeof(float2);
  MRX_LDS_CAP(ctx, screen_half_spectrum_fft_y, lds1);
  if (lj == 2) MRX_LDS_CAP(ctx, screen_c2r_x<2>, lds2);
  else if (lj == 1) MRX_LDS_CAP(ctx, screen_c2r_x<1>, lds2);
  else MRX_LDS_CAP(ctx, screen_c2r_x<0>, lds2);

  for (int first = 0; first < n_screens; first += kMaxBatch) {
    const int nb = n_screens - first < kMaxBatch ? n_screens - first : kMaxBatch;
    ScreenBatchArgs args{};
    int max_out_ny = 0;
    bool late_smooth[kMaxBatch] = {false};
    ++ctx->ftaps_batch;  // the tap slots taken from here on stay put until this batch is launched
    for (int i = 0; i < nb; ++i) {
      const mrx_screen_desc& d = screens[first + i];
      MRX_REQUIRE(ctx, d.d_out != nullptr, "null output pointer");
      MRX_REQUIRE(ctx, from_work || d.d_amp || (d.dy > 0 && d.dx > 0 && d.r0 > 0 && d.nu > 0),
                  "steps, r0 and nu must be positive");
      const int out_ny = d.out_ny > 0 ? d.out_ny : ny, out_nx = d.out_nx > 0 ? d.out_nx : nx;
      MRX_REQUIRE(ctx, out_ny <= ny && out_nx <= nx, "written block larger than the FFT domain");
      const size_t ld = d.ld_out ? d.ld_out : (size_t)out_nx;
      MRX_REQUIRE(ctx, ld >= (size_t)out_nx, "ld_out smaller than the row");
      MRX_REQUIRE(ctx, d.sigma_y >= 0.0 && d.sigma_x >= 0.0, "sigma must be >= 0");
      ScreenLayerArgs& L = args.l[i];
      const double* d_sum = d_sum3d;
      int rc = MRX_OK;
      L.amp = nullptr;
      if (!from_work && d.d_amp) {  // tabulated amplitudes: the table's header is their sum
        MRX_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(d.d_amp) & 15u) == 0, "d_amp must be 16-byte aligned");
        d_sum = reinterpret_cast<const double*>(d.d_amp);
        L.amp = d.d_amp + kAmpHeader;
      } else if (!from_work) {
        rc = psd_sum_slot(ctx, ny, nx, d.dy, d.dx, d.r0, d.nu, &d_sum);
        if (rc != MRX_OK) return rc;
      }
      L.work = reinterpret_cast<float2*>(d_work) + (size_t)(first + i) * (per_screen / 2);
      L.out = d.d_out;
      L.psd_sum = d_sum;
      L.dy = d.dy;
      L.dx = d.dx;
      L.k0sq = from_work ? 0.0 : 2.0 * d.nu / (d.r0 * d.r0);
      L.ld_out = ld;
      L.expo = (float)(-(d.nu + 1.0) / 2.0);
      L.from_work = from_work ? 1 : 0;
      // scipy: radius = int(truncate * sigma + 0.5), truncate = 4; a sigma <= 1e-15 skips the axis
      const int ry = d.sigma_y > 1e-15 ? (int)(4.0 * d.sigma_y + 0.5) : 0;
      const int rx = d.sigma_x > 1e-15 ? (int)(4.0 * d.sigma_x + 0.5) : 0;
      L.taps_y = L.taps_x = L.resp_y = L.resp_x = nullptr;
      if (d.periodic_beam && (ry > 0 || rx > 0)) {
        // the beam as a factor of the spectrum (a skipped axis: sigma = 0 has the response 1: one tap)
        MRX_REQUIRE(ctx, ry < ny / 2 && rx < nx / 2, "periodic_beam: the stencil radius must stay below half the domain");
        if ((rc = get_fresp(ctx, ry > 0 ? d.sigma_y : 1.0, ry, ny, &L.resp_y)) != MRX_OK) return rc;
        if ((rc = get_fresp(ctx, rx > 0 ? d.sigma_x : 1.0, rx, nx, &L.resp_x)) != MRX_OK) return rc;
        L.ry = L.rx = 0;
      } else {
      // radii beyond the LDS tap table (beams of > 32 pixels) take the separate stencil kernels
      late_smooth[i] = ry > kMaxFusedRadius || rx > kMaxFusedRadius;
      L.ry = late_smooth[i] ? 0 : ry;
      L.rx = late_smooth[i] ? 0 : rx;
      if (L.ry > 0 && (rc = get_ftaps(ctx, d.sigma_y, L.ry, &L.taps_y)) != MRX_OK) return rc;
      if (L.rx > 0 && (rc = get_ftaps(ctx, d.sigma_x, L.rx, &L.taps_x)) != MRX_OK) return rc;
      }
      L.out_ny = out_ny;
      L.out_nx = out_nx;
      L.stream = d.stream;
      if (out_ny > max_out_ny) max_out_ny = out_ny;
    }
    // the column transforms: in registers for ny = 1024, 2048, 4096 (the two self-mirrored columns
    // through the Stockham kernel), MRX_OPT_SCREEN_STOCKHAM keeps the LDS form for all columns
    const int rb = ctx->options[MRX_OPT_SCREEN_STOCKHAM] ? 0 : ny == 1024 ? 4 : ny == 2048 ? 8 : ny == 4096 ? 16 : 0;
    if (rb && n2 > 1) {
      // the exchange image(s) of the register form.  The two self-mirrored columns take the Stockham form, whose LDS is
      // 2.25 ny complex: as the launch's last two workgroups where that costs the others no place on a CU (ny <= 2048),
      // else in a launch of their own (ny = 4096: 73.7 KB against the register form's 34.8)
      const size_t lds_images = kScreenImages * (size_t)kFft4096Pitch * kBlock * sizeof(float2);
      const int per_cu_images = (int)((160u * 1024u) / (lds_images + 512)), per_cu_both = (int)((160u * 1024u) / (std::max(lds_images, lds1) + 512));
      const int inside = per_cu_both >= per_cu_images ? 1 : 0;
      const size_t lds_r = inside ? std::max(lds_images, lds1) : lds_images;
      const int cols = kBlock / (16 * rb);
      const dim3 grid_r(mrx_ceil_div(n2 - 1, cols) + 2 * inside, nb);
      if (rb == 4) {
        MRX_LDS_CAP(ctx, screen_half_spectrum_regs<4>, lds_r);
        hipLaunchKernelGGL(screen_half_spectrum_regs<4>, grid_r, dim3(kBlock), lds_r, ctx->stream, args, nx, key0, key1, inside);
      } else if (rb == 8) {
        MRX_LDS_CAP(ctx, screen_half_spectrum_regs<8>, lds_r);
        hipLaunchKernelGGL(screen_half_spectrum_regs<8>, grid_r, dim3(kBlock), lds_r, ctx->stream, args, nx, key0, key1, inside);
      } else {
        MRX_LDS_CAP(ctx, screen_half_spectrum_regs<16>, lds_r);
        hipLaunchKernelGGL(screen_half_spectrum_regs<16>, grid_r, dim3(kBlock), lds_r, ctx->stream, args, nx, key0, key1, inside);
      }
      if (!inside) hipLaunchKernelGGL(screen_half_spectrum_fft_y, dim3(2, nb), dim3(kBlock), lds1, ctx->stream, args, ny, nx, ly, key0, key1, n2);
    } else {
      hipLaunchKernelGGL(screen_half_spectrum_fft_y, dim3(n2 + 1, nb), dim3(kBlock), lds1, ctx->stream,
                         args, ny, nx, ly, key0, key1, 1);
    }
    MRX_CHECK_LAUNCH(ctx);
    // the row transforms: in registers for nx = 2048, 4096, 8192 (MRX_OPT_SCREEN_STOCKHAM keeps the LDS form)
    const int rb2 = ctx->options[MRX_OPT_SCREEN_STOCKHAM] ? 0 : nx == 2048 ? 4 : nx == 4096 ? 8 : nx == 8192 ? 16 : 0;
    if (rb2) {
      const size_t lds_r2 = kScreenImages * (size_t)kFft4096Pitch * kBlock * sizeof(float2);
      const dim3 grid_r2(mrx_ceil_div(max_out_ny, 16 / rb2), nb);
      if (rb2 == 4) {
        MRX_LDS_CAP(ctx, screen_c2r_regs<4>, lds_r2);
        hipLaunchKernelGGL(screen_c2r_regs<4>, grid_r2, dim3(kBlock), lds_r2, ctx->stream, args, ny);
      } else if (rb2 == 8) {
        MRX_LDS_CAP(ctx, screen_c2r_regs<8>, lds_r2);
        hipLaunchKernelGGL(screen_c2r_regs<8>, grid_r2, dim3(kBlock), lds_r2, ctx->stream, args, ny);
      } else {
        MRX_LDS_CAP(ctx, screen_c2r_regs<16>, lds_r2);
        hipLaunchKernelGGL(screen_c2r_regs<16>, grid_r2, dim3(kBlock), lds_r2, ctx->stream, args, ny);
      }
    } else {
    const dim3 grid2(mrx_ceil_div(max_out_ny, 1 << lj), nb);
    if (lj == 2)
      hipLaunchKernelGGL(screen_c2r_x<2>, grid2, dim3(kBlock), lds2, ctx->stream, args, ny, nx, lx - 1);
    else if (lj == 1)
      hipLaunchKernelGGL(screen_c2r_x<1>, grid2, dim3(kBlock), lds2, ctx->stream, args, ny, nx, lx - 1);
    else
      hipLaunchKernelGGL(screen_c2r_x<0>, grid2, dim3(kBlock), lds2, ctx->stream, args, ny, nx, lx - 1);
    }
    MRX_CHECK_LAUNCH(ctx);
    for (int i = 0; i < nb; ++i) {
      if (!late_smooth[i]) continue;
      const mrx_screen_desc& d = screens[first + i];
      const ScreenLayerArgs& L = args.l[i];
      MRX_REQUIRE(ctx, L.ld_out == (unsigned long long)L.out_nx,
                  "a beam wider than 32 pixels needs a contiguous output (ld_out == out_nx)");
      MRX_REQUIRE(ctx, !from_work, "a beam wider than 32 pixels is not supported by the 3-D generator");
      // the half spectra are consumed: the work buffer is free scratch now
      int rc = mrx_gauss_smooth2d(ctx, L.out, L.out, d_work, L.out_ny, L.out_nx, d.sigma_y, d.sigma_x, 4.0);
      if (rc != MRX_OK) return rc;
    }
  }
  return MRX_OK;
}

int mrx_screen_generate_batch(mrx_ctx* ctx, uint64_t seed, int ny, int nx,
                              const mrx_screen_desc* screens, int n_screens, float* d_work,
                              size_t work_floats) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, n_screens >= 0, "negative count");
  if (n_screens == 0) return MRX_OK;
  MRX_REQUIRE(ctx, screens && d_work, "null pointer");
  const int ly = ilog2_exact(ny), lx = ilog2_exact(nx);
  if (ly < 0 || lx < 0 || ny < 64 || nx < 64 || ny > 8192 || nx > 8192)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED,
                    "screen sides must be powers of two in [64, 8192] (got %d x %d)", ny, nx);
  size_t need = 0;
  mrx_screen_work_floats(ny, nx, n_screens, &need);
  MRX_REQUIRE(ctx, work_floats >= need, "work buffer smaller than mrx_screen_work_floats()");
  MRX_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(d_work) & 15u) == 0, "d_work must be 16-byte aligned");
  return run_screen_passes(ctx, seed, ny, nx, screens, n_screens, d_work, false, nullptr);
}

int mrx_screen3d_work_floats(int nh, int ny, int nx, int n_planes, size_t* floats) {
  if (!floats || nh <= 0 || ny <= 0 || nx <= 0 || n_planes < 0) return MRX_ERR_INVALID;
  size_t planes = 0;
  mrx_screen_work_floats(ny, nx, n_planes, &planes);
  *floats = planes + 8 * (size_t)n_planes + 16;  // + the device copy of the plane table
  return MRX_OK;
}

int mrx_screen_generate_3d(mrx_ctx* ctx, uint64_t seed, uint32_t stream, int nh, int ny, int nx,
                           double dh, double dy, double dx, double r0, double nu,
                           const double* plane_pos, const double* plane_scale,
                           const mrx_screen_desc* planes, int n_planes, float* d_work,
                           size_t work_floats, const float* d_amp) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, n_planes >= 0, "negative count");
  if (n_planes == 0) return MRX_OK;
  MRX_REQUIRE(ctx, planes && plane_pos && d_work, "null pointer");
  MRX_REQUIRE(ctx, d_amp || (dh > 0 && dy > 0 && dx > 0 && r0 > 0 && nu > 0), "steps, r0 and nu must be positive");
  MRX_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(d_amp) & 15u) == 0, "d_amp must be 16-byte aligned");
  MRX_REQUIRE(ctx, stream < 65536u, "the 3-D generator takes streams below 65536");
  const int lh = ilog2_exact(nh), ly = ilog2_exact(ny), lx = ilog2_exact(nx);
  if (lh < 0 || ly < 0 || lx < 0 || nh < 8 || nh > 2048 || ny < 64 || nx < 64 || ny > 8192 || nx > 8192)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED,
                    "3-D domain: nh a power of two in [8, 2048], ny and nx in [64, 8192] (got %d x %d x %d)", nh, ny, nx);
  size_t need = 0, plane_floats = 0;
  mrx_screen3d_work_floats(nh, ny, nx, n_planes, &need);
  mrx_screen_work_floats(ny, nx, n_planes, &plane_floats);
  MRX_REQUIRE(ctx, work_floats >= need, "work buffer smaller than mrx_screen3d_work_floats()");
  MRX_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(d_work) & 15u) == 0, "d_work must be 16-byte aligned");
  const size_t per_plane = plane_floats / (size_t)n_planes;  // floats
  std::vector<Screen3dPlane> table((size_t)n_planes);
  for (int p = 0; p < n_planes; ++p) {
    const double pos = plane_pos[p];
    MRX_REQUIRE(ctx, pos >= 0.0 && pos <= (double)(nh - 1), "plane position outside the FFT planes");
    int h0 = (int)pos;
    if (h0 > nh - 2) h0 = nh - 2;
    table[(size_t)p].work = reinterpret_cast<float2*>(d_work) + (size_t)p * (per_plane / 2);
    table[(size_t)p].h0 = h0;
    table[(size_t)p].w = (float)(pos - (double)h0);
    table[(size_t)p].scale = plane_scale ? (float)plane_scale[p] : 1.0f;
  }
  static_assert(sizeof(Screen3dPlane) <= 8 * sizeof(float), "plane table slot");
  Screen3dPlane* d_table = reinterpret_cast<Screen3dPlane*>(d_work + ((plane_floats + 3) & ~(size_t)3));
  MRX_HIP(ctx, hipMemcpyAsync(d_table, table.data(), sizeof(Screen3dPlane) * (size_t)n_planes, hipMemcpyHostToDevice,
                              ctx->stream));
  MRX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the host table goes out of scope
  const double* d_sum = reinterpret_cast<const double*>(d_amp);
  if (!d_amp) {
    int rc = psd_sum_slot(ctx, ny, nx, dy, dx, r0, nu, &d_sum, nh, dh);
    if (rc != MRX_OK) return rc;
  }

  Screen3dArgs g{};
  g.amp = d_amp ? d_amp + kAmpHeader : nullptr;
  g.planes = d_table;
  g.n_planes = n_planes;
  g.dh = dh;
  g.dy = dy;
  g.dx = dx;
  g.k0sq = 2.0 * nu / (r0 * r0);
  g.expo = (float)(-(nu + 1.5) / 2.0);
  g.stream = stream;
  int lj = 0;
  while (((nh << (lj + 1)) <= 2048) && (1 << (lj + 1)) <= ny) ++lj;  // J sequences of nh cells: <= 2048 per workgroup
  const size_t lds0 = (size_t)(2 * (nh << lj) + nh / 4) * sizeof(float2);
  MRX_LDS_CAP(ctx, screen3d_fft_h, lds0);
  hipLaunchKernelGGL(screen3d_fft_h, dim3(ny >> lj, nx / 2 + 1), dim3(kBlock), lds0, ctx->stream, g, nh, ny, nx, lh, lj,
                     (uint32_t)seed, (uint32_t)(seed >> 32));
  MRX_CHECK_LAUNCH(ctx);
  return run_screen_passes(ctx, seed, ny, nx, planes, n_planes, d_work, true, d_sum);
}

int mrx_screen_amp_floats(int nh, int ny, int nx, int n_radial, size_t* table_floats, size_t* work_floats) {
  if (nh < 0 || ny <= 0 || nx <= 0 || n_radial < 0) return MRX_ERR_INVALID;
  const size_t m = (size_t)(nh > 0 ? nh / 2 + 1 : 1) * ((size_t)ny / 2 + 1) * ((size_t)nx / 2 + 1);
  if (table_floats) *table_floats = kAmpHeader + ((m + 3) & ~(size_t)3);
  if (work_floats) *work_floats = 2 * (m + 2 * (size_t)n_radial + 1) + 4;
  return MRX_OK;
}

int mrx_screen_amplitudes(mrx_ctx* ctx, int nh, int ny, int nx, double dh, double dy, double dx, double r0,
                          const double* log_cov, const double* log_sf, int n_radial, double log_first,
                          double log_step, double x_cut, float* d_table, float* d_work, size_t work_floats) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, log_cov && log_sf && d_table && d_work, "null pointer");
  MRX_REQUIRE(ctx, n_radial >= 4 && n_radial <= (1 << 20) && log_step > 0, "the radial table needs 4 .. 2^20 nodes on an increasing log grid");
  MRX_REQUIRE(ctx, dy > 0 && dx > 0 && r0 > 0 && (nh == 0 || dh > 0), "steps and r0 must be positive");
  MRX_REQUIRE(ctx, x_cut > 0, "x_cut must be positive");
  const int lh = nh ? ilog2_exact(nh) : 0, ly = ilog2_exact(ny), lx = ilog2_exact(nx);
  if (lh < 0 || ly < 0 || lx < 0 || (nh && (nh < 8 || nh > 2048)) || ny < 64 || nx < 64 || ny > 8192 || nx > 8192)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED,
                    "amplitude table: ny and nx powers of two in [64, 8192], nh 0 or in [8, 2048] (got %d x %d x %d)", nh, ny, nx);
  size_t need = 0;
  mrx_screen_amp_floats(nh, ny, nx, n_radial, nullptr, &need);
  MRX_REQUIRE(ctx, work_floats >= need, "work buffer smaller than mrx_screen_amp_floats()");
  MRX_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(d_work) | reinterpret_cast<uintptr_t>(d_table)) & 15u) == 0,
              "d_table and d_work must be 16-byte aligned");
  const int mz = nh ? nh / 2 + 1 : 1, my = ny / 2 + 1, mx = nx / 2 + 1;
  const size_t m = (size_t)mz * my * mx;
  double* R = reinterpret_cast<double*>(d_work);
  double* d_cov = R + m;
  double* d_sf = d_cov + n_radial;
  double* d_rho0 = d_sf + n_radial;
  MRX_HIP(ctx, hipMemcpyAsync(d_cov, log_cov, sizeof(double) * (size_t)n_radial, hipMemcpyHostToDevice, ctx->stream));
  MRX_HIP(ctx, hipMemcpyAsync(d_sf, log_sf, sizeof(double) * (size_t)n_radial, hipMemcpyHostToDevice, ctx->stream));
  MRX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's host arrays may go away
  RadialTable t{};
  t.log_cov = d_cov;
  t.log_sf = d_sf;
  t.n = n_radial;
  t.log_first = log_first;
  t.inv_dlog = 1.0 / log_step;
  t.r_min = std::exp(log_first);
  t.r_max = std::exp(log_first + log_step * (double)(n_radial - 1));
  const int blocks = (int)std::min<size_t>((m + kBlock - 1) / kBlock, 4096);
  ImageSum g{};
  g.Lz = nh ? nh * dh : 1.0;
  g.Ly = ny * dy;
  g.Lx = nx * dx;
  g.x_cut = x_cut;
  // an offset of the half grid is at most half a period from the origin: images up to x_cut r0 + L/2 away
  auto images = [&](double L) { return (int)std::ceil(x_cut * r0 / L + 0.5); };
  g.iz = nh ? images(g.Lz) : 0;
  g.iy = images(g.Ly);
  g.ix = images(g.Lx);
  MRX_REQUIRE(ctx, (double)(2 * g.iz + 1) * (2 * g.iy + 1) * (2 * g.ix + 1) <= 1.0e6,
              "the periodic domain is too small against x_cut r0 (more than 1e6 images per grid point)");
  hipLaunchKernelGGL(cov_grid_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, R, mz, my, mx, nh ? dh : 0.0, dy, dx,
                     1.0 / r0, t, g);
  MRX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(keep_first_kernel, dim3(1), dim3(1), 0, ctx->stream, R, d_rho0);
  MRX_CHECK_LAUNCH(ctx);
  auto axis = [&](int n, int lines, int n_inner, size_t outer_stride, size_t inner_stride, size_t stride) -> int {
    const size_t lds = (size_t)(n / 2 + 1 + n) * sizeof(double);
    MRX_LDS_CAP(ctx, even_dft_axis_kernel, lds);
    hipLaunchKernelGGL(even_dft_axis_kernel, dim3(lines), dim3(kBlock), lds, ctx->stream, R, n, n_inner, outer_stride,
                       inner_stride, stride);
    MRX_CHECK_LAUNCH(ctx);
    return MRX_OK;
  };
  int rc = axis(nx, mz * my, 1, (size_t)mx, 0, 1);                                  // along x: lines (iz, iy)
  if (rc == MRX_OK) rc = axis(ny, mz * mx, mx, (size_t)my * mx, 1, (size_t)mx);      // along y: lines (iz, ix)
  if (rc == MRX_OK && nh) rc = axis(nh, my * mx, my * mx, 0, 1, (size_t)my * mx);    // along h: lines (iy, ix)
  if (rc != MRX_OK) return rc;
  MRX_HIP(ctx, hipMemsetAsync(d_table, 0, sizeof(float) * kAmpHeader, ctx->stream));
  hipLaunchKernelGGL(amp_table_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, R, mz, my, mx, d_table + kAmpHeader,
                     reinterpret_cast<double*>(d_table), d_rho0);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_screen_generate(mrx_ctx* ctx, uint64_t seed, uint32_t stream, int ny,
                        int nx, double dy, double dx, double r0, double nu,
                        float* d_out, float* d_work) {
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, d_out && d_work, "null pointer");
  mrx_screen_desc d{};
  d.d_out = d_out;
  d.stream = stream;
  d.dy = dy;
  d.dx = dx;
  d.r0 = r0;
  d.nu = nu;
  size_t need = 0;
  if (mrx_screen_work_floats(ny, nx, 1, &need) != MRX_OK)
    return mrx_fail(ctx, MRX_ERR_INVALID, "mrx_screen_generate: sizes must be positive");
  return mrx_screen_generate_batch(ctx, seed, ny, nx, &d, 1, d_work, need);
}

int mrx_philox_normal(mrx_ctx* ctx, uint64_t seed, uint32_t stream, size_t n,
                      float* d_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, d_out != nullptr || n == 0, "null pointer");
  if (n == 0) return MRX_OK;
  const size_t blocks = (n + kBlock - 1) / kBlock;
  MRX_REQUIRE(ctx, blocks <= 0x7fffffffu, "n too large");
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)blocks), dim3(kBlock),
                     0, ctx->stream, d_out, n, (uint32_t)seed,
                     (uint32_t)(seed >> 32), stream);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

// Host evaluation of the same Philox routine the kernels compile (one source,
// __host__ __device__), for the known-answer test.
int mrx_philox_raw(mrx_ctx* ctx, uint64_t seed, uint32_t c0, uint32_t c1,
                   uint32_t c2, uint32_t c3, uint32_t host_out[4]) {
  (void)ctx;
  if (!host_out) return MRX_ERR_INVALID;
  const U4 r = philox4x32_10(U4{c0, c1, c2, c3}, (uint32_t)seed,
                             (uint32_t)(seed >> 32));
  host_out[0] = r.x;
  host_out[1] = r.y;
  host_out[2] = r.z;
  host_out[3] = r.w;
  return MRX_OK;
}

}  // extern "C"
