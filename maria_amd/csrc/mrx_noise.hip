// Detector noise for gfx950: white + 1/f ("pink") noise with spatially
// correlated modes, the model of sim/noise.py:18-63 and
// noise/generation.py:11-51:
//
//   noise[d,t] = scale_d * ( sqrt(fs) w[d,t]
//                            + sqrt(c)   sum_m B[d,m] (sqrt(fs) w'[m,t] + P_m[t])
//                            + sqrt(1-c) p_d[t] )
//
// (scale_d becomes scale_d + per_loading * loading[d,t] when the band's NEP grows
// with the optical loading, sim/noise.py:35-37)
//
// with w, w' white N(0,1), p_d and P_m independent pink series of two-sided
// spectrum S(f) = (knee/2)/|f| (pink == white at f = knee), c the correlated
// proportion and B the spatial basis (utils/linalg.py:105-126, host side).  A
// mode carries its own white part because the reference builds the modes by
// calling the generator again (generation.py:41-43).
//
// The reference shapes white noise with a length-T FFT per detector
// (generation.py:31-37).  Here the pink and the correlated parts are synthesised
// directly in the frequency domain on a power-of-two period N = N1*N2 >= T, and
// the first T samples are kept:
//   * own pink part: X_k = sqrt(knee/|k|) (g1 + i g2), independent normals for
//     every k, so that the variance per frequency bin is the reference's
//     knee/|k|.  The amplitude is even in k, so the inverse transform is a
//     circular complex Gaussian series whose real and imaginary parts are two
//     independent real series of that spectrum: one transform serves two
//     detectors (rows a, b).
//   * modes: each mode is a real series with a Hermitian spectrum F_m[k] of
//     variance fs/N + knee/|k| per bin (white + pink), tabulated once per call;
//     the pair's spectrum gets sum_m sqrt(c) (B[a,m] + i B[b,m]) F_m[k], which
//     puts mode m into the real part with weight B[a,m] and into the imaginary
//     part with B[b,m].  All of the linear algebra happens before the transform.
// The length-N transform is a four-step FFT on the in-LDS Stockham passes the
// screens use, k = k1 + N1 k2, t = j + N2 m:
//   pass 1  per (series, k1): build the cells, transform over k2 (length N2),
//           multiply by exp(2 pi i k1 j / N)                   -> A[k1][j]  (HBM)
//   pass 2  transform over k1 (length N1) and finish the sample: scale, store.
//           N1 = 64 (periods up to 2^19): one thread per j, the transform in
//           registers.  Otherwise a workgroup takes J = 4096/N1 adjacent j
//           through LDS.
// A detector's own white part, sqrt(fs) w[d,t], is drawn in the same spectrum: white
// noise of period N has variance fs/N in every cell and any T of its N samples are
// independent, and the sum of two independent Gaussian series is the Gaussian series of
// the summed spectra -- one draw of amplitude sqrt(fs/N + (1-c) knee/|k|) per cell
// instead of a spectral draw per cell plus a time-domain draw per sample (-20 % of the
// arithmetic).  Only the cells below k_cut keep two draws (see window_mean_factor).
// HBM traffic per detector: 8 N bytes of scratch (A written and read once, shared
// by two detectors) + 4 T of TOD; the arithmetic (Philox, Box-Muller, ~5 log2 N
// flops per cell) is what bounds the kernels.
// Normals come from Philox-4x32-10 keyed by (seed, global detector index): every
// GPU can regenerate any detector's noise independently, and the modes (keyed by
// the seed alone) are the same on every shard.  Parity with the reference is
// statistical (different generator and period), like the screens'.
#include "mrx_internal.h"

#include "mrx_spectral.h"
#include "mrx_krj.h"  // the two-rate writer's K_RJ division (the machinery of mrx_tod_to_krj)


namespace {

using namespace mrx_dev;

constexpr uint32_t kTagPink = 0x50494e4bu;   // counter word 3: 'PINK'
constexpr uint32_t kTagWhite = 0x57484954u;  // 'WHIT'
constexpr uint32_t kTagMode = 0x4d4f4445u;   // 'MODE'
constexpr uint32_t kTagOwnWhite = 0x57485432u;  // 'WHT2': white part of a pair's cells below k_cut
constexpr int kTileCells = 4096;             // complex values per workgroup of the LDS pass 2
constexpr int kMaxModes = 8;

typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef float vfloat2 __attribute__((ext_vector_type(2)));

// amplitude of spectrum cell k of a length-n series: sqrt(knee/|k|), 0 below k_min (>= 1)
__device__ __forceinline__ float pink_amp(int k, int n, float sqrt_knee, int k_min) {
  const int kk = k < n - k ? k : n - k;
  return kk < k_min ? 0.0f : sqrt_knee * __builtin_amdgcn_rsqf((float)kk);
}

// own part of a pair's cell k above k_cut: white of variance white_var and pink of variance
// pink_var / |k| in one draw
__device__ __forceinline__ float merged_amp(int k, int n, float white_var, float pink_var, int k_min) {
  const int kk = k < n - k ? k : n - k;
  return __builtin_sqrtf(white_var + (kk < k_min ? 0.0f : pink_var / (float)kk));
}

// four white normals for samples 4 q .. 4 q + 3 of row `id`
__device__ __forceinline__ vfloat4 white4(uint64_t q, uint32_t id, uint32_t key0, uint32_t key1) {
  const U4 rnd = philox4x32_10(U4{(uint32_t)q, id, (uint32_t)(q >> 32), kTagWhite}, key0, key1);
  const float2 a = box_muller(rnd.x, rnd.y), b = box_muller(rnd.z, rnd.w);
  return vfloat4{a.x, a.y, b.x, b.y};
}

// The reference's pink series has period T (generation.py:31-37): no power below fs/T and a
// mean of exactly zero over the TOD (tests/noise/test_noise.py:7-31 relies on it: the mean of
// a detector's noise is white).  On the longer period N the same two properties are restored
// by (a) dropping the cells below k_min = ceil(N/T) and (b) subtracting the window mean of the
// pink part, (1/T) sum_k X_k W_k with W_k = sum_{t<T} exp(2 pi i k t / N), which converges
// like k^-3: the cells below k_cut = 64 k_min carry all of it.
// W_k / T = exp(i pi k (T-1)/N) sin(pi k T / N) / (T sin(pi k / N)), k != 0.
__device__ __forceinline__ double2 window_mean_factor(int k, int n, int T) {
  const long long two_n = 2LL * n;
  const double a = (double)(((long long)k * (T - 1)) % two_n) / (double)n;  // in units of pi
  const double b = (double)(((long long)k * T) % two_n) / (double)n;
  double sa, ca;
  sincospi(a, &sa, &ca);
  const double r = sinpi(b) / ((double)T * sinpi((double)k / (double)n));
  return make_double2(ca * r, sa * r);
}

struct WindowArgs {
  int T, k_min, k_cut;
  double2* mean;  // [pairs]: window mean of each pair's pink part (re: row a, im: row b)
  double* mu;     // [n_modes]: window mean of each mode's pink part
};

// Spectra of the modes, F[m][k1][k2] (the order pass 1 reads them in), k = k1 + n1 k2:
// Hermitian, white part of variance fs/N plus pink part of variance knee/|k| per cell (two
// independent draws, so that the pink part's window mean can be taken out); the cells k and
// N - k share their draws.
__global__ __launch_bounds__(kBlock) void noise_mode_table(float2* __restrict__ F, int n1, int n2,
                                                           float white_var, float knee, WindowArgs w,
                                                           uint32_t key0, uint32_t key1) {
  const int n = n1 * n2;
  const int c = blockIdx.x * kBlock + threadIdx.x;  // k1 * n2 + k2
  if (c >= n) return;
  const int k1 = c / n2, k2 = c - k1 * n2;
  const int k = k1 + n1 * k2;
  const int kk = k < n - k ? k : n - k;
  const U4 rnd = philox4x32_10(U4{(uint32_t)kk, blockIdx.y, 0u, kTagMode}, key0, key1);
  const float2 hw = box_muller(rnd.x, rnd.y), hp = box_muller(rnd.z, rnd.w);
  const float var_p = kk < w.k_min ? 0.0f : knee / (float)kk;
  const bool self = kk == 0 || 2 * kk == n;  // self-conjugate cells are real
  const float aw = sqrtf(self ? white_var : 0.5f * white_var), ap = sqrtf(self ? var_p : 0.5f * var_p);
  const float sgn = k == kk ? 1.0f : -1.0f;
  const float2 vp = make_float2(ap * hp.x, self ? 0.0f : sgn * ap * hp.y);
  F[(size_t)blockIdx.y * n + c] = make_float2(aw * hw.x + vp.x, (self ? 0.0f : sgn * aw * hw.y) + vp.y);
}

// sum of a double over the workgroup (result valid in thread 0)
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  scratch[threadIdx.x] = v;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) scratch[threadIdx.x] += scratch[threadIdx.x + s];
    __syncthreads();
  }
  return scratch[0];
}

// mu[m] = window mean of mode m's pink part: the cells k_min <= |k| < k_cut of the table above,
// redrawn (block m); the pair k, N - k contributes 2 Re(F_k W_k / T).
__global__ __launch_bounds__(kBlock) void noise_mode_means(int n, float knee, WindowArgs w, uint32_t key0,
                                                           uint32_t key1) {
  __shared__ double scratch[kBlock];
  double acc = 0.0;
  for (int kk = w.k_min + threadIdx.x; kk < w.k_cut; kk += kBlock) {
    const U4 rnd = philox4x32_10(U4{(uint32_t)kk, blockIdx.x, 0u, kTagMode}, key0, key1);
    const float2 hp = box_muller(rnd.z, rnd.w);
    const bool self = 2 * kk == n;
    const double ap = sqrt((self ? 1.0 : 0.5) * (double)(knee / (float)kk));
    const double2 f = window_mean_factor(kk, n, w.T);
    acc += self ? ap * hp.x * f.x : 2.0 * ap * ((double)hp.x * f.x - (double)hp.y * f.y);
  }
  const double total = block_sum(acc, scratch);
  if (threadIdx.x == 0) w.mu[blockIdx.x] = total;
}

struct SpectrumArgs {
  const float2* F;     // [n_modes][n] mode spectra, or null
  const float* basis;  // [rows][n_modes]
  int n_modes;
  int row0, rows;      // series s holds rows row0 + 2 s and row0 + 2 s + 1 (< row0 + rows)
  float w_ind, w_corr;
  float knee;
  float white_var;     // fs / N: a detector's own white part, per cell
  uint32_t series0;    // Philox id of series 0
  WindowArgs win;
};

// mean[pair] = window mean of the pair's pink part (re: row a, im: row b): its own cells
// k_min <= |k| < k_cut redrawn exactly as pass 1 draws them, plus the modes' means through the
// pair's coefficients.  One block per pair.
__global__ __launch_bounds__(kBlock) void noise_pair_means(int n1, int n2, SpectrumArgs g, uint32_t key0,
                                                           uint32_t key1) {
  __shared__ double scratch[kBlock];
  const int n = n1 * n2, half = n2 >> 1;
  const int pair = blockIdx.x;
  const uint32_t series = g.series0 + pair;
  const float amp = g.w_ind * sqrtf(g.knee);
  double ar = 0.0, ai = 0.0;
  const int span = g.win.k_cut - g.win.k_min;
  for (int idx = threadIdx.x; idx < 2 * span; idx += kBlock) {
    const int kk = g.win.k_min + (idx >> 1);
    if ((idx & 1) && 2 * kk == n) continue;  // the self-conjugate cell once
    const int k = (idx & 1) ? n - kk : kk;
    const int k1 = k & (n1 - 1), k2 = k / n1;
    const int k2c = k2 < half ? k2 : k2 - half;
    const U4 rnd = philox4x32_10(U4{(uint32_t)k1, (uint32_t)k2c, series, kTagPink}, key0, key1);
    const float2 gq = k2 < half ? box_muller(rnd.x, rnd.y) : box_muller(rnd.z, rnd.w);
    const float a = pink_amp(k, n, amp, g.win.k_min);
    const double xr = (double)(a * gq.x), xi = (double)(a * gq.y);
    const double2 f = window_mean_factor(k, n, g.win.T);
    ar += xr * f.x - xi * f.y;
    ai += xr * f.y + xi * f.x;
  }
  const double sr = block_sum(ar, scratch);
  __syncthreads();
  const double si = block_sum(ai, scratch);
  if (threadIdx.x == 0) {
    double mr = sr, mi = si;
    const int row_a = g.row0 + 2 * pair;
    const bool has_b = row_a + 1 < g.row0 + g.rows;
    for (int m = 0; m < g.n_modes; ++m) {  // sqrt(c) (B[a,m] + i B[b,m]) mu_m, mu_m real
      mr += (double)(g.w_corr * g.basis[(size_t)row_a * g.n_modes + m]) * g.win.mu[m];
      if (has_b) mi += (double)(g.w_corr * g.basis[(size_t)(row_a + 1) * g.n_modes + m]) * g.win.mu[m];
    }
    g.win.mean[pair] = make_double2(mr, mi);
  }
}

// pass 1: block (group of kPairsPerBlock series, k1): spectrum cells k = k1 + N1*k2, FFT over
// k2, twiddle.  The mode spectra of the block's cells are read once into registers and serve
// every series of the group (they are the same for all detectors; reading them per series
// makes the kernel L2-bound).  kIter = max(1, n2 / 512) cell pairs per thread; kModes = the
// number of modes, or -1 for "any" (read per series, no register copy).
constexpr int kPairsPerBlock = 4;

// kThreads = 256.  The two LDS images of a long first transform (72 KiB at n2 = 4096) admit two
// workgroups per CU; 1024-thread workgroups behind the same footprint (8 waves per SIMD) measured
// 8 % SLOWER (15.6 vs 14.5 ms at 10 000 x 240 000): the kernel is bound by instruction count
// (~147 VALU per cell: Philox ~50, Box-Muller ~30, modes ~20, transform ~55, final twiddle ~15;
// VALU pipe ~60 % busy at two waves per SIMD), not by latency, and wider barriers cost more.
template <int kIter, int kModes, int kThreads>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(kIter <= 8 && kModes >= 0 ? 2 : 1, 2))) void noise_spectrum_fft(
    float2* __restrict__ A, int n1, int n2, int log2n2, SpectrumArgs g, int pairs, uint32_t key0,
    uint32_t key1) {
  extern __shared__ float2 lds2[];
  float2* data = lds2;
  float2* tw = lds2 + 2 * n2;
  const int k1 = blockIdx.y;
  const int n = n1 * n2;  // <= 2^23
  const float amp = g.w_ind * sqrtf(g.knee);
  const float inv_n = 1.0f / (float)n;
  fill_twiddles<kThreads>(tw, n2);
  const int half = n2 >> 1;
  constexpr int kRegModes = kModes > 0 ? kModes : 1;
  float2 f0[kIter][kRegModes], f1[kIter][kRegModes];
  const float2* F = g.F + (size_t)k1 * n2;
  if constexpr (kModes > 0) {
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
      const int k2 = threadIdx.x + it * kThreads;
#pragma unroll
      for (int m = 0; m < kModes; ++m) {
        f0[it][m] = k2 < half ? F[(size_t)m * n + k2] : make_float2(0.f, 0.f);
        f1[it][m] = k2 < half ? F[(size_t)m * n + k2 + half] : make_float2(0.f, 0.f);
      }
    }
  }
  const int n_modes = kModes >= 0 ? kModes : g.n_modes;
  // the same for every series of the block: the merged amplitudes of the thread's cells, which of
  // them lie below k_cut, the first of the thread's final twiddles exp(2 pi i k1 j / N)
  // (j = tid + i kThreads: the others by multiplication with the uniform step)
  float am0[kIter], am1[kIter];
  uint32_t low = 0;
  const float pink_var = amp * amp;
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const int k2 = threadIdx.x + it * kThreads;
    const int ka = k1 + n1 * k2, kb = k1 + n1 * (k2 + half);
    am0[it] = merged_amp(ka, n, g.white_var, pink_var, g.win.k_min);
    am1[it] = merged_amp(kb, n, g.white_var, pink_var, g.win.k_min);
    if (k2 < half) {
      if ((ka < n - ka ? ka : n - ka) < g.win.k_cut) low |= 1u << (2 * it);
      if ((kb < n - kb ? kb : n - kb) < g.win.k_cut) low |= 2u << (2 * it);
    }
  }
  const float sw = __builtin_sqrtf(g.white_var);
  float2 tw_first, tw_step;
  {
    // k1 * j < N <= 2^23 is exact in float32, and so is the fraction of a revolution the
    // hardware sine and cosine take (absolute error ~1e-6)
    const float r0 = (float)(k1 * (int)threadIdx.x) * inv_n, r1 = (float)(k1 * kThreads) * inv_n;
    tw_first = make_float2(__builtin_amdgcn_cosf(r0), __builtin_amdgcn_sinf(r0));
    tw_step = make_float2(__builtin_amdgcn_cosf(r1), __builtin_amdgcn_sinf(r1));
  }
  for (int p = 0; p < kPairsPerBlock; ++p) {
    const int pair = blockIdx.x * kPairsPerBlock + p;
    if (pair >= pairs) break;  // uniform
    const uint32_t series = g.series0 + pair;
    // the pair's mode coefficients sqrt(c) (B[a,m] + i B[b,m])
    float2 coef[kMaxModes];
    const int row_a = g.row0 + 2 * pair;
    const bool has_b = row_a + 1 < g.row0 + g.rows;
#pragma unroll
    for (int m = 0; m < kMaxModes; ++m)
      coef[m] = m < n_modes ? make_float2(g.w_corr * g.basis[(size_t)row_a * n_modes + m],
                                          has_b ? g.w_corr * g.basis[(size_t)(row_a + 1) * n_modes + m] : 0.0f)
                            : make_float2(0.0f, 0.0f);
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
      const int k2 = threadIdx.x + it * kThreads;
      if (k2 < half) {
        const U4 rnd = philox4x32_10(U4{(uint32_t)k1, (uint32_t)k2, series, kTagPink}, key0, key1);
        const float2 g0 = box_muller(rnd.x, rnd.y), g1 = box_muller(rnd.z, rnd.w);
        float2 x0 = make_float2(am0[it] * g0.x, am0[it] * g0.y), x1 = make_float2(am1[it] * g1.x, am1[it] * g1.y);
        if ((low >> (2 * it)) & 3u) {
          // below k_cut the pink part keeps its own draw (noise_pair_means redraws exactly it)
          // and the white part gets another
          const U4 rw = philox4x32_10(U4{(uint32_t)k1, (uint32_t)k2, series, kTagOwnWhite}, key0, key1);
          const float2 h0 = box_muller(rw.x, rw.y), h1 = box_muller(rw.z, rw.w);
          if ((low >> (2 * it)) & 1u) {
            const float a0 = pink_amp(k1 + n1 * k2, n, amp, g.win.k_min);
            x0 = make_float2(a0 * g0.x + sw * h0.x, a0 * g0.y + sw * h0.y);
          }
          if ((low >> (2 * it)) & 2u) {
            const float a1 = pink_amp(k1 + n1 * (k2 + half), n, amp, g.win.k_min);
            x1 = make_float2(a1 * g1.x + sw * h1.x, a1 * g1.y + sw * h1.y);
          }
        }
        if constexpr (kModes > 0) {
#pragma unroll
          for (int m = 0; m < kModes; ++m) {
            x0 = cfma(coef[m], f0[it][m], x0);
            x1 = cfma(coef[m], f1[it][m], x1);
          }
        } else if constexpr (kModes < 0) {
#pragma unroll
          for (int m = 0; m < kMaxModes; ++m)
            if (m < n_modes) {
              x0 = cfma(coef[m], F[(size_t)m * n + k2], x0);
              x1 = cfma(coef[m], F[(size_t)m * n + k2 + half], x1);
            }
        }
        data[k2] = x0;
        data[k2 + half] = x1;
      }
    }
    __syncthreads();
    // n2 = 2 kThreads kIter whenever kIter > 1: a compile-time length unrolls the butterfly loops
    constexpr int kN = kIter > 1 ? 2 * kThreads * kIter : 0;
    const float2* res = fft_lds_inverse<4, kN, kThreads>(data, data + n2, tw, n2, log2n2);
    float2* dst = A + ((size_t)pair * n1 + k1) * n2;
    float2 w = tw_first;
#pragma unroll
    for (int i = 0; i < 2 * kIter; ++i) {
      const int j = threadIdx.x + i * kThreads;
      if (j < n2) dst[j] = cmul(res[j], w);
      w = cmul(w, tw_step);
    }
    __syncthreads();  // the next series reuses both LDS images
  }
}

typedef void (*SpectrumKernel)(float2*, int, int, int, SpectrumArgs, int, uint32_t, uint32_t);

template <int kIter, int kThreads>
SpectrumKernel spectrum_kernel_modes(int n_modes) {
  switch (n_modes) {
    case 0: return noise_spectrum_fft<kIter, 0, kThreads>;
    case 1: return noise_spectrum_fft<kIter, 1, kThreads>;
    case 5: return noise_spectrum_fft<kIter, 5, kThreads>;
    default: return noise_spectrum_fft<kIter, -1, kThreads>;
  }
}

// the instantiation for a first transform of length n2 (64 .. 8192) and n_modes modes, and its
// workgroup size
SpectrumKernel spectrum_kernel(int n2, int n_modes, int* threads) {
  *threads = kBlock;
  switch (n2 <= 512 ? 1 : n2 / 512) {
    case 1: return spectrum_kernel_modes<1, kBlock>(n_modes);
    case 2: return spectrum_kernel_modes<2, kBlock>(n_modes);
    case 4: return spectrum_kernel_modes<4, kBlock>(n_modes);
    case 8: return spectrum_kernel_modes<8, kBlock>(n_modes);
    default: return spectrum_kernel_modes<16, kBlock>(n_modes);
  }
}

// pass 1 for n2 = 256 RB, RB = 4, 8, 16 (periods of 2^16, 2^17 and 2^18 ... 2^22 samples): a team of
// T = 16 RB threads takes a series; a thread builds its 16 cells k2 = t + T b in registers -- they
// are the inputs of the first radix-16 transform of fft_regs -- so the spectrum never goes through
// LDS before the transform.  The cells of all kPairsPerBlock series of a team are built first, cell
// by cell, so that a cell's mode spectra are read once and dropped (16 cells x kModes spectra do
// not fit the register file beside the data); then the series are transformed one after the
// other.  A workgroup holds 16 / RB teams on the same k1.  Same draws, same cells as
// noise_spectrum_fft (the two agree to rounding).
template <int kModes, int RB>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 2))) void noise_spectrum_r16(
    float2* __restrict__ A, int n1, SpectrumArgs g, int pairs, uint32_t key0, uint32_t key1) {
  extern __shared__ float2 lds2[];
  constexpr int T = 16 * RB, kTeams = kBlock / T, n2 = 256 * RB, half = n2 / 2;
  const int team = threadIdx.x / T, t = threadIdx.x % T;
  float2* ex1 = lds2 + (size_t)team * 2 * kFft4096Pitch * T;
  float2* ex2 = ex1 + kFft4096Pitch * T;
  const int k1 = blockIdx.y;
  const int n = n1 * n2;
  const float amp = g.w_ind * sqrtf(g.knee);
  const float pink_var = amp * amp;
  const float sw = __builtin_sqrtf(g.white_var);
  const int pair0 = (blockIdx.x * kTeams + team) * kPairsPerBlock;
  constexpr int kRegModes = kModes > 0 ? kModes : 1;
  // the pairs' mode coefficients sqrt(c) (B[a,m] + i B[b,m]) go through LDS: as scalar loads inside
  // the cell loop each is a stall of its full latency (four in a row per cell and pair: a quarter
  // of the kernel's time), and hoisted into registers they cost more than the file has left
  __shared__ float2 coef[kTeams][kPairsPerBlock][kMaxModes];
  if constexpr (kModes > 0) {
    if ((int)threadIdx.x < kTeams * kPairsPerBlock * kModes) {
      const int q = threadIdx.x / kModes, m = threadIdx.x - q * kModes;  // q = team * kPairsPerBlock + p
      const int want = blockIdx.x * kTeams * kPairsPerBlock + q;
      const int pair = want < pairs ? want : pairs - 1;
      const int row_a = g.row0 + 2 * pair;
      const bool has_b = row_a + 1 < g.row0 + g.rows;
      coef[q / kPairsPerBlock][q % kPairsPerBlock][m] =
          make_float2(g.w_corr * g.basis[(size_t)row_a * kModes + m],
                      has_b ? g.w_corr * g.basis[(size_t)(row_a + 1) * kModes + m] : 0.0f);
    }
    __syncthreads();
  }
  float2 v[kPairsPerBlock][16];
  // the mode spectra of cells b + 1 are fetched while the cells b are built: at two waves per
  // SIMD a load waited for on the spot is a stall of its full latency, eight times per block
  float2 f0[kRegModes], f1[kRegModes], nf0[kRegModes], nf1[kRegModes];
  const float2* Fk = g.F + (size_t)k1 * n2 + t;
  if constexpr (kModes > 0) {
#pragma unroll
    for (int m = 0; m < kModes; ++m) {
      nf0[m] = Fk[(size_t)m * n];
      nf1[m] = Fk[(size_t)m * n + half];
    }
  }
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int k2 = t + T * b;  // < n2 / 2 = 8 T
    const int ka = k1 + n1 * k2, kb = k1 + n1 * (k2 + half);
    const float am0 = merged_amp(ka, n, g.white_var, pink_var, g.win.k_min);
    const float am1 = merged_amp(kb, n, g.white_var, pink_var, g.win.k_min);
    const bool low0 = (ka < n - ka ? ka : n - ka) < g.win.k_cut, low1 = (kb < n - kb ? kb : n - kb) < g.win.k_cut;
    if constexpr (kModes > 0) {
#pragma unroll
      for (int m = 0; m < kModes; ++m) {
        f0[m] = nf0[m];
        f1[m] = nf1[m];
        if (b + 1 < 8) {
          nf0[m] = Fk[(size_t)m * n + T * (b + 1)];
          nf1[m] = Fk[(size_t)m * n + T * (b + 1) + half];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < kPairsPerBlock; ++p) {
      // (this branch also keeps the scheduler from hoisting every cell's loads and draws to the top:
      // without it the kernel spills 390 registers)
      const int pair = pair0 + p;
      if (pair >= pairs) break;  // uniform over the team's waves; no barrier inside this loop
      const uint32_t series = g.series0 + pair;
      const U4 rnd = philox4x32_10(U4{(uint32_t)k1, (uint32_t)k2, series, kTagPink}, key0, key1);
      const float2 g0 = box_muller(rnd.x, rnd.y), g1 = box_muller(rnd.z, rnd.w);
      float2 x0 = make_float2(am0 * g0.x, am0 * g0.y), x1 = make_float2(am1 * g1.x, am1 * g1.y);
      if (low0 || low1) {
        // below k_cut the pink part keeps its own draw (noise_pair_means redraws exactly it)
        const U4 rw = philox4x32_10(U4{(uint32_t)k1, (uint32_t)k2, series, kTagOwnWhite}, key0, key1);
        const float2 h0 = box_muller(rw.x, rw.y), h1 = box_muller(rw.z, rw.w);
        if (low0) {
          const float a0 = pink_amp(ka, n, amp, g.win.k_min);
          x0 = make_float2(a0 * g0.x + sw * h0.x, a0 * g0.y + sw * h0.y);
        }
        if (low1) {
          const float a1 = pink_amp(kb, n, amp, g.win.k_min);
          x1 = make_float2(a1 * g1.x + sw * h1.x, a1 * g1.y + sw * h1.y);
        }
      }
      if constexpr (kModes > 0) {
#pragma unroll
        for (int m = 0; m < kModes; ++m) {
          x0 = cfma(coef[team][p][m], f0[m], x0);
          x1 = cfma(coef[team][p][m], f1[m], x1);
        }
      }
      v[p][b] = x0;
      v[p][b + 8] = x1;
    }
  }
  const float inv_n = 1.0f / (float)n;
  const float r0 = (float)(k1 * t) * inv_n, r1 = (float)(k1 * T) * inv_n;  // k1 j < N <= 2^23: exact
  const float2 tw_first = make_float2(__builtin_amdgcn_cosf(r0), __builtin_amdgcn_sinf(r0));
  const float2 tw_step = make_float2(__builtin_amdgcn_cosf(r1), __builtin_amdgcn_sinf(r1));
#pragma unroll
  for (int p = 0; p < kPairsPerBlock; ++p) {
    const int pair = pair0 + p;
    fft_regs<RB>(v[p], ex1, ex2, t);  // every thread of the workgroup reaches its barriers: a team past the end transforms leftovers
    if (pair >= pairs) continue;
    float2* dst = A + ((size_t)pair * n1 + k1) * n2 + t;
    float2 w = tw_first;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      dst[T * f] = cmul(v[p][dft16_pos(f)], w);
      w = cmul(w, tw_step);
    }
  }
}

typedef void (*SpectrumR16Kernel)(float2*, int, SpectrumArgs, int, uint32_t, uint32_t);

template <int RB>
SpectrumR16Kernel spectrum_r16_modes(int n_modes) {
  switch (n_modes) {
    case 0: return noise_spectrum_r16<0, RB>;
    case 1: return noise_spectrum_r16<1, RB>;
    case 2: return noise_spectrum_r16<2, RB>;
    case 3: return noise_spectrum_r16<3, RB>;
    case 4: return noise_spectrum_r16<4, RB>;
    case 5: return noise_spectrum_r16<5, RB>;
    default: return nullptr;  // more modes than registers: the Stockham kernel
  }
}

// the register-transform instance for a first transform of n2 points, or null
SpectrumR16Kernel spectrum_r16_kernel(int n2, int n_modes) {
  switch (n2) {
    case 1024: return spectrum_r16_modes<4>(n_modes);
    case 2048: return spectrum_r16_modes<8>(n_modes);
    case 4096: return spectrum_r16_modes<16>(n_modes);
    default: return nullptr;
  }
}

struct CombineArgs {
  const float* scale;    // [rows] or null
  const float* loading;  // [rows][ld_loading] or null
  size_t ld_loading;
  float per_loading;
  float* out;            // [rows][ld]
  size_t ld;
  int row0, rows;        // this launch writes rows row0 .. row0 + rows - 1
  uint32_t id0;          // white-noise id of out row 0
  int T;
  float sqrt_fs;
  int accumulate, vec_ok;
  const double2* mean;   // [pairs of this launch] window mean of the pink part, or null
};

// one group of 4 consecutive samples of one row, given its pink + correlated part
template <bool kDrawWhite>
__device__ __forceinline__ void finish4(const CombineArgs& g, int row, size_t t0, vfloat4 pink,
                                        uint32_t key0, uint32_t key1) {
  vfloat4 v = pink;  // from the spectrum: the white part is in it
  if constexpr (kDrawWhite) v += g.sqrt_fs * white4(t0 >> 2, g.id0 + (uint32_t)row, key0, key1);
  const float sc = g.scale ? g.scale[row] : 1.0f;
  float* dst = g.out + (size_t)row * g.ld + t0;
  const bool full = g.vec_ok && t0 + 4 <= (size_t)g.T;
  if (g.loading) {
    // total NEP of a sample: NEP + NEP_per_loading x loading (sim/noise.py:35-37)
    const float* L = g.loading + (size_t)row * g.ld_loading + t0;
    const float l0 = L[0], l1 = t0 + 1 < (size_t)g.T ? L[1] : 0.f, l2 = t0 + 2 < (size_t)g.T ? L[2] : 0.f,
                l3 = t0 + 3 < (size_t)g.T ? L[3] : 0.f;
    v *= sc + g.per_loading * vfloat4{l0, l1, l2, l3};
  } else {
    v *= sc;
  }
  if (full) {
    vfloat4* d4 = reinterpret_cast<vfloat4*>(dst);
    if (g.accumulate) v += *d4;
    __builtin_nontemporal_store(v, d4);
  } else {
    const float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    dst[0] = g.accumulate ? dst[0] + v0 : v0;
    if (t0 + 1 < (size_t)g.T) dst[1] = g.accumulate ? dst[1] + v1 : v1;
    if (t0 + 2 < (size_t)g.T) dst[2] = g.accumulate ? dst[2] + v2 : v2;
    if (t0 + 3 < (size_t)g.T) dst[3] = g.accumulate ? dst[3] + v3 : v3;
  }
}

// pass 2, general: block (tile of J adjacent j, series): FFT over k1 for J sequences at
// once in LDS, then the epilogue with 16-byte stores of 4 consecutive samples.
__global__ __launch_bounds__(kBlock) void noise_fft_combine(
    const float2* __restrict__ A, int n1, int n2, int log2n1, int lj, CombineArgs g,
    uint32_t key0, uint32_t key1) {
  extern __shared__ float2 lds2[];
  float2* data = lds2;
  float2* tw = lds2 + 2 * kTileCells;
  const int J = 1 << lj;
  const int j0 = blockIdx.x << lj;
  // a tile whose first sample is past the end produces nothing: t = j + n2 m >= j0
  if ((size_t)j0 >= (size_t)g.T) return;
  fill_twiddles(tw, n1);
  const float2* src = A + (size_t)blockIdx.y * n1 * n2 + j0;
  for (int e = threadIdx.x; e < n1 * J; e += kBlock) {
    const int k1 = e >> lj, b = e & (J - 1);
    data[e] = src[(size_t)k1 * n2 + b];
  }
  __syncthreads();
  const float2* res = fft_lds_inverse_batched<8>(data, data + kTileCells, tw, n1, log2n1, lj);
  const int row_a = g.row0 + 2 * blockIdx.y;
  const float mean_a = g.mean ? (float)g.mean[blockIdx.y].x : 0.0f, mean_b = g.mean ? (float)g.mean[blockIdx.y].y : 0.0f;
  for (int e = threadIdx.x * 4; e < n1 * J; e += kBlock * 4) {
    const int m = e >> lj, b = e & (J - 1);
    const size_t t0 = (size_t)n2 * m + j0 + b;
    if (t0 >= (size_t)g.T) continue;
    const float2 r0 = res[e], r1 = res[e + 1], r2 = res[e + 2], r3 = res[e + 3];
    finish4<false>(g, row_a, t0, vfloat4{r0.x, r1.x, r2.x, r3.x} - mean_a, key0, key1);
    if (row_a + 1 < g.row0 + g.rows)
      finish4<false>(g, row_a + 1, t0, vfloat4{r0.y, r1.y, r2.y, r3.y} - mean_b, key0, key1);
  }
}

// pass 2 when n1 == 64 (every period up to 2^19 samples): one thread per j, the
// 64-point transform over k1 in registers -- no LDS, no barrier.  Lane l of a wave
// holds samples t = j0 + l + n2 m, so each (m, row) is one 256-byte store per wave.
#ifndef MRX_FFT64_WAVES
#define MRX_FFT64_WAVES 2  // waves per SIMD of the 64-point register pass: 256 registers for the 64 complex values and the transform's temporaries
#endif
template <bool kExtras>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MRX_FFT64_WAVES, MRX_FFT64_WAVES))) void noise_fft64_combine(
    const float2* __restrict__ A, int n2, CombineArgs g, uint32_t key0, uint32_t key1) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= n2 || j >= g.T) return;
  const float2* src = A + (size_t)blockIdx.y * 64 * n2 + j;
  float re[64], im[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    const vfloat2 v = __builtin_nontemporal_load(reinterpret_cast<const vfloat2*>(src + (size_t)k * n2));
    re[k] = v[0];
    im[k] = v[1];
  }
  fft64_inverse_reg(re, im);
  const int row_a = g.row0 + 2 * blockIdx.y;
  const bool has_b = row_a + 1 < g.row0 + g.rows;
  const int row_b = has_b ? row_a + 1 : row_a;
  const float sa = g.scale ? g.scale[row_a] : 1.0f, sb = g.scale ? g.scale[row_b] : 1.0f;
  float* out_a = g.out + (size_t)row_a * g.ld + j;
  float* out_b = g.out + (size_t)row_b * g.ld + j;
  const float mean_a = g.mean ? (float)g.mean[blockIdx.y].x : 0.0f, mean_b = g.mean ? (float)g.mean[blockIdx.y].y : 0.0f;
#pragma unroll
  for (int m = 0; m < 64; ++m) {
    const size_t t = (size_t)j + (size_t)n2 * m;
    if (t >= (size_t)g.T) continue;
    float va = re[fft64_pos(m)] - mean_a;
    float vb = im[fft64_pos(m)] - mean_b;
    const size_t o = (size_t)n2 * m;
    if (kExtras) {
      float aa = sa, ab = sb;
      if (g.loading) {
        aa += g.per_loading * g.loading[(size_t)row_a * g.ld_loading + t];
        ab += g.per_loading * g.loading[(size_t)row_b * g.ld_loading + t];
      }
      va *= aa;
      vb *= ab;
      if (g.accumulate) {
        va += out_a[o];
        vb += out_b[o];
      }
      out_a[o] = va;
      if (has_b) out_b[o] = vb;
    } else {
      __builtin_nontemporal_store(sa * va, out_a + o);
      if (has_b) __builtin_nontemporal_store(sb * vb, out_b + o);
    }
  }
}

// pass 2 when n1 = 64 m, m = 2, 4, 8, 16 (periods of 2^19 ... 2^23): k1 = a + m b, sample block
// mm = c + 64 d:  exp(2 pi i k1 mm / n1) = exp(2 pi i a c / n1) exp(2 pi i b c / 64) exp(2 pi i a d / m).
//   stage a  per (j, a): the 64-point register transform over b of the rows a + m b, the twiddle
//            exp(2 pi i a c / n1), back into the rows a + m c -- in place: a thread writes the rows
//            it read, lanes are consecutive j (512 contiguous bytes per row and wave);
//   stage b  per (j, c): the m-point transform over a of the rows a + m c (m consecutive rows),
//            then the epilogue of noise_fft64_combine for the samples t = j + n2 (c + 64 d).
// Three sweeps over the scratch instead of one, but every access whole lines: the LDS form's
// tiles are 4096 / n1 adjacent j wide (64-byte rows at n1 = 512) and it took 72 % of the time.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 2))) void noise_fft64_rows(
    float2* __restrict__ A, int n2, int m, int n1, int j_used) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= j_used) return;
  const int a = blockIdx.y;
  float2* base = A + ((size_t)blockIdx.z * n1 + a) * n2 + j;
  const size_t stride = (size_t)m * n2;
  float re[64], im[64];
#pragma unroll
  for (int b = 0; b < 64; ++b) {
    const vfloat2 v = *reinterpret_cast<const vfloat2*>(base + (size_t)b * stride);
    re[b] = v[0];
    im[b] = v[1];
  }
  fft64_inverse_reg(re, im);
  const float step = (float)a / (float)n1;  // a c / n1 < 64 / 128: exact enough in float32 (a c < 2^10)
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    const float rev = (float)c * step;
    const float2 w = make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev));
    const float2 y = cmul(make_float2(re[fft64_pos(c)], im[fft64_pos(c)]), w);
    *reinterpret_cast<vfloat2*>(base + (size_t)c * stride) = vfloat2{y.x, y.y};
  }
}

template <int kM, bool kExtras>
__global__ __launch_bounds__(kBlock) void noise_combine_rows(const float2* __restrict__ A, int n2, CombineArgs g) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= n2 || j >= g.T) return;
  const int c = blockIdx.y;
  const float2* src = A + ((size_t)blockIdx.z * 64 * kM + (size_t)kM * c) * n2 + j;
  float2 v[16];
#pragma unroll
  for (int a = 0; a < 16; ++a) v[a] = make_float2(0.f, 0.f);
#pragma unroll
  for (int a = 0; a < kM; ++a) {
    const vfloat2 q = __builtin_nontemporal_load(reinterpret_cast<const vfloat2*>(src + (size_t)a * n2));
    v[a] = make_float2(q[0], q[1]);
  }
  // the kM-point transform over a: output d at v[pos(d)]
  if constexpr (kM == 16) {
    dft16(v);
  } else if constexpr (kM == 8) {
    float2 u[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) u[a] = v[a];
    dft8(u);
#pragma unroll
    for (int a = 0; a < 8; ++a) v[a] = u[a];
  } else if constexpr (kM == 4) {
    radix4_inverse(v[0], v[1], v[2], v[3]);
  } else {
    const float2 s0 = cadd(v[0], v[1]), s1 = csub(v[0], v[1]);
    v[0] = s0;
    v[1] = s1;
  }
  const int row_a = g.row0 + 2 * blockIdx.z;
  const bool has_b = row_a + 1 < g.row0 + g.rows;
  const int row_b = has_b ? row_a + 1 : row_a;
  const float sa = g.scale ? g.scale[row_a] : 1.0f, sb = g.scale ? g.scale[row_b] : 1.0f;
  float* out_a = g.out + (size_t)row_a * g.ld;
  float* out_b = g.out + (size_t)row_b * g.ld;
  const float mean_a = g.mean ? (float)g.mean[blockIdx.z].x : 0.0f, mean_b = g.mean ? (float)g.mean[blockIdx.z].y : 0.0f;
#pragma unroll
  for (int d = 0; d < kM; ++d) {
    const size_t t = (size_t)j + (size_t)n2 * (size_t)(c + 64 * d);
    if (t >= (size_t)g.T) continue;
    const float2 x = v[kM == 16 ? dft16_pos(d) : d];
    float va = x.x - mean_a, vb = x.y - mean_b;
    if (kExtras) {
      float aa = sa, ab = sb;
      if (g.loading) {
        aa += g.per_loading * g.loading[(size_t)row_a * g.ld_loading + t];
        ab += g.per_loading * g.loading[(size_t)row_b * g.ld_loading + t];
      }
      va *= aa;
      vb *= ab;
      if (g.accumulate) {
        va += out_a[t];
        vb += out_b[t];
      }
      out_a[t] = va;
      if (has_b) out_b[t] = vb;
    } else {
      __builtin_nontemporal_store(sa * va, out_a + t);
      if (has_b) __builtin_nontemporal_store(sb * vb, out_b + t);
    }
  }
}

// the modes' white parts as time series (two-rate form): unit normals, 4 samples per thread, mode = blockIdx.y
__global__ __launch_bounds__(kBlock) void noise_mode_white_kernel(float* __restrict__ mw, size_t ld, int T, uint32_t key0, uint32_t key1) {
  const size_t t0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 4;
  if (t0 >= (size_t)T) return;
  const U4 rnd = philox4x32_10(U4{(uint32_t)(t0 >> 2), blockIdx.y, 0u, kMrxTagModeWhite}, key0, key1);
  const float2 a = box_muller(rnd.x, rnd.y), b = box_muller(rnd.z, rnd.w);
  *reinterpret_cast<vfloat4*>(mw + (size_t)blockIdx.y * ld + t0) = vfloat4{a.x, a.y, b.x, b.y};  // (ld: whole groups of 4)
}

// knee = 0: white noise only (generation.py:25), 4 samples per thread
__global__ __launch_bounds__(kBlock) void noise_white_kernel(CombineArgs g, uint32_t key0, uint32_t key1) {
  const size_t t0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 4;
  if (t0 >= (size_t)g.T) return;
  finish4<true>(g, g.row0 + blockIdx.y, t0, vfloat4{0.f, 0.f, 0.f, 0.f}, key0, key1);
}

// test hook: the in-LDS transforms on caller data (rows of n complex values;
// lj > 0: rows of n << lj values holding 2^lj interleaved sequences)
__global__ __launch_bounds__(kBlock) void fft_rows_kernel(const float2* __restrict__ in,
                                                          float2* __restrict__ out, int n,
                                                          int log2n, int lj) {
  extern __shared__ float2 lds2[];
  const int cells = n << lj;
  float2* data = lds2;
  float2* tw = lds2 + 2 * cells;
  fill_twiddles(tw, n);
  for (int e = threadIdx.x; e < cells; e += kBlock) data[e] = in[(size_t)blockIdx.x * cells + e];
  __syncthreads();
  const float2* res = lj ? fft_lds_inverse_batched(data, data + cells, tw, n, log2n, lj)
                         : fft_lds_inverse(data, data + cells, tw, n, log2n);
  for (int e = threadIdx.x; e < cells; e += kBlock) out[(size_t)blockIdx.x * cells + e] = res[e];
}

__global__ __launch_bounds__(kBlock) void fft4096_rows_kernel(const float2* __restrict__ in, float2* __restrict__ out) {
  extern __shared__ float2 lds2[];
  float2 v[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) v[b] = in[(size_t)blockIdx.x * 4096 + threadIdx.x + 256 * b];
  fft4096_workgroup(v, lds2, lds2 + kFft4096Image);
#pragma unroll
  for (int f = 0; f < 16; ++f) out[(size_t)blockIdx.x * 4096 + threadIdx.x + 256 * f] = v[dft16_pos(f)];
}

// test hook of fft_regs<RB>: rows of 256 RB values, 16 / RB rows per workgroup
template <int RB>
__global__ __launch_bounds__(kBlock) void fft_regs_rows_kernel(const float2* __restrict__ in, float2* __restrict__ out, int rows) {
  extern __shared__ float2 lds2[];
  constexpr int T = 16 * RB, n = 256 * RB;
  const int which = threadIdx.x / T, t = threadIdx.x % T;
  const int row = blockIdx.x * (kBlock / T) + which;
  const bool live = row < rows;
  float2* ex1 = lds2 + which * 2 * kFft4096Pitch * T;
  float2 v[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) v[b] = live ? in[(size_t)row * n + t + T * b] : make_float2(0.f, 0.f);
  fft_regs<RB>(v, ex1, ex1 + kFft4096Pitch * T, t);
  if (live) {
#pragma unroll
    for (int f = 0; f < 16; ++f) out[(size_t)row * n + t + T * f] = v[dft16_pos(f)];
  }
}

__global__ __launch_bounds__(64) void fft64_reg_rows_kernel(const float2* __restrict__ in, float2* __restrict__ out, int rows) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float re[64], im[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    re[k] = in[(size_t)r * 64 + k].x;
    im[k] = in[(size_t)r * 64 + k].y;
  }
  fft64_inverse_reg(re, im);
#pragma unroll
  for (int k = 0; k < 64; ++k) out[(size_t)r * 64 + k] = make_float2(re[fft64_pos(k)], im[fft64_pos(k)]);
}

int ilog2(long long n) {
  int l = 0;
  while ((1LL << l) < n) ++l;
  return (1LL << l) == n ? l : -1;
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
    krj_stage_rows(cdet, red, a.dx, a.dy, a.band, nullptr, a.n_bands, a.row0 + r0, nd, cal_cells, a.n_el);
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
  // Catmull-Rom weights of (P[-1], P[0], P[1], P[2]) at u = 1/4, 1/2, 3/4 (u = 0: P[0] itself)
  constexpr float kW14[4] = {-0.0703125f, 0.8671875f, 0.2265625f, -0.0234375f};
  constexpr float kW12[4] = {-0.0625f, 0.5625f, 0.5625f, -0.0625f};
  constexpr float kW34[4] = {-0.0234375f, 0.2265625f, 0.8671875f, -0.0703125f};
  // The rows one after the other, the NEXT row's slow samples (and its scale) fetched while this row is computed: the loop
  // does not unroll (the compiler says so, six instances out of six), and with the load at the top of every iteration a wave
  // sat out a memory latency per row -- its waves waited three quarters of their cycles while the white-only path, the same
  // draws and stores without the load, keeps the vector ALU busy (profiles/r05_noise_mix.txt).
  const int slow_at = a.rate == 4 ? (sb >> 2) : (sb >> 1);
  auto fetch_slow = [&](int dl, nvfloat4u& q, float& q4, float& sc) {
    const float* lo = a.lo + (size_t)(r0 + dl) * a.ld_lo;
    q = *reinterpret_cast<const nvfloat4u*>(lo + slow_at);
    q4 = a.rate == 4 ? 0.0f : lo[slow_at + 4];
    sc = a.scale ? a.scale[a.row0 + r0 + dl] : 1.0f;
  };
  {
    nvfloat4u q_next;
    float q4_next, sc_next;
    fetch_slow(0, q_next, q4_next, sc_next);
    for (int dl = 0; dl < nd; ++dl) {
      const int row = a.row0 + r0 + dl;  // row of the call
      // the slow part: samples t' - 1 .. t' + 2 (.. t' + 3 at rate 2) around the thread's interval(s), stored one to the right
      const nvfloat4u q = q_next;
      const float q4 = q4_next, sc = sc_next;
      if (dl + 1 < nd) fetch_slow(dl + 1, q_next, q4_next, sc_next);  // (uniform)
      float p[kSamplesPerThread];
      if (a.rate == 4) {
        const float p0 = q[0], p1 = q[1], p2 = q[2], p3 = q[3];
        p[0] = p1;
        p[1] = kW14[0] * p0 + kW14[1] * p1 + kW14[2] * p2 + kW14[3] * p3;
        p[2] = kW12[0] * p0 + kW12[1] * p1 + kW12[2] * p2 + kW12[3] * p3;
        p[3] = kW34[0] * p0 + kW34[1] * p1 + kW34[2] * p2 + kW34[3] * p3;
      } else {
        const float p0 = q[0], p1 = q[1], p2 = q[2], p3 = q[3], p4 = q4;
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
        krj_row<false>(c, cal_cells + c.band * (a.n_el - 1), a.n_el, ks, sv, o, a.bore_el, sb, a.T, cal_cells, a.cal_axis);
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

int mrx_noise_period(int T, int* n1, int* n2) {
  // smallest power of two N = n1*n2 >= max(T, 4096).  n2 is the length of the first transform (one
  // workgroup), n1 = N/n2 <= 1024 the second's: N/64 up to 2^18; from there 4096 -- the length the
  // register transform of pass 1 takes (the 8192-point Stockham pass spills: 69 against 200-300 G
  // samples/s) -- and 8192 only for the longest period, 2^23
  if (T <= 0 || !n1 || !n2) return MRX_ERR_INVALID;
  int l = 12;
  while (l < 23 && (1LL << l) < (long long)T) ++l;
  if ((1LL << l) < (long long)T) return MRX_ERR_UNSUPPORTED;
  const int l2 = l <= 18 ? l - 6 : l <= 22 ? 12 : 13;
  *n2 = 1 << l2;
  *n1 = 1 << (l - l2);
  return MRX_OK;
}

constexpr int kTwoRateMinT = 32768;  // shorter series keep the one-rate form (two_rate_factor)

int mrx_noise_work_floats(int T, int n_modes, int batch, size_t* floats) {
  int n1, n2;
  int rc = mrx_noise_period(T, &n1, &n2);
  if (rc != MRX_OK || !floats || batch < 1 || n_modes < 0) return rc != MRX_OK ? rc : MRX_ERR_INVALID;
  const size_t pairs = ((size_t)batch + 1) / 2, m = (size_t)n_modes;
  // one-rate form: window means (8 doubles for the modes, a double2 per pair), mode spectra [n_modes][N], one
  // complex series [N] per pair of detectors
  size_t need = 32 + (2 * (size_t)n1 * n2 + 4) * pairs + 2 * (size_t)n1 * n2 * m;
  // two-rate form (noise_generate_impl picks it from the sample rate and the knee, which this function does not
  // see: the size covers every rate it can pick): the same for the slow series of T / rate + 4 samples, plus the
  // slow series themselves [2 pairs][ld_lo] and the modes' white table [n_modes][ld_mw].  The period is a power of
  // two, so for T at or just below one the slow series' period can equal the full one and this form needs MORE
  // than the one-rate form for batches of a few rows (a band or a shard with a handful of detectors: ADVICE r4).
  for (int rate = 2; rate <= 4 && T >= kTwoRateMinT; rate <<= 1) {
    const int Ts = (T + rate - 1) / rate + 4;
    int s1, s2;
    if (mrx_noise_period(Ts, &s1, &s2) != MRX_OK) continue;
    const size_t n = (size_t)s1 * s2, ld_lo = ((size_t)Ts + 3) & ~(size_t)3, ld_mw = ((size_t)T + 3) & ~(size_t)3;
    need = std::max(need, 32 + 2 * n * m + m * ld_mw + 16 + (2 * n + 4 + 2 * ld_lo) * pairs);
  }
  *floats = need;
  return MRX_OK;
}

// TOD.to("K_RJ") of the field as it is written (mrx_noise_generate_krj), arrays indexed by the call's rows
struct NoiseKrj {
  const float* bore_el;
  const float* dx;
  const float* dy;
  const int32_t* band;
  const float* cal_axis;
  const float* cal_values;
  int n_el, n_bands;
};

// The slow part's sample rate: fs / rate, rate the largest of 4, 2 for which the pink part at the slow Nyquist
// frequency, knee / (fs / (2 rate)), stays below 2 % of the white level (what the form leaves out above it; the
// interpolation's roll-off just below is smaller still) -- 4 at 400 Hz with a knee of 1 Hz, 1 (the one-rate form)
// at 50 Hz.  Short series and option bit 8 of MRX_OPT_NOISE_GENERIC keep the one-rate form.
static int two_rate_factor(const mrx_ctx* ctx, int T, double sample_rate, double knee) {
  if ((ctx->options[MRX_OPT_NOISE_GENERIC] & 8) || !(knee > 0.0) || T < kTwoRateMinT) return 1;
  for (int rate = 4; rate >= 2; rate >>= 1)
    if (2.0 * rate * knee / sample_rate <= 0.0205) return rate;  // (2 %, with room for a rate of 399.99 Hz read off a time axis)
  return 1;
}

static int noise_generate_impl(mrx_ctx* ctx, uint64_t seed, int D, int det_offset, int T,
                               double sample_rate, double knee, double corr_prop,
                               const float* d_basis, int n_modes, const float* d_scale,
                               const float* d_loading, size_t ld_loading, double per_loading,
                               float* d_out, size_t ld_out, int accumulate,
                               float* d_work, size_t work_floats, const NoiseKrj* krj);

int mrx_noise_generate(mrx_ctx* ctx, uint64_t seed, int D, int det_offset, int T,
                       double sample_rate, double knee, double corr_prop,
                       const float* d_basis, int n_modes, const float* d_scale,
                       const float* d_loading, size_t ld_loading, double per_loading,
                       float* d_out, size_t ld_out, int accumulate,
                       float* d_work, size_t work_floats) {
  MRX_ENTER(ctx);
  return noise_generate_impl(ctx, seed, D, det_offset, T, sample_rate, knee, corr_prop, d_basis, n_modes, d_scale, d_loading,
                             ld_loading, per_loading, d_out, ld_out, accumulate, d_work, work_floats, nullptr);
}

int mrx_noise_generate_krj(mrx_ctx* ctx, uint64_t seed, int D, int det_offset, int T,
                           double sample_rate, double knee, double corr_prop,
                           const float* d_basis, int n_modes, const float* d_scale,
                           const float* d_loading, size_t ld_loading, double per_loading,
                           float* d_out, size_t ld_out,
                           float* d_work, size_t work_floats,
                           const float* d_bore_el, const float* d_dx, const float* d_dy, const int32_t* d_band,
                           const float* d_cal_axis_el, const float* d_cal_values, int n_el, int n_bands) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, d_bore_el && d_dx && d_dy && d_band && d_cal_axis_el && d_cal_values, "null calibration pointer");
  const NoiseKrj k{d_bore_el, d_dx, d_dy, d_band, d_cal_axis_el, d_cal_values, n_el, n_bands};
  if (D > 0 && T > 0 && two_rate_factor(ctx, T, sample_rate, knee) > 1)  // the conversion rides on the two-rate form's writer
    return noise_generate_impl(ctx, seed, D, det_offset, T, sample_rate, knee, corr_prop, d_basis, n_modes, d_scale, d_loading,
                               ld_loading, per_loading, d_out, ld_out, 0, d_work, work_floats, &k);
  // otherwise: the field in pW, then mrx_tod_to_krj's pass over it
  int rc = noise_generate_impl(ctx, seed, D, det_offset, T, sample_rate, knee, corr_prop, d_basis, n_modes, d_scale, d_loading,
                               ld_loading, per_loading, d_out, ld_out, 0, d_work, work_floats, nullptr);
  if (rc != MRX_OK) return rc;
  return mrx_tod_to_krj(ctx, d_out, ld_out, D, T, nullptr, nullptr, d_bore_el, d_dx, d_dy, d_band, d_cal_axis_el, d_cal_values, n_el,
                        n_bands);
}

static int noise_generate_impl(mrx_ctx* ctx, uint64_t seed, int D, int det_offset, int T_full,
                               double sample_rate, double knee, double corr_prop,
                               const float* d_basis, int n_modes, const float* d_scale,
                               const float* d_loading, size_t ld_loading, double per_loading,
                               float* d_out, size_t ld_out, int accumulate,
                               float* d_work, size_t work_floats, const NoiseKrj* krj) {
  if (!ctx) return MRX_ERR_INVALID;
  int T = T_full;  // samples the spectral part is made for: the TOD's, or the slow series' of the two-rate form
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_out, "null pointer");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  MRX_REQUIRE(ctx, det_offset >= 0 && det_offset % 2 == 0 && (long long)det_offset + D < (1LL << 31),
              "det_offset must be even and non-negative");
  MRX_REQUIRE(ctx, !d_loading || ld_loading >= (size_t)T, "ld_loading smaller than T");
  MRX_REQUIRE(ctx, !d_loading || !accumulate || d_loading != d_out,
              "accumulating into the loading the noise level is read from");
  MRX_REQUIRE(ctx, sample_rate > 0 && knee >= 0 && corr_prop >= 0 && corr_prop <= 1,
              "need sample_rate > 0, knee >= 0, 0 <= corr_prop <= 1");
  MRX_REQUIRE(ctx, n_modes >= 0 && n_modes <= kMaxModes && (n_modes == 0 || d_basis),
              "0 <= n_modes <= 8 and a basis when n_modes > 0");
  const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
  const bool pink = knee > 0.0;
  const int rate = (D > 0 && T > 0) ? two_rate_factor(ctx, T_full, sample_rate, knee) : 1;
  const bool two_rate = rate > 1;
  MRX_REQUIRE(ctx, !krj || two_rate, "internal: the fused K_RJ conversion belongs to the two-rate form");

  CombineArgs g{};
  g.scale = d_scale;
  g.loading = d_loading;
  g.ld_loading = ld_loading;
  g.per_loading = (float)per_loading;
  g.out = d_out;
  g.ld = ld_out;
  g.rows = D;
  g.id0 = (uint32_t)det_offset;
  g.T = T;
  g.sqrt_fs = (float)sqrt(sample_rate);
  g.accumulate = accumulate;
  g.vec_ok = (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0);

  if (!pink) {  // white only: no scratch, no modes (the basis only enters the pink part)
    const int per_launch = 32768;
    for (int d0 = 0; d0 < D; d0 += per_launch) {
      CombineArgs h = g;
      h.row0 = d0;
      h.rows = D - d0 < per_launch ? D - d0 : per_launch;
      hipLaunchKernelGGL(noise_white_kernel, dim3(mrx_ceil_div(mrx_ceil_div(T, 4), kBlock), h.rows),
                         dim3(kBlock), 0, ctx->stream, h, key0, key1);
      MRX_CHECK_LAUNCH(ctx);
    }
    return MRX_OK;
  }

  MRX_REQUIRE(ctx, d_work && (reinterpret_cast<uintptr_t>(d_work) & 15u) == 0,
              "work buffer must be 16-byte aligned");
  // Two-rate form: the spectral machinery below makes the SLOW series -- pink parts only, T_full / rate + 4 samples
  // (one before and up to three after the TOD's span, for the interpolation) at fs / rate -- into a buffer of their
  // own; noise_two_rate_kernel then writes the TOD.  Its scratch comes out of the same work buffer: the slow
  // series' spectra take a quarter (half) of the one-rate form's room.
  size_t ld_lo = 0, ld_mw = 0;
  if (two_rate) {
    T = (T_full + rate - 1) / rate + 4;
    ld_lo = ((size_t)T + 3) & ~(size_t)3;
    ld_mw = ((size_t)T_full + 3) & ~(size_t)3;
  }
  int n1, n2;
  if (mrx_noise_period(T, &n1, &n2) != MRX_OK)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "T = %d exceeds the 2^23-sample noise period", T);
  const size_t n = (size_t)n1 * n2;
  const size_t fixed = 32 + 2 * n * (size_t)n_modes + (two_rate ? (size_t)n_modes * ld_mw + 16 : 0);
  const size_t per_pair = 2 * n + 4 + (two_rate ? 2 * ld_lo : 0);
  MRX_REQUIRE(ctx, work_floats >= fixed + per_pair, "work buffer too small: see mrx_noise_work_floats");
  const size_t fit = (work_floats - fixed) / per_pair;
  const int pairs_max = (int)(fit < 16384 ? fit : 16384);
  const int l1 = ilog2(n1), l2 = ilog2(n2);
  const int lj = ilog2(kTileCells) - l1;  // J = 4096 / n1 >= 4

  double* mu = reinterpret_cast<double*>(d_work);                  // [8] modes' pink window means
  double2* mean = reinterpret_cast<double2*>(d_work + 16);         // [pairs_max] pairs' pink window means
  float2* F = reinterpret_cast<float2*>(d_work + 16 + 4 * (size_t)pairs_max);  // [n_modes][n]
  float2* A = F + (size_t)n_modes * n;                             // [pairs][n]
  float* Lo = reinterpret_cast<float*>(A + (size_t)pairs_max * n);  // two-rate: [2 pairs_max][ld_lo]
  float* MW = Lo + 2 * (size_t)pairs_max * ld_lo;                   // two-rate: [n_modes][ld_mw] (16-byte aligned: all sizes are multiples of 4)
  const size_t lds1 = (size_t)(2 * n2 + n2 / 4) * sizeof(float2);
  const size_t lds2 = (size_t)(2 * kTileCells + n1 / 4) * sizeof(float2);
  int threads1 = kBlock;
  const SpectrumKernel pass1 = spectrum_kernel(n2, n_modes, &threads1);
  // every instantiation is sized for the longest first transform once (per context: the
  // attribute belongs to the device)
  MRX_LDS_CAP(ctx, pass1, (2 * 8192 + 8192 / 4) * sizeof(float2));
  MRX_LDS_CAP(ctx, noise_fft_combine, lds2);
  // first transforms of 1024, 2048, 4096 points: the register transform (option bit 1 keeps the Stockham kernel, for A/B runs)
  const size_t lds_r16 = 2 * (size_t)kFft4096Image * sizeof(float2);
  const SpectrumR16Kernel pass1_r16 = !(ctx->options[MRX_OPT_NOISE_GENERIC] & 2) ? spectrum_r16_kernel(n2, n_modes) : nullptr;
  const int pairs_per_wg = kPairsPerBlock * (n2 >= 1024 && n2 <= 4096 ? 4096 / n2 : 1);  // 16 / RB teams
  if (pass1_r16) MRX_LDS_CAP(ctx, pass1_r16, lds_r16);

  SpectrumArgs sp{};
  sp.knee = (float)knee;
  sp.w_ind = 1.0f;
  sp.white_var = two_rate ? 0.0f : (float)(sample_rate / (double)n);  // (two-rate: the white parts are drawn per sample by the writer)
  sp.win.T = T;
  sp.win.k_min = (int)((n + (size_t)T - 1) / (size_t)T);  // ceil(N / T): nothing slower than the TOD
  sp.win.k_cut = (int)(64LL * sp.win.k_min < (long long)(n / 2) ? 64 * sp.win.k_min : n / 2);
  sp.win.mean = mean;
  sp.win.mu = mu;
  if (n_modes > 0) {
    hipLaunchKernelGGL(noise_mode_means, dim3(n_modes), dim3(kBlock), 0, ctx->stream, (int)n, (float)knee,
                       sp.win, key0, key1);
    hipLaunchKernelGGL(noise_mode_table, dim3(mrx_ceil_div((long long)n, kBlock), n_modes), dim3(kBlock),
                       0, ctx->stream, F, n1, n2, sp.white_var, (float)knee, sp.win, key0, key1);
    if (two_rate)
      hipLaunchKernelGGL(noise_mode_white_kernel, dim3(mrx_ceil_div(mrx_ceil_div(T_full, 4), kBlock), n_modes), dim3(kBlock), 0,
                         ctx->stream, MW, ld_mw, T_full, key0, key1);
    MRX_CHECK_LAUNCH(ctx);
    sp.F = F;
    sp.basis = d_basis;
    sp.n_modes = n_modes;
    sp.w_corr = (float)sqrt(corr_prop);
    sp.w_ind = (float)sqrt(1.0 - corr_prop);
  }
  // tiles of the LDS pass 2 that hold a sample t < T: j0 < min(T, n2)
  const int j_used = (long long)T < n2 ? T : n2;
  // Lanes: the batches go round-robin onto up to four streams (the context's and its side
  // streams), each with its own share of the work buffer.  Pass 1 is arithmetic-bound and pass 2
  // HBM-bound, and both end in a tail of half-empty CUs: batches in flight on other lanes fill
  // both (measured at 10 000 x 240 000: 10.0 ms on one lane, 8.3 ms on four).
  const int pairs_total = (D + 1) / 2;
  int lanes = ctx->options[MRX_OPT_NOISE_LANES] > 0 ? ctx->options[MRX_OPT_NOISE_LANES] : 4;
  if (lanes > 1 + mrx_ctx::kSideStreams) lanes = 1 + mrx_ctx::kSideStreams;
  while (lanes > 1 && (pairs_max / lanes < 64 || (long long)(lanes - 1) * (pairs_max / lanes) >= pairs_total)) --lanes;
  const int pairs_lane = pairs_max / lanes;
  if (lanes > 1) {
    const int rc = mrx_side_streams(ctx, lanes - 1);
    if (rc != MRX_OK) return rc;
    MRX_HIP(ctx, hipEventRecord(ctx->side_ev[0], ctx->stream));  // after the tables and whatever precedes the call
    for (int l = 1; l < lanes; ++l) MRX_HIP(ctx, hipStreamWaitEvent(ctx->side_streams[l - 1], ctx->side_ev[0], 0));
  }
  // The lanes' first batches are of different lengths -- (l + 1) / lanes of a batch on lane l -- so that the lanes run a
  // fraction of a cycle apart: started together they stay in step (the kernel trace showed three of four lanes in their
  // first pass at the same time, 520 us each instead of 350-430, then in their second pass together), which is the
  // one arrangement in which an arithmetic-bound pass never runs beside an HBM-bound one.  (Option bit 4 of
  // MRX_OPT_NOISE_GENERIC: equal batches, for A/B runs; the result does not depend on the batching.)
  const bool stagger = lanes > 1 && !(ctx->options[MRX_OPT_NOISE_GENERIC] & 4);
  int d0 = 0;
  for (int batch = 0; d0 < D; ++batch) {
    const int lane = batch % lanes;
    hipStream_t stream = lane ? ctx->side_streams[lane - 1] : ctx->stream;
    int want = 2 * pairs_lane;
    if (stagger && batch < lanes) {
      const int p = (int)((long long)pairs_lane * (lane + 1) / lanes);
      want = 2 * (p > 1 ? p : 1);
    }
    const int count = D - d0 < want ? D - d0 : want;
    const int pairs = (count + 1) / 2;
    CombineArgs h = g;
    h.row0 = d0;
    h.rows = count;
    float* lo_lane = Lo + 2 * (size_t)lane * pairs_lane * ld_lo;
    if (two_rate) {  // pass 2 writes the slow series of the batch's rows, unscaled, into the lane's buffer
      h.out = lo_lane - (size_t)d0 * ld_lo;  // (row d0 + i -> lo_lane[i])
      h.ld = ld_lo;
      h.T = T;
      h.scale = nullptr;
      h.loading = nullptr;
      h.accumulate = 0;
      h.vec_ok = 1;
    }
    sp.row0 = d0;
    sp.rows = count;
    sp.series0 = 16u + (uint32_t)((det_offset + d0) / 2);  // detector pair (2q, 2q+1) is series 16 + q
    sp.win.mean = mean + (size_t)lane * pairs_lane;
    h.mean = sp.win.mean;
    float2* Al = A + (size_t)lane * pairs_lane * n;
    hipLaunchKernelGGL(noise_pair_means, dim3(pairs), dim3(kBlock), 0, stream, n1, n2, sp, key0, key1);
    if (pass1_r16)
      hipLaunchKernelGGL(pass1_r16, dim3(mrx_ceil_div(pairs, pairs_per_wg), n1), dim3(kBlock), lds_r16,
                         stream, Al, n1, sp, pairs, key0, key1);
    else
      hipLaunchKernelGGL(pass1, dim3(mrx_ceil_div(pairs, kPairsPerBlock), n1), dim3(threads1), lds1,
                         stream, Al, n1, n2, l2, sp, pairs, key0, key1);
    if (n1 == 64 && !(ctx->options[MRX_OPT_NOISE_GENERIC] & 1)) {
      const dim3 grid(mrx_ceil_div(j_used, kBlock), pairs);
      if (h.loading || h.accumulate)
        hipLaunchKernelGGL(noise_fft64_combine<true>, grid, dim3(kBlock), 0, stream, Al, n2, h, key0, key1);
      else
        hipLaunchKernelGGL(noise_fft64_combine<false>, grid, dim3(kBlock), 0, stream, Al, n2, h, key0, key1);
    } else if (n1 % 64 == 0 && n1 / 64 <= 16 && !(ctx->options[MRX_OPT_NOISE_GENERIC] & 1)) {
      // n1 = 64 m: 64-point register transforms in place, then the m-point ones with the epilogue
      const int m = n1 / 64;
      const bool extras = h.loading || h.accumulate;
      hipLaunchKernelGGL(noise_fft64_rows, dim3(mrx_ceil_div(j_used, kBlock), m, pairs), dim3(kBlock), 0, stream, Al, n2, m, n1, j_used);
      const dim3 grid_b(mrx_ceil_div(j_used, kBlock), 64, pairs);
#define MRX_COMBINE_ROWS(M)                                                                                   \
  do {                                                                                                       \
    if (extras) hipLaunchKernelGGL((noise_combine_rows<M, true>), grid_b, dim3(kBlock), 0, stream, Al, n2, h); \
    else hipLaunchKernelGGL((noise_combine_rows<M, false>), grid_b, dim3(kBlock), 0, stream, Al, n2, h);       \
  } while (0)
      if (m == 2) MRX_COMBINE_ROWS(2);
      else if (m == 4) MRX_COMBINE_ROWS(4);
      else if (m == 8) MRX_COMBINE_ROWS(8);
      else MRX_COMBINE_ROWS(16);
#undef MRX_COMBINE_ROWS
    } else {
      hipLaunchKernelGGL(noise_fft_combine, dim3(mrx_ceil_div(j_used, 1 << lj), pairs), dim3(kBlock), lds2,
                         stream, Al, n1, n2, l1, lj, h, key0, key1);
    }
    MRX_CHECK_LAUNCH(ctx);
    if (two_rate) {
      mrx_two_rate_args a{};
      a.lo = lo_lane;
      a.ld_lo = ld_lo;
      a.rate = rate;
      a.mode_white = n_modes > 0 ? MW : nullptr;
      a.ld_mw = ld_mw;
      a.n_modes = n_modes;
      a.basis = d_basis;
      a.w_corr = n_modes > 0 ? (float)sqrt(corr_prop) : 0.0f;
      a.sqrt_fs = (float)sqrt(sample_rate);
      a.scale = d_scale;
      a.loading = d_loading;
      a.ld_loading = ld_loading;
      a.per_loading = (float)per_loading;
      a.out = d_out;
      a.ld = ld_out;
      a.row0 = d0;
      a.rows = count;
      a.id0 = (uint32_t)det_offset;
      a.T = T_full;
      a.accumulate = accumulate;
      if (krj) {
        a.bore_el = krj->bore_el; a.dx = krj->dx; a.dy = krj->dy; a.band = krj->band;
        a.cal_axis = krj->cal_axis; a.cal_values = krj->cal_values; a.n_el = krj->n_el; a.n_bands = krj->n_bands;
      }
      a.key0 = key0;
      a.key1 = key1;
      const int rc = mrx_noise_two_rate_write(ctx, stream, a);
      if (rc != MRX_OK) return rc;
    }
    d0 += count;
  }
  for (int l = 1; l < lanes; ++l) {  // join: the context's stream continues after every lane
    MRX_HIP(ctx, hipEventRecord(ctx->side_ev[l], ctx->side_streams[l - 1]));
    MRX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_ev[l], 0));
  }
  return MRX_OK;
}

int mrx_fft_rows(mrx_ctx* ctx, const float* d_in, int rows, int n, int interleave_log2,
                 float* d_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  const int l = ilog2(n);
  if (ctx && interleave_log2 == -3) {  // fft_regs<RB>: n = 1024, 2048 or 4096
    MRX_REQUIRE(ctx, d_in && d_out && rows >= 1 && (n == 1024 || n == 2048 || n == 4096), "bad argument");
    const size_t lds = 2 * (size_t)kFft4096Pitch * 256 * sizeof(float2);  // 16 / RB transforms of 17 * 16 RB values, twice
    const float2* in = reinterpret_cast<const float2*>(d_in);
    float2* out = reinterpret_cast<float2*>(d_out);
    if (n == 1024) {
      MRX_LDS_CAP(ctx, fft_regs_rows_kernel<4>, lds);
      hipLaunchKernelGGL(fft_regs_rows_kernel<4>, dim3(mrx_ceil_div(rows, 4)), dim3(kBlock), lds, ctx->stream, in, out, rows);
    } else if (n == 2048) {
      MRX_LDS_CAP(ctx, fft_regs_rows_kernel<8>, lds);
      hipLaunchKernelGGL(fft_regs_rows_kernel<8>, dim3(mrx_ceil_div(rows, 2)), dim3(kBlock), lds, ctx->stream, in, out, rows);
    } else {
      MRX_LDS_CAP(ctx, fft_regs_rows_kernel<16>, lds);
      hipLaunchKernelGGL(fft_regs_rows_kernel<16>, dim3(rows), dim3(kBlock), lds, ctx->stream, in, out, rows);
    }
    MRX_CHECK_LAUNCH(ctx);
    return MRX_OK;
  }
  if (ctx && interleave_log2 == -2) {  // the workgroup register transform: n must be 4096
    MRX_REQUIRE(ctx, d_in && d_out && rows >= 1 && n == 4096, "bad argument");
    const size_t lds = 2 * (size_t)kFft4096Image * sizeof(float2);
    MRX_LDS_CAP(ctx, fft4096_rows_kernel, lds);
    hipLaunchKernelGGL(fft4096_rows_kernel, dim3(rows), dim3(kBlock), lds, ctx->stream,
                       reinterpret_cast<const float2*>(d_in), reinterpret_cast<float2*>(d_out));
    MRX_CHECK_LAUNCH(ctx);
    return MRX_OK;
  }
  if (ctx && interleave_log2 == -1) {  // the register transform: n must be 64
    MRX_REQUIRE(ctx, d_in && d_out && rows >= 1 && n == 64, "bad argument");
    hipLaunchKernelGGL(fft64_reg_rows_kernel, dim3(mrx_ceil_div(rows, 64)), dim3(64), 0, ctx->stream,
                       reinterpret_cast<const float2*>(d_in), reinterpret_cast<float2*>(d_out), rows);
    MRX_CHECK_LAUNCH(ctx);
    return MRX_OK;
  }
  MRX_REQUIRE(ctx, d_in && d_out && rows >= 1 && l >= 2 && interleave_log2 >= 0, "bad argument");
  const size_t cells = (size_t)n << interleave_log2;
  MRX_REQUIRE(ctx, cells <= 8192, "at most 8192 complex values per row");
  const size_t lds = (2 * cells + n / 4) * sizeof(float2);
  MRX_LDS_CAP(ctx, fft_rows_kernel, lds);
  hipLaunchKernelGGL(fft_rows_kernel, dim3(rows), dim3(kBlock), lds, ctx->stream,
                     reinterpret_cast<const float2*>(d_in), reinterpret_cast<float2*>(d_out), n, l,
                     interleave_log2);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

}  // extern "C"
