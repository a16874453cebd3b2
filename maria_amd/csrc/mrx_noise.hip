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
// (generation.py:31-37).  Here a pink series is synthesised directly in the
// frequency domain -- X_k = sqrt(knee/|k|) (g1 + i g2), real part of the
// inverse transform, so that the variance per frequency bin is the reference's
// knee/|k| -- on a power-of-two period N = N1*N2 >= T, and the first T samples
// are kept.  The length-N transform is a four-step FFT built from the same
// in-LDS Stockham passes as the screens:
//   1. per k1: draw X[k1 + N1 k2], transform over k2 (length N2), apply the
//      twiddle exp(2 pi i k1 n2 / N)                       -> A[k1][n2]
//   2. transpose                                             -> A^T[n2][k1]
//   3. per n2: transform over k1 (length N1), keep Re        -> R[n2][n1]
//   4. time order t = N2 n1 + n2 is R transposed: done while combining.
// Normals come from Philox-4x32-10 keyed by (seed, series): every GPU can
// regenerate any detector's noise independently.  Parity with the reference is
// statistical (different generator and period), like the screens'.
#include "mrx_internal.h"

#include "mrx_spectral.h"

namespace {

using namespace mrx_dev;

constexpr uint32_t kTagPink = 0x50494e4bu;   // counter word 3: 'PINK'
constexpr uint32_t kTagWhite = 0x57484954u;  // 'WHIT'
constexpr uint32_t kModeWhite = 0xffff0000u;  // white part of mode m: detector word kModeWhite + m

// amplitude of spectrum cell k of a length-n series: sqrt(knee/|k|), 0 at k = 0
__device__ __forceinline__ float pink_amp(long long k, long long n, float knee) {
  const long long kk = k < n - k ? k : n - k;
  return kk == 0 ? 0.0f : sqrtf(knee / (float)kk);
}

// pass 1: block (k1, series): spectrum cells k = k1 + N1*k2, FFT over k2, twiddle
__global__ __launch_bounds__(kBlock) void noise_spectrum_fft(
    float2* __restrict__ work1, int n1, int n2, int log2n2, float knee,
    uint32_t key0, uint32_t key1, uint32_t series0) {
  extern __shared__ float2 lds2[];
  float2* data = lds2;
  float2* tw = lds2 + 2 * n2;
  const int k1 = blockIdx.x;
  const uint32_t series = series0 + blockIdx.y;
  const long long n = (long long)n1 * n2;
  fill_twiddles(tw, n2);
  const int half = n2 >> 1;
  for (int k2 = threadIdx.x; k2 < half; k2 += kBlock) {
    const U4 rnd = philox4x32_10(U4{(uint32_t)k1, (uint32_t)k2, series, kTagPink}, key0, key1);
    const float a0 = pink_amp(k1 + (long long)n1 * k2, n, knee);
    const float a1 = pink_amp(k1 + (long long)n1 * (k2 + half), n, knee);
    const float2 g0 = box_muller(rnd.x, rnd.y), g1 = box_muller(rnd.z, rnd.w);
    data[k2] = make_float2(a0 * g0.x, a0 * g0.y);
    data[k2 + half] = make_float2(a1 * g1.x, a1 * g1.y);
  }
  __syncthreads();
  const float2* res = fft_lds_inverse(data, data + n2, tw, n2, log2n2);
  float2* dst = work1 + ((size_t)blockIdx.y * n1 + k1) * n2;
  const float inv_n = 1.0f / (float)n;
  for (int j = threadIdx.x; j < n2; j += kBlock) {
    // exp(2 pi i k1 j / N); k1*j < N, reduced exactly in integers
    float s, c;
    sincospif(2.0f * (float)((long long)k1 * j) * inv_n, &s, &c);
    dst[j] = cmul(res[j], make_float2(c, s));
  }
}

// pass 2: batched transpose [rows][cols] -> [cols][rows] of complex values
__global__ __launch_bounds__(kBlock) void noise_transpose_c(
    const float2* __restrict__ in, float2* __restrict__ out, int rows, int cols) {
  __shared__ float2 tile[32][33];
  const size_t base = (size_t)blockIdx.z * rows * cols;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int k = ty; k < 32; k += 8) tile[k][tx] = in[base + (size_t)(r0 + k) * cols + c0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8) out[base + (size_t)(c0 + k) * rows + r0 + tx] = tile[tx][k];
}

// pass 3: block (n2, series): FFT over k1, real part -> R[series][n2][n1]
__global__ __launch_bounds__(kBlock) void noise_fft_real(
    const float2* __restrict__ work2, float* __restrict__ R, int n1, int n2,
    int log2n1) {
  extern __shared__ float2 lds2[];
  float2* data = lds2;
  float2* tw = lds2 + 2 * n1;
  fill_twiddles(tw, n1);
  const size_t row = (size_t)blockIdx.y * n2 + blockIdx.x;
  const float2* src = work2 + row * n1;
  for (int j = threadIdx.x; j < n1; j += kBlock) data[j] = src[j];
  __syncthreads();
  const float2* res = fft_lds_inverse(data, data + n1, tw, n1, log2n1);
  float* dst = R + row * n1;
  for (int j = threadIdx.x; j < n1; j += kBlock) dst[j] = res[j].x;
}

// modes: time order P[m][t] = sqrt(fs) w_m[t] + R_m[n2][n1] with t = n2_len*n1 + n2 (first T
// samples): a mode is itself white + pink (generation.py:41-43 recurses with the same knee)
__global__ __launch_bounds__(kBlock) void noise_modes_to_time(
    const float* __restrict__ R, float* __restrict__ P, int n1, int n2, int T,
    float sqrt_fs, uint32_t key0, uint32_t key1) {
  __shared__ float tile[32][33];
  const size_t base = (size_t)blockIdx.z * n1 * n2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32;  // a: n1, b: n2
  for (int k = ty; k < 32; k += 8) tile[k][tx] = R[base + (size_t)(b0 + k) * n1 + a0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const size_t t = (size_t)n2 * (a0 + k) + b0 + tx;
    if (t >= (size_t)T) continue;
    const U4 rnd = philox4x32_10(U4{(uint32_t)(t >> 1), kModeWhite + blockIdx.z, (uint32_t)(t >> 33), kTagWhite}, key0, key1);
    const float2 g = box_muller(rnd.x, rnd.y);
    P[(size_t)blockIdx.z * T + t] = tile[tx][k] + sqrt_fs * ((t & 1) ? g.y : g.x);
  }
}

// combine: out[d][t] = scale_d (sqrt(fs) w + sqrt(1-c) p_d[t] + sqrt(c) sum_m B[d,m] P_m[t])
__global__ __launch_bounds__(kBlock) void noise_combine(
    const float* __restrict__ R, const float* __restrict__ P,
    const float* __restrict__ basis, int n_modes, const float* __restrict__ scale,
    float* __restrict__ out, size_t ld, int n1, int n2, int T, int d_first,
    int D, float sqrt_fs, float w_corr, float w_ind, uint32_t key0, uint32_t key1,
    int accumulate, const float* __restrict__ loading, size_t ld_loading, float per_loading) {
  __shared__ float tile[32][33];
  const int dl = blockIdx.z;          // detector within the batch
  const int d = d_first + dl;
  if (d >= D) return;
  const size_t base = (size_t)dl * n1 * n2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  if ((size_t)n2 * a0 + b0 >= (size_t)T) return;  // whole tile past the end
  for (int k = ty; k < 32; k += 8)
    tile[k][tx] = R ? R[base + (size_t)(b0 + k) * n1 + a0 + tx] : 0.0f;  // no pink part: zeros
  __syncthreads();
  const float sc = scale ? scale[d] : 1.0f;
  float bm[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) bm[m] = (basis && m < n_modes) ? basis[(size_t)d * n_modes + m] : 0.0f;
  for (int k = ty; k < 32; k += 8) {
    const size_t t = (size_t)n2 * (a0 + k) + b0 + tx;
    if (t >= (size_t)T) continue;
    float corr = 0.0f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
      if (m < n_modes) corr += bm[m] * P[(size_t)m * T + t];
    // white: one Philox call per pair of samples, the pair's two Box-Muller outputs
    const U4 rnd = philox4x32_10(U4{(uint32_t)(t >> 1), (uint32_t)d, (uint32_t)(t >> 33), kTagWhite}, key0, key1);
    const float2 g = box_muller(rnd.x, rnd.y);
    const float white = (t & 1) ? g.y : g.x;
    // total NEP of this sample: NEP + NEP_per_loading x loading (sim/noise.py:35-37)
    const float amp = loading ? sc + per_loading * loading[(size_t)d * ld_loading + t] : sc;
    const float v = amp * (sqrt_fs * white + w_ind * tile[tx][k] + w_corr * corr);
    float* dst = out + (size_t)d * ld + t;
    *dst = accumulate ? *dst + v : v;
  }
}

int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return (1 << l) == n ? l : -1;
}

}  // namespace

extern "C" {

int mrx_noise_period(int T, int* n1, int* n2) {
  // smallest power of two N = n1*n2 >= T with 64 <= n2 <= n1 <= 8192, n1 in {n2, 2 n2}
  if (T <= 0 || !n1 || !n2) return MRX_ERR_INVALID;
  int l = 12;  // N >= 4096
  while (l < 26 && (1LL << l) < (long long)T) ++l;
  if ((1LL << l) < (long long)T) return MRX_ERR_UNSUPPORTED;
  *n2 = 1 << (l / 2);
  *n1 = 1 << (l - l / 2);
  return MRX_OK;
}

int mrx_noise_work_floats(int T, int n_modes, int batch, size_t* floats) {
  int n1, n2;
  int rc = mrx_noise_period(T, &n1, &n2);
  if (rc != MRX_OK || !floats || batch < 1 || n_modes < 0) return rc != MRX_OK ? rc : MRX_ERR_INVALID;
  const size_t n = (size_t)n1 * n2;
  const size_t series = (size_t)(batch > n_modes ? batch : n_modes);
  *floats = 5 * n * series + (size_t)n_modes * T + 16;
  return MRX_OK;
}

int mrx_noise_generate(mrx_ctx* ctx, uint64_t seed, int D, int T,
                       double sample_rate, double knee, double corr_prop,
                       const float* d_basis, int n_modes, const float* d_scale,
                       const float* d_loading, size_t ld_loading, double per_loading,
                       float* d_out, size_t ld_out, int accumulate,
                       float* d_work, size_t work_floats) {
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_out && d_work, "null pointer");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  MRX_REQUIRE(ctx, !d_loading || ld_loading >= (size_t)T, "ld_loading smaller than T");
  MRX_REQUIRE(ctx, !d_loading || !accumulate || d_loading != d_out,
              "accumulating into the loading the noise level is read from");
  MRX_REQUIRE(ctx, sample_rate > 0 && knee >= 0 && corr_prop >= 0 && corr_prop <= 1,
              "need sample_rate > 0, knee >= 0, 0 <= corr_prop <= 1");
  MRX_REQUIRE(ctx, n_modes >= 0 && n_modes <= 8 && (n_modes == 0 || d_basis),
              "0 <= n_modes <= 8 and a basis when n_modes > 0");
  int n1, n2;
  if (mrx_noise_period(T, &n1, &n2) != MRX_OK)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "T = %d exceeds the 2^26-sample noise period", T);
  const size_t n = (size_t)n1 * n2;
  const size_t fixed = (size_t)n_modes * T + 16;
  MRX_REQUIRE(ctx, work_floats >= fixed + 5 * n * (size_t)(n_modes > 1 ? n_modes : 1),
              "work buffer too small: see mrx_noise_work_floats");
  const size_t fit = (work_floats - fixed) / (5 * n);
  const int batch_max = (int)(fit < 32768 ? fit : 32768);  // grid.y / grid.z limits
  const int l1 = ilog2(n1), l2 = ilog2(n2);
  const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
  const float kneef = (float)knee;

  float* P = d_work;  // [n_modes][T]
  float2* work1 = reinterpret_cast<float2*>(d_work + (((size_t)n_modes * T + 15) & ~(size_t)15));
  const size_t lds1 = (size_t)(2 * n2 + n2 / 4) * sizeof(float2);
  const size_t lds3 = (size_t)(2 * n1 + n1 / 4) * sizeof(float2);
  static size_t cap1 = 0, cap3 = 0;
  if (lds1 > cap1) {
    MRX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(noise_spectrum_fft),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    cap1 = lds1;
  }
  if (lds3 > cap3) {
    MRX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(noise_fft_real),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
    cap3 = lds3;
  }

  // series ids: modes are 0..n_modes-1 of stream "modes"; detector d is series 16 + d
  auto synthesise = [&](uint32_t series0, int count, float2* w1) -> int {
    float2* w2 = w1 + (size_t)count * n;
    hipLaunchKernelGGL(noise_spectrum_fft, dim3(n1, count), dim3(kBlock), lds1, ctx->stream,
                       w1, n1, n2, l2, kneef, key0, key1, series0);
    hipLaunchKernelGGL(noise_transpose_c, dim3(n2 / 32, n1 / 32, count), dim3(kBlock), 0,
                       ctx->stream, w1, w2, n1, n2);
    // R overwrites the front of w1 (no longer needed once transposed)
    hipLaunchKernelGGL(noise_fft_real, dim3(n2, count), dim3(kBlock), lds3, ctx->stream, w2,
                       reinterpret_cast<float*>(w1), n1, n2, l1);
    MRX_CHECK_LAUNCH(ctx);
    return MRX_OK;
  };

  const bool pink = knee > 0.0;
  if (pink && n_modes > 0) {
    int rc = synthesise(0u, n_modes, work1);
    if (rc != MRX_OK) return rc;
    hipLaunchKernelGGL(noise_modes_to_time, dim3(n1 / 32, n2 / 32, n_modes), dim3(kBlock), 0,
                       ctx->stream, reinterpret_cast<float*>(work1), P, n1, n2, T,
                       (float)sqrt(sample_rate), key0, key1);
    MRX_CHECK_LAUNCH(ctx);
  }
  const float w_corr = (pink && n_modes > 0) ? (float)sqrt(corr_prop) : 0.0f;
  const float w_ind = pink ? (float)sqrt(n_modes > 0 ? 1.0 - corr_prop : 1.0) : 0.0f;
  for (int d0 = 0; d0 < D; d0 += batch_max) {
    const int count = D - d0 < batch_max ? D - d0 : batch_max;
    if (pink) {
      int rc = synthesise(16u + (uint32_t)d0, count, work1);
      if (rc != MRX_OK) return rc;
    }
    // tiles needed to cover t < T: n1 index up to ceil(T / n2)
    const int a_tiles = mrx_ceil_div(mrx_ceil_div(T, n2), 32);
    hipLaunchKernelGGL(noise_combine, dim3(a_tiles < n1 / 32 ? a_tiles : n1 / 32, n2 / 32, count),
                       dim3(kBlock), 0, ctx->stream,
                       pink ? reinterpret_cast<float*>(work1) : (float*)nullptr, P, d_basis,
                       (pink ? n_modes : 0), d_scale, d_out, ld_out, n1, n2, T, d0, D,
                       (float)sqrt(sample_rate), w_corr, w_ind, key0, key1, accumulate,
                       d_loading, ld_loading, (float)per_loading);
    MRX_CHECK_LAUNCH(ctx);
  }
  return MRX_OK;
}

}  // extern "C"
