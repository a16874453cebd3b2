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
// (generation.py:31-37).  Here pink series are synthesised directly in the
// frequency domain on a power-of-two period N = N1*N2 >= T, and the first T
// samples are kept: X_k = sqrt(knee/|k|) (g1 + i g2) with independent normals
// for every k, so that the variance per frequency bin is the reference's
// knee/|k|.  Because the amplitude is even in k, the inverse transform is a
// circular complex Gaussian series: its real and imaginary parts are two
// independent real series of that spectrum -- one transform serves two
// detectors.  The length-N transform is a four-step FFT on the in-LDS Stockham
// passes the screens use, k = k1 + N1 k2, t = j + N2 m:
//   pass 1  per (series, k1): draw the cells, transform over k2 (length N2),
//           multiply by exp(2 pi i k1 j / N)                   -> A[k1][j]  (HBM)
//   pass 2  per (series, tile of J = 4096/N1 adjacent j): load A[:, j0:j0+J],
//           transform over k1 (length N1, J interleaved sequences), and finish
//           the sample in the epilogue: white draw, mode sum, scale, 16-byte
//           stores of J consecutive samples per (m, detector).
// HBM traffic per detector: 8 N bytes of scratch (A written and read once, shared
// by two detectors) + 4 T of TOD.
// Normals come from Philox-4x32-10 keyed by (seed, global detector index): every
// GPU can regenerate any detector's noise independently, and the modes (keyed by
// the seed alone) are the same on every shard.  Parity with the reference is
// statistical (different generator and period), like the screens'.
#include "mrx_internal.h"

#include "mrx_spectral.h"

namespace {

using namespace mrx_dev;

constexpr uint32_t kTagPink = 0x50494e4bu;   // counter word 3: 'PINK'
constexpr uint32_t kTagWhite = 0x57484954u;  // 'WHIT'
constexpr uint32_t kModeId = 0xffff0000u;    // row id of mode m: kModeId + m (detectors: their global index)
constexpr int kTileCells = 4096;             // complex values per pass-2 workgroup

typedef float vfloat4 __attribute__((ext_vector_type(4)));

// amplitude of spectrum cell k of a length-n series: sqrt(knee/|k|), 0 at k = 0
__device__ __forceinline__ float pink_amp(int k, int n, float sqrt_knee) {
  const int kk = k < n - k ? k : n - k;
  return kk == 0 ? 0.0f : sqrt_knee * __builtin_amdgcn_rsqf((float)kk);
}

// four white normals for samples 4 q .. 4 q + 3 of row `id`
__device__ __forceinline__ vfloat4 white4(uint64_t q, uint32_t id, uint32_t key0, uint32_t key1) {
  const U4 rnd = philox4x32_10(U4{(uint32_t)q, id, (uint32_t)(q >> 32), kTagWhite}, key0, key1);
  const float2 a = box_muller(rnd.x, rnd.y), b = box_muller(rnd.z, rnd.w);
  return vfloat4{a.x, a.y, b.x, b.y};
}

// pass 1: block (k1, series): spectrum cells k = k1 + N1*k2, FFT over k2, twiddle
__global__ __launch_bounds__(kBlock) void noise_spectrum_fft(
    float2* __restrict__ A, int n1, int n2, int log2n2, float knee,
    uint32_t key0, uint32_t key1, uint32_t series0) {
  extern __shared__ float2 lds2[];
  float2* data = lds2;
  float2* tw = lds2 + 2 * n2;
  const int k1 = blockIdx.x;
  const uint32_t series = series0 + blockIdx.y;
  const int n = n1 * n2;  // <= 2^23
  const float sqrt_knee = sqrtf(knee);
  fill_twiddles(tw, n2);
  const int half = n2 >> 1;
  for (int k2 = threadIdx.x; k2 < half; k2 += kBlock) {
    const U4 rnd = philox4x32_10(U4{(uint32_t)k1, (uint32_t)k2, series, kTagPink}, key0, key1);
    const float a0 = pink_amp(k1 + n1 * k2, n, sqrt_knee);
    const float a1 = pink_amp(k1 + n1 * (k2 + half), n, sqrt_knee);
    const float2 g0 = box_muller(rnd.x, rnd.y), g1 = box_muller(rnd.z, rnd.w);
    data[k2] = make_float2(a0 * g0.x, a0 * g0.y);
    data[k2 + half] = make_float2(a1 * g1.x, a1 * g1.y);
  }
  __syncthreads();
  const float2* res = fft_lds_inverse(data, data + n2, tw, n2, log2n2);
  float2* dst = A + ((size_t)blockIdx.y * n1 + k1) * n2;
  const float inv_n = 1.0f / (float)n;
  for (int j = threadIdx.x; j < n2; j += kBlock) {
    // exp(2 pi i k1 j / N); k1*j < N <= 2^23 is exact in float32, and so is the fraction
    // of a revolution the hardware sine and cosine take (absolute error ~1e-6)
    const float rev = (float)(k1 * j) * inv_n;
    dst[j] = cmul(res[j], make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)));
  }
}

struct CombineArgs {
  const float* P;        // [n_modes][ldp] modes (white + pink), or null
  size_t ldp;
  const float* basis;    // [rows][n_modes] or null
  int n_modes;
  const float* scale;    // [rows] or null
  const float* loading;  // [rows][ld_loading] or null
  size_t ld_loading;
  float per_loading;
  float* out;            // [rows][ld]
  size_t ld;
  int row0, rows;        // this launch writes rows row0 .. row0 + rows - 1
  uint32_t id0;          // white-noise id of out row 0
  int T;
  float sqrt_fs, w_corr, w_ind;
  int accumulate, vec_ok;
};

// one group of 4 consecutive samples of one row: everything but the pink value
constexpr int kMaxModes = 8;

// the modes' values at samples t0 .. t0 + 3, loaded once for both rows of a pair
struct Modes4 {
  vfloat4 p[kMaxModes];
};

__device__ __forceinline__ void load_modes(const CombineArgs& g, size_t t0, Modes4& M) {
#pragma unroll
  for (int m = 0; m < kMaxModes; ++m)
    if (m < g.n_modes) M.p[m] = *reinterpret_cast<const vfloat4*>(g.P + (size_t)m * g.ldp + t0);
}

__device__ __forceinline__ void finish4(const CombineArgs& g, int row, size_t t0, vfloat4 pink,
                                        const Modes4& M, uint32_t key0, uint32_t key1) {
  const vfloat4 w = white4(t0 >> 2, g.id0 + (uint32_t)row, key0, key1);
  vfloat4 v = g.sqrt_fs * w + g.w_ind * pink;
  if (g.n_modes > 0) {
    vfloat4 corr = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < kMaxModes; ++m)
      if (m < g.n_modes) corr += g.basis[(size_t)row * g.n_modes + m] * M.p[m];
    v += g.w_corr * corr;
  }
  const float sc = g.scale ? g.scale[row] : 1.0f;
  float* dst = g.out + (size_t)row * g.ld + t0;
  const bool full = g.vec_ok && t0 + 4 <= (size_t)g.T;
  if (g.loading) {
    // total NEP of a sample: NEP + NEP_per_loading x loading (sim/noise.py:35-37)
    const float* L = g.loading + (size_t)row * g.ld_loading + t0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (t0 + q < (size_t)g.T) v[q] *= sc + g.per_loading * L[q];
  } else {
    v *= sc;
  }
  if (full) {
    vfloat4* d4 = reinterpret_cast<vfloat4*>(dst);
    if (g.accumulate) v += *d4;
    __builtin_nontemporal_store(v, d4);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (t0 + q < (size_t)g.T) dst[q] = g.accumulate ? dst[q] + v[q] : v[q];
  }
}

// pass 2: block (tile of J adjacent j, series): FFT over k1 for J sequences at once,
// then the epilogue.  Series s holds rows row0 + 2 s (real part) and row0 + 2 s + 1.
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
  const float2* res = fft_lds_inverse_batched(data, data + kTileCells, tw, n1, log2n1, lj);
  const int row_a = g.row0 + 2 * blockIdx.y;
  for (int e = threadIdx.x * 4; e < n1 * J; e += kBlock * 4) {
    const int m = e >> lj, b = e & (J - 1);
    const size_t t0 = (size_t)n2 * m + j0 + b;
    if (t0 >= (size_t)g.T) continue;
    const float2 r0 = res[e], r1 = res[e + 1], r2 = res[e + 2], r3 = res[e + 3];
    Modes4 M;
    load_modes(g, t0, M);
    finish4(g, row_a, t0, vfloat4{r0.x, r1.x, r2.x, r3.x}, M, key0, key1);
    if (row_a + 1 < g.row0 + g.rows)
      finish4(g, row_a + 1, t0, vfloat4{r0.y, r1.y, r2.y, r3.y}, M, key0, key1);
  }
}

// knee = 0: white noise only (generation.py:25), 4 samples per thread
__global__ __launch_bounds__(kBlock) void noise_white_kernel(CombineArgs g, uint32_t key0, uint32_t key1) {
  const size_t t0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 4;
  if (t0 >= (size_t)g.T) return;
  Modes4 M;  // unused: n_modes == 0 without a pink part
  finish4(g, g.row0 + blockIdx.y, t0, vfloat4{0.f, 0.f, 0.f, 0.f}, M, key0, key1);
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

int ilog2(long long n) {
  int l = 0;
  while ((1LL << l) < n) ++l;
  return (1LL << l) == n ? l : -1;
}

size_t padded4(size_t n) { return (n + 3) & ~(size_t)3; }

}  // namespace

extern "C" {

int mrx_noise_period(int T, int* n1, int* n2) {
  // smallest power of two N = n1*n2 >= max(T, 4096); n2 = min(8192, N/64) is the length of
  // the first transform (one workgroup's LDS), n1 = N/n2 <= 1024 the second's
  if (T <= 0 || !n1 || !n2) return MRX_ERR_INVALID;
  int l = 12;
  while (l < 23 && (1LL << l) < (long long)T) ++l;
  if ((1LL << l) < (long long)T) return MRX_ERR_UNSUPPORTED;
  const int l2 = l - 6 < 13 ? l - 6 : 13;
  *n2 = 1 << l2;
  *n1 = 1 << (l - l2);
  return MRX_OK;
}

int mrx_noise_work_floats(int T, int n_modes, int batch, size_t* floats) {
  int n1, n2;
  int rc = mrx_noise_period(T, &n1, &n2);
  if (rc != MRX_OK || !floats || batch < 1 || n_modes < 0) return rc != MRX_OK ? rc : MRX_ERR_INVALID;
  const size_t n = (size_t)n1 * n2;
  const size_t pairs = (size_t)((batch > n_modes ? batch : n_modes) + 1) / 2;
  *floats = 2 * n * pairs + (((size_t)n_modes * padded4(T) + 15) & ~(size_t)15) + 16;
  return MRX_OK;
}

int mrx_noise_generate(mrx_ctx* ctx, uint64_t seed, int D, int det_offset, int T,
                       double sample_rate, double knee, double corr_prop,
                       const float* d_basis, int n_modes, const float* d_scale,
                       const float* d_loading, size_t ld_loading, double per_loading,
                       float* d_out, size_t ld_out, int accumulate,
                       float* d_work, size_t work_floats) {
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_out, "null pointer");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  MRX_REQUIRE(ctx, det_offset >= 0 && det_offset % 2 == 0 && (long long)det_offset + D < (long long)kModeId,
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
  int n1, n2;
  if (mrx_noise_period(T, &n1, &n2) != MRX_OK)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "T = %d exceeds the 2^23-sample noise period", T);
  const size_t n = (size_t)n1 * n2;
  const size_t ldp = padded4((size_t)T);
  const size_t fixed = (((size_t)n_modes * ldp + 15) & ~(size_t)15) + 16;
  const size_t mode_pairs = (size_t)(n_modes + 1) / 2;
  MRX_REQUIRE(ctx, work_floats >= fixed + 2 * n * (mode_pairs > 1 ? mode_pairs : 1),
              "work buffer too small: see mrx_noise_work_floats");
  const size_t fit = (work_floats - fixed) / (2 * n);
  const int pairs_max = (int)(fit < 16384 ? fit : 16384);
  const int l1 = ilog2(n1), l2 = ilog2(n2);
  const int lj = ilog2(kTileCells) - l1;  // J = 4096 / n1 >= 4
  const float kneef = (float)knee;

  float* P = d_work;  // [n_modes][ldp]
  float2* A = reinterpret_cast<float2*>(d_work + (((size_t)n_modes * ldp + 15) & ~(size_t)15));
  const size_t lds1 = (size_t)(2 * n2 + n2 / 4) * sizeof(float2);
  const size_t lds2 = (size_t)(2 * kTileCells + n1 / 4) * sizeof(float2);
  static size_t cap1 = 0, cap2 = 0;
  if (lds1 > cap1) {
    MRX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(noise_spectrum_fft),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    cap1 = lds1;
  }
  if (lds2 > cap2) {
    MRX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(noise_fft_combine),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    cap2 = lds2;
  }
  // tiles of pass 2 that hold a sample t < T: j0 < min(T, n2)
  const int tiles = mrx_ceil_div((long long)T < n2 ? T : n2, 1 << lj);

  auto synthesise = [&](uint32_t series0, int pairs, const CombineArgs& h) -> int {
    hipLaunchKernelGGL(noise_spectrum_fft, dim3(n1, pairs), dim3(kBlock), lds1, ctx->stream,
                       A, n1, n2, l2, kneef, key0, key1, series0);
    hipLaunchKernelGGL(noise_fft_combine, dim3(tiles, pairs), dim3(kBlock), lds2, ctx->stream,
                       A, n1, n2, l1, lj, h, key0, key1);
    MRX_CHECK_LAUNCH(ctx);
    return MRX_OK;
  };

  // pink series ids: mode pair p is series p; detector pair (global rows 2q, 2q+1) is 16 + q
  if (n_modes > 0) {
    CombineArgs h{};
    h.out = P;
    h.ld = ldp;
    h.rows = n_modes;
    h.id0 = kModeId;
    h.T = T;
    h.sqrt_fs = g.sqrt_fs;
    h.w_ind = 1.0f;
    h.vec_ok = 1;
    int rc = synthesise(0u, (int)mode_pairs, h);
    if (rc != MRX_OK) return rc;
    g.P = P;
    g.ldp = ldp;
    g.basis = d_basis;
    g.n_modes = n_modes;
    g.w_corr = (float)sqrt(corr_prop);
    g.w_ind = (float)sqrt(1.0 - corr_prop);
  } else {
    g.w_ind = 1.0f;
  }
  for (int d0 = 0; d0 < D; d0 += 2 * pairs_max) {
    const int count = D - d0 < 2 * pairs_max ? D - d0 : 2 * pairs_max;
    CombineArgs h = g;
    h.row0 = d0;
    h.rows = count;
    int rc = synthesise(16u + (uint32_t)((det_offset + d0) / 2), (count + 1) / 2, h);
    if (rc != MRX_OK) return rc;
  }
  return MRX_OK;
}

int mrx_fft_rows(mrx_ctx* ctx, const float* d_in, int rows, int n, int interleave_log2,
                 float* d_out) {
  if (!ctx) return MRX_ERR_INVALID;
  const int l = ilog2(n);
  MRX_REQUIRE(ctx, d_in && d_out && rows >= 1 && l >= 2 && interleave_log2 >= 0, "bad argument");
  const size_t cells = (size_t)n << interleave_log2;
  MRX_REQUIRE(ctx, cells <= 8192, "at most 8192 complex values per row");
  const size_t lds = (2 * cells + n / 4) * sizeof(float2);
  static size_t cap = 0;
  if (lds > cap) {
    MRX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fft_rows_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    cap = lds;
  }
  hipLaunchKernelGGL(fft_rows_kernel, dim3(rows), dim3(kBlock), lds, ctx->stream,
                     reinterpret_cast<const float2*>(d_in), reinterpret_cast<float2*>(d_out), n, l,
                     interleave_log2);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

}  // extern "C"
