// Turbulent screen generator for gfx950: Philox-4x32-10 complex normals in
// k space x sqrt(von Karman / Matern PSD) -> 2-D inverse FFT -> real part.
//
// Replaces the reference's autoregressive generator
// (atmosphere/process.py:111-209) by a spectral one with the same target
// covariance, Matern(nu, r0) (functions/__init__.py:30-39): in two dimensions
// its spectrum is PSD(k) ~ (k0^2 + |k|^2)^-(nu+1), k0 = sqrt(2 nu)/r0.
//
// Passes (ny rows, nx columns, both powers of two):
//   1. one workgroup per kx: draw S[ky][kx] for all ky straight into LDS,
//      inverse FFT along y                       -> work1[kx][y]
//   2. LDS-tiled transpose                        -> work2[y][kx]
//   3. one workgroup per y: inverse FFT along x, keep the real part, scale to
//      unit variance                              -> out[y][x]
// The random number of spectrum cell (kx, ky) depends only on (seed, stream,
// kx, ky), so every GPU regenerates bit-identical screens with no broadcast.
#include "mrx_internal.h"

#include "mrx_spectral.h"

namespace {

using namespace mrx_dev;

// sqrt of the von Karman / Matern spectrum: (k0^2 + |k|^2)^expo via exp2/log2
__device__ __forceinline__ float spectrum_amp(double k2, float expo) {
  return __builtin_amdgcn_exp2f(expo * __builtin_amdgcn_logf((float)k2));
}

__device__ __forceinline__ double wavenumber(int i, int n, double d) {
  const int s = i < (n + 1) / 2 ? i : i - n;  // numpy.fft.fftfreq ordering
  return 6.283185307179586476925 * (double)s / ((double)n * d);
}


// pass 1: spectrum column kx, all ky; FFT along y
__global__ __launch_bounds__(kBlock) void screen_spectrum_fft_y(
    float2* __restrict__ work1, int ny, int nx, int log2ny, double dy,
    double dx, double k0sq, float expo, uint32_t key0, uint32_t key1,
    uint32_t stream) {
  extern __shared__ float2 lds2[];
  float2* data = lds2;
  float2* tw = lds2 + 2 * ny;
  const int ix = blockIdx.x;
  const double kx = wavenumber(ix, nx, dx);
  fill_twiddles(tw, ny);
  // one Philox call feeds two cells: words (x, y) -> ky index iy, (z, w) -> iy + ny/2
  const int half = ny >> 1;
  for (int iy = threadIdx.x; iy < half; iy += kBlock) {
    const U4 rnd = philox4x32_10(U4{(uint32_t)ix, (uint32_t)iy, stream, 0u},
                                 key0, key1);
    const double ky0 = wavenumber(iy, ny, dy), ky1 = wavenumber(iy + half, ny, dy);
    const float amp0 = spectrum_amp(k0sq + kx * kx + ky0 * ky0, expo);
    const float amp1 = spectrum_amp(k0sq + kx * kx + ky1 * ky1, expo);
    const float2 g0 = box_muller(rnd.x, rnd.y), g1 = box_muller(rnd.z, rnd.w);
    data[iy] = make_float2(amp0 * g0.x, amp0 * g0.y);
    data[iy + half] = make_float2(amp1 * g1.x, amp1 * g1.y);
  }
  __syncthreads();
  const float2* res = fft_lds_inverse(data, data + ny, tw, ny, log2ny);
  float2* dst = work1 + (size_t)ix * ny;
  for (int y = threadIdx.x; y < ny; y += kBlock) dst[y] = res[y];
}

// pass 2: transpose [rows][cols] -> [cols][rows], 32x32 tiles, padded LDS
__global__ __launch_bounds__(kBlock) void transpose_c32(
    const float2* __restrict__ in, float2* __restrict__ out, int rows,
    int cols) {
  __shared__ float2 tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    if (r < rows && c < cols) tile[k][tx] = in[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][k];
  }
}

// pass 3: FFT along x of row y; real part * norm
__global__ __launch_bounds__(kBlock) void screen_fft_x_real(
    const float2* __restrict__ work2, float* __restrict__ out, int ny, int nx,
    int log2nx, const double* __restrict__ psd_sum) {
  extern __shared__ float2 lds2[];
  float2* data = lds2;
  float2* tw = lds2 + 2 * nx;
  const int y = blockIdx.x;
  fill_twiddles(tw, nx);
  const float2* src = work2 + (size_t)y * nx;
  for (int x = threadIdx.x; x < nx; x += kBlock) data[x] = src[x];
  __syncthreads();
  const float2* res = fft_lds_inverse(data, data + nx, tw, nx, log2nx);
  const float norm = (float)(1.0 / sqrt(*psd_sum));
  float* dst = out + (size_t)y * nx;
  for (int x = threadIdx.x; x < nx; x += kBlock) dst[x] = res[x].x * norm;
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

__global__ void philox_normal_kernel(float* __restrict__ out, size_t n,
                                     uint32_t key0, uint32_t key1,
                                     uint32_t stream) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const U4 rnd = philox4x32_10(
      U4{(uint32_t)i, (uint32_t)(i >> 32), stream, 0u}, key0, key1);
  out[i] = box_muller(rnd.x, rnd.y).x;
}

int ilog2_exact(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return (1 << l) == n ? l : -1;
}

// The normalisation depends on the grid and spectrum only, not on the draw: it
// is reduced once per (ny, nx, dy, dx, r0, nu) and kept in a device slot.
int psd_sum_slot(mrx_ctx* ctx, int ny, int nx, double dy, double dx, double r0,
                 double nu, const double** d_sum) {
  if (!ctx->d_reduce) {
    MRX_HIP(ctx, hipMalloc(&ctx->d_reduce, mrx_ctx::kPsdSlots * sizeof(double)));
    ctx->reduce_cap = mrx_ctx::kPsdSlots;
  }
  for (int i = 0; i < mrx_ctx::kPsdSlots; ++i) {
    const auto& k = ctx->psd[i];
    if (k.valid && k.ny == ny && k.nx == nx && k.dy == dy && k.dx == dx &&
        k.r0 == r0 && k.nu == nu) {
      *d_sum = ctx->d_reduce + i;
      return MRX_OK;
    }
  }
  const int slot = ctx->psd_next;
  ctx->psd_next = (ctx->psd_next + 1) % mrx_ctx::kPsdSlots;
  double* dst = ctx->d_reduce + slot;
  const double k0sq = 2.0 * nu / (r0 * r0);
  const float expo = (float)(-(nu + 1.0) / 2.0);
  MRX_HIP(ctx, hipMemsetAsync(dst, 0, sizeof(double), ctx->stream));
  const size_t n = (size_t)ny * nx;
  const int blocks = (int)((n + kBlock - 1) / kBlock < 2048
                               ? (n + kBlock - 1) / kBlock
                               : 2048);
  hipLaunchKernelGGL(psd_sum_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream,
                     dst, ny, nx, dy, dx, k0sq, expo);
  MRX_CHECK_LAUNCH(ctx);
  ctx->psd[slot] = {true, ny, nx, dy, dx, r0, nu};
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

int mrx_screen_generate(mrx_ctx* ctx, uint64_t seed, uint32_t stream, int ny,
                        int nx, double dy, double dx, double r0, double nu,
                        float* d_out, float* d_work) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, d_out && d_work, "null pointer");
  MRX_REQUIRE(ctx, dy > 0 && dx > 0 && r0 > 0 && nu > 0,
              "steps, r0 and nu must be positive");
  const int ly = ilog2_exact(ny), lx = ilog2_exact(nx);
  if (ly < 0 || lx < 0 || ny < 64 || nx < 64 || ny > 8192 || nx > 8192)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED,
                    "screen sides must be powers of two in [64, 8192] "
                    "(got %d x %d)", ny, nx);
  const double* d_sum = nullptr;
  int rc = psd_sum_slot(ctx, ny, nx, dy, dx, r0, nu, &d_sum);
  if (rc != MRX_OK) return rc;

  const double k0sq = 2.0 * nu / (r0 * r0);
  const float expo = (float)(-(nu + 1.0) / 2.0);
  float2* work1 = reinterpret_cast<float2*>(d_work);
  float2* work2 = work1 + (size_t)ny * nx;
  const uint32_t key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);

  // two ping-pong images + n/4 twiddles
  const size_t lds_y = (size_t)(2 * ny + ny / 4) * sizeof(float2);
  const size_t lds_x = (size_t)(2 * nx + nx / 4) * sizeof(float2);
  static size_t lds_set_y = 0, lds_set_x = 0;  // raise the dynamic-LDS cap once
  if (lds_y > lds_set_y) {
    MRX_HIP(ctx, hipFuncSetAttribute(
                     reinterpret_cast<const void*>(screen_spectrum_fft_y),
                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_y));
    lds_set_y = lds_y;
  }
  if (lds_x > lds_set_x) {
    MRX_HIP(ctx, hipFuncSetAttribute(
                     reinterpret_cast<const void*>(screen_fft_x_real),
                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_x));
    lds_set_x = lds_x;
  }

  hipLaunchKernelGGL(screen_spectrum_fft_y, dim3(nx), dim3(kBlock), lds_y,
                     ctx->stream, work1, ny, nx, ly, dy, dx, k0sq, expo, key0,
                     key1, stream);
  MRX_CHECK_LAUNCH(ctx);
  // work1 is [nx rows][ny cols] -> work2 [ny][nx]
  hipLaunchKernelGGL(transpose_c32, dim3(ny / 32, nx / 32), dim3(kBlock), 0,
                     ctx->stream, work1, work2, nx, ny);
  MRX_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(screen_fft_x_real, dim3(ny), dim3(kBlock), lds_x,
                     ctx->stream, work2, d_out, ny, nx, lx, d_sum);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
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
