// Separable Gaussian stencil with scipy.ndimage.gaussian_filter semantics,
// LDS-tiled for gfx950.
//
// Reference call sites: atmosphere/atmosphere.py:341-344 (screen smoothing to
// the beam) and map/projection.py:485-504 (ProjectionMap.smooth).  scipy runs
// gaussian_filter1d along axis 0 then axis 1, each a correlation with
// exp(-x^2/2 sigma^2) normalised over |x| <= int(truncate*sigma + 0.5), on a
// "reflect" (edge-repeating) extension, accumulating in float64 and storing
// the pass result in the input dtype (float32 here).
//
// Both passes stage the tile plus its halo in LDS once and read every tap from
// there; lanes run along x in both passes so LDS reads are conflict-free and
// global traffic is coalesced.
#include <cmath>
#include <vector>

#include "mrx_internal.h"

namespace {

constexpr int kBlock = 256;

// scipy "reflect": (d c b a | a b c d | d c b a)
__device__ __forceinline__ int reflect_index(int i, int n) {
  if (i >= 0 && i < n) return i;
  const int p = 2 * n;
  int m = i % p;
  if (m < 0) m += p;
  return m < n ? m : p - 1 - m;
}

// The same for an index that is the same in every lane of the wave (a row of the y pass): the fold -- two integer
// divisions, forty instructions -- only where the row does lie outside the array, decided on the scalar unit.  (Per
// element it was most of the y pass at large radii: 78 rows staged per thread for 8 outputs.)
__device__ __forceinline__ int reflect_row(int i, int n) {
  const int u = __builtin_amdgcn_readfirstlane(i);
  return (u >= 0 && u < n) ? u : reflect_index(u, n);
}

// Both passes stage tile + halo in LDS as float32 and widen a value when it is read; a thread makes four outputs from ONE
// sliding window, so the taps meet the values in window order -- NOT scipy's symmetric order (centre tap, then
// (left + right) w_k): the float64 sums differ from scipy's by their association, i.e. by float32 roundings of the result
// (the parity test holds the exact mode to 2.5e-7, a last-bit flip in a few per cent of the pixels).
//
// Two accumulation modes (MRX_OPT_GAUSS_ACCUM):
//  * exact: every product and sum in float64, scipy's arithmetic -- bound by the float64 multiply-adds (2 radius + 1 per
//    pixel and pass at half the float32 rate, plus a conversion a value);
//  * blocked (round 6): float32 products summed in float32 over 16 window steps, the 16-step sums added up in float64.
//    A block's sum carries at most 16 float32 roundings of terms that are 1/16 .. 1/257 of the result: measured 1.5e-7
//    against scipy (float64) at 257 taps, bound 1e-6 in the test -- ten times inside the north star's 1e-5 -- at twice the
//    speed.  The default from radius 16 (sigma >= 4 pixels); smaller stencils are not bound by their arithmetic.

// pass along x (contiguous axis): one workgroup = 1024 consecutive outputs of a row, a thread 4 consecutive ones from
// ONE sliding window of 2 radius + 4 values: the window comes out of LDS 16 bytes a read (conflict-free: a lane's four
// values sit in four banks of their own), every value meets four taps -- the window step's and the three before it --
// and four float64 accumulators, one per output.  One output per thread read its 2 radius + 1 values one by one, as
// float64: 16 bytes of LDS traffic per tap and output, which bound the pass (round 1); this form reads 1 byte.
// wz holds the tap row with 3 zeros on either side, as in the y pass.
constexpr int kXPerThread = 4;
constexpr int kXCols = kBlock * kXPerThread;

constexpr int kAccBlock = 16;  // window steps per float32 partial sum (blocked mode)

// The taps as the kernels read them, in global memory: [4 zeros][ntap taps][zeros up to a whole block past the window] as
// float64, then the same as float32 (get_taps).  The index of a tap is the same for every lane, so the compiler fetches
// them with SCALAR loads (16 bytes at a time in the blocked mode: the 4 zeros in front keep the taps of window steps
// k .. k + 3, k a multiple of 4, on a 16-byte boundary) and a tap is a scalar operand of its multiply-add.  Rounds 1-5
// staged them in LDS and read them back per step: every lane the same address, but a read all the same -- half of the
// LDS traffic of the x pass, and what bound both passes (round 6: float32 sums alone gained 5 %).
__host__ __device__ inline int gauss_tap_slots(int ntap) { return (4 + ntap + 7 + kAccBlock + 3) & ~3; }

// kXRows rows of the plane per workgroup, staged together (one barrier; a thread's loads of all of them in flight at once):
// with one row a workgroup the pass was a chain of load - barrier - 300 multiply-adds - store per 1024 outputs, 65 536
// workgroups of it at 8192^2, and waited more than it computed.
constexpr int kXRows = 4;

template <bool kBlocked>
__global__ __launch_bounds__(kBlock) void gauss_x_kernel(
    const float* __restrict__ in, float* __restrict__ out, int ny, int nx,
    const double* __restrict__ tapsd, const float* __restrict__ tapsf, int radius) {
  extern __shared__ __align__(16) float img_all[];  // [kXRows][pitch], pitch = span + 4 + kAccBlock rounded up to 4
  const int ntap = 2 * radius + 1;
  const int span = kXCols + 2 * radius;
  const int pitch = (span + 4 + kAccBlock + 3) & ~3;
  const int y_first = blockIdx.y * kXRows;
  const int x0 = blockIdx.x * kXCols;
  // (a tile whose window lies inside the row takes its values as they stand: the fold of reflect_index -- two integer
  // divisions per element -- and its branches kept the loads from being issued together)
  const bool interior = x0 - radius >= 0 && x0 - radius + span <= nx;
#pragma unroll
  for (int r = 0; r < kXRows; ++r) {
    const int y = min(y_first + r, ny - 1);
    const float* src = in + (size_t)y * nx;
    float* img = img_all + r * pitch;
    if (interior) {
      const float* s0 = src + (x0 - radius);
#pragma unroll 4
      for (int i = threadIdx.x; i < pitch; i += kBlock) img[i] = i < span ? s0[i] : 0.0f;
    } else {
      for (int i = threadIdx.x; i < pitch; i += kBlock)
        img[i] = i < span ? src[reflect_index(x0 - radius + i, nx)] : 0.0f;
    }
  }
  __syncthreads();
  const int x = x0 + threadIdx.x * kXPerThread;
  if (x >= nx) return;
  const int steps = ntap + 3;  // (the steps past the window meet zero taps and the image's zero padding)
  for (int r = 0; r < kXRows && y_first + r < ny; ++r) {
  const int y = y_first + r;
  // outputs x .. x + 3 read img[4 tid .. 4 tid + 2 radius + 3]; window step k: value img[4 tid + k], tap of output j: step k - j
  const float4* c4 = reinterpret_cast<const float4*>(img_all + r * pitch + threadIdx.x * kXPerThread);
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  if constexpr (kBlocked) {
    float t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
    for (int k = 0; k < steps; k += kAccBlock) {
      float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll
      for (int kk = 0; kk < kAccBlock; kk += 4) {
        const float4 q = c4[(k + kk) >> 2];
        const float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float t0 = tapsf[4 + k + kk + i];  // (uniform: a scalar load)
          p0 = fmaf(t0, v[i], p0);
          p1 = fmaf(t1, v[i], p1);
          p2 = fmaf(t2, v[i], p2);
          p3 = fmaf(t3, v[i], p3);
          t3 = t2;
          t2 = t1;
          t1 = t0;
        }
      }
      acc0 += (double)p0;
      acc1 += (double)p1;
      acc2 += (double)p2;
      acc3 += (double)p3;
    }
  } else {
    double t1 = 0.0, t2 = 0.0, t3 = 0.0;  // the taps of the three steps before
    for (int k = 0; k < steps; k += 4) {
      const float4 q = c4[k >> 2];
      const float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double t0 = tapsd[4 + k + i];
        const double vv = (double)v[i];
        acc0 += t0 * vv;
        acc1 += t1 * vv;
        acc2 += t2 * vv;
        acc3 += t3 * vv;
        t3 = t2;
        t2 = t1;
        t1 = t0;
      }
    }
  }
  float* dst = out + (size_t)y * nx + x;
  if (x + 3 < nx && (nx & 3) == 0) {
    *reinterpret_cast<float4*>(dst) = make_float4((float)acc0, (float)acc1, (float)acc2, (float)acc3);
  } else {
    const double acc[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (x + j < nx) dst[j] = (float)acc[j];
  }
  }
}

// pass along y: tile of tile_rows outputs x 64 columns; 4 row groups of threads.  A thread produces kRows consecutive
// output rows from ONE sliding window of 2 radius + kRows LDS values: one 4-byte LDS read per step and kRows multiply-adds
// on scalar taps.  kRows = 8 (round 6; 4 before): with four outputs a step the pass was bound by those reads -- a wave's
// read takes the LDS two cycles, its four multiply-adds the SIMD eight, and four SIMDs share the LDS.  kRows = 4 stays for
// the tiles of 16 rows that the largest radii need.
constexpr int kYCols = 64;

template <bool kBlocked, int kRows>
__global__ __launch_bounds__(kBlock) void gauss_y_kernel(
    const float* __restrict__ in, float* __restrict__ out, int ny, int nx,
    const double* __restrict__ tapsd, const float* __restrict__ tapsf, int radius, int tile_rows) {
  // float32 image; values widen to float64 as they are read (exact mode)
  extern __shared__ __align__(16) float img[];
  const int lane = threadIdx.x % kYCols;
  const int grp = threadIdx.x / kYCols;     // 0..3
  const int x = blockIdx.x * kYCols + lane;
  const int y0 = blockIdx.y * tile_rows;
  const int span = tile_rows + 2 * radius;
  const int ntap = 2 * radius + 1;
  const int xs = min(x, nx - 1);
  // (the window is walked in whole blocks: kAccBlock + kRows rows of zeros behind the image for the steps past it)
  if (y0 - radius >= 0 && y0 - radius + span <= ny) {
    // an interior tile: its rows as they stand, eight loads of a thread in flight together (with the fold's branch in the
    // loop every row waited for its own load: 38 memory latencies a thread at radius 32, most of the pass)
    const float* s0 = in + (size_t)(y0 - radius) * nx + xs;
    int i = grp;
    for (; i + 7 * (kBlock / kYCols) < span; i += 8 * (kBlock / kYCols)) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s0[(size_t)(i + u * (kBlock / kYCols)) * nx];
#pragma unroll
      for (int u = 0; u < 8; ++u) img[(i + u * (kBlock / kYCols)) * kYCols + lane] = v[u];
    }
    for (; i < span + kAccBlock + kRows; i += kBlock / kYCols) img[i * kYCols + lane] = i < span ? s0[(size_t)i * nx] : 0.0f;
  } else {
    for (int i = grp; i < span + kAccBlock + kRows; i += kBlock / kYCols)
      img[i * kYCols + lane] = i < span ? in[(size_t)reflect_row(y0 - radius + i, ny) * nx + xs] : 0.0f;
  }
  __syncthreads();
  if (x >= nx) return;
  const int rows_per_grp = tile_rows / (kBlock / kYCols);
  const int steps = ntap + kRows - 1;
  for (int r0 = grp * rows_per_grp; r0 < (grp + 1) * rows_per_grp; r0 += kRows) {
    if (y0 + r0 >= ny) break;
    // outputs r0 .. r0 + kRows - 1 read LDS rows r0 .. r0 + 2 radius + kRows - 1; output j meets the tap of step k - j
    const float* c = img + r0 * kYCols + lane;
    double acc[kRows];
#pragma unroll
    for (int j = 0; j < kRows; ++j) acc[j] = 0.0;
    if constexpr (kBlocked) {
      float t[kRows];  // t[j]: the tap of step k - j
#pragma unroll
      for (int j = 0; j < kRows; ++j) t[j] = 0.0f;
      for (int k = 0; k < steps; k += kAccBlock) {
        float p[kRows];
#pragma unroll
        for (int j = 0; j < kRows; ++j) p[j] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < kAccBlock; ++kk) {
          const float v = c[(k + kk) * kYCols];
#pragma unroll
          for (int j = kRows - 1; j > 0; --j) t[j] = t[j - 1];
          t[0] = tapsf[4 + k + kk];  // (uniform: a scalar load)
#pragma unroll
          for (int j = 0; j < kRows; ++j) p[j] = fmaf(t[j], v, p[j]);
        }
#pragma unroll
        for (int j = 0; j < kRows; ++j) acc[j] += (double)p[j];
      }
    } else {
      double t[kRows];
#pragma unroll
      for (int j = 0; j < kRows; ++j) t[j] = 0.0;
      for (int k = 0; k < steps; k += kRows) {
#pragma unroll
        for (int kk = 0; kk < kRows; ++kk) {
          const double v = (double)c[(k + kk) * kYCols];
#pragma unroll
          for (int j = kRows - 1; j > 0; --j) t[j] = t[j - 1];
          t[0] = tapsd[4 + k + kk];
#pragma unroll
          for (int j = 0; j < kRows; ++j) acc[j] += t[j] * v;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kRows; ++j)
      if (y0 + r0 + j < ny) out[(size_t)(y0 + r0 + j) * nx + x] = (float)acc[j];
  }
}

// out[x][y] = in[y][x], 64 x 64 tiles through LDS (both sides coalesced): the y pass of a WIDE stencil runs as the x pass
// of the transposed plane (smooth_axis0)
__global__ __launch_bounds__(kBlock) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int ny, int nx) {
  __shared__ float tile[64][65];
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll
  for (int r = ly; r < 64; r += 4)
    if (y0 + r < ny && x0 + lx < nx) tile[r][lx] = in[(size_t)(y0 + r) * nx + x0 + lx];
  __syncthreads();
#pragma unroll
  for (int r = ly; r < 64; r += 4)
    if (x0 + r < nx && y0 + lx < ny) out[(size_t)(x0 + r) * ny + y0 + lx] = tile[lx][r];
}

__global__ void mul_kernel(const float* __restrict__ a,
                           const float* __restrict__ b, float* __restrict__ o,
                           size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) o[i] = a[i] * b[i];
}

__global__ void fill_kernel(float* __restrict__ o, float v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) o[i] = v;
}

// out = denom > 0 ? numer/denom : 0   (map/projection.py:498-499)
__global__ void ratio_kernel(const float* __restrict__ numer,
                             const float* __restrict__ denom,
                             float* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float dn = denom[i];
    o[i] = dn > 0.0f ? numer[i] / dn : 0.0f;
  }
}

// scipy.ndimage._filters._gaussian_kernel1d, order 0; device copy cached in ctx
int get_taps(mrx_ctx* ctx, double sigma, double truncate, int* radius_out,
             const double** d_taps_out) {
  for (auto& slot : ctx->taps)
    if (slot.d_taps && slot.sigma == sigma && slot.truncate == truncate) {
      *radius_out = slot.radius;
      *d_taps_out = slot.d_taps;
      return MRX_OK;
    }
  const int radius = (int)(truncate * sigma + 0.5);
  const int ntap = 2 * radius + 1, slots = gauss_tap_slots(ntap);
  // [slots float64][slots float32]: 4 zeros, the taps, zeros (see gauss_tap_slots)
  std::vector<double> w((size_t)slots + (size_t)(slots + 1) / 2, 0.0);
  const double s2 = sigma * sigma;
  double sum = 0.0;
  for (int k = -radius; k <= radius; ++k) {
    const double v = std::exp(-0.5 / s2 * (double)k * (double)k);
    w[(size_t)(4 + k + radius)] = v;
    sum += v;
  }
  float* wf = reinterpret_cast<float*>(w.data() + slots);
  for (int k = 0; k < ntap; ++k) {
    w[(size_t)(4 + k)] /= sum;
    wf[4 + k] = (float)w[(size_t)(4 + k)];
  }
  auto& slot = ctx->taps[ctx->taps_next];
  ctx->taps_next = (ctx->taps_next + 1) % mrx_ctx::kTapSlots;
  if (slot.d_taps) {
    // a kernel still in flight (on any stream the caller alternates between) may be
    // reading the evicted taps
    MRX_HIP(ctx, hipDeviceSynchronize());
    (void)hipFree(slot.d_taps);
    slot.d_taps = nullptr;
  }
  MRX_HIP(ctx, hipMalloc(&slot.d_taps, w.size() * sizeof(double)));
  MRX_HIP(ctx, hipMemcpyAsync(slot.d_taps, w.data(), w.size() * sizeof(double),
                              hipMemcpyHostToDevice, ctx->stream));
  MRX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // w goes out of scope
  slot.sigma = sigma;
  slot.truncate = truncate;
  slot.radius = radius;
  *radius_out = radius;
  *d_taps_out = slot.d_taps;
  return MRX_OK;
}

constexpr size_t kMaxLds = 128 * 1024;  // of the CU's 160 KiB
constexpr int kTransposeRadius = 64;    // from this radius on the y pass runs as an x pass of the transposed plane (mrx_gauss_smooth2d)

int raise_lds_cap(mrx_ctx* ctx) {
  MRX_LDS_CAP(ctx, gauss_x_kernel<false>, kMaxLds);
  MRX_LDS_CAP(ctx, gauss_x_kernel<true>, kMaxLds);
  MRX_LDS_CAP(ctx, (gauss_y_kernel<false, 4>), kMaxLds);
  MRX_LDS_CAP(ctx, (gauss_y_kernel<true, 4>), kMaxLds);
  MRX_LDS_CAP(ctx, (gauss_y_kernel<false, 8>), kMaxLds);
  MRX_LDS_CAP(ctx, (gauss_y_kernel<true, 8>), kMaxLds);
  return MRX_OK;
}

// MRX_OPT_GAUSS_ACCUM: 0 = blocked float32 sums from radius 16 on, exact float64 below; 1 = exact; 2 = blocked
bool blocked_mode(const mrx_ctx* ctx, int radius) {
  const int mode = ctx->options[MRX_OPT_GAUSS_ACCUM];
  return mode == 2 || (mode == 0 && radius >= 16);
}

int smooth_axis1(mrx_ctx* ctx, const float* in, float* out, int ny, int nx,
                 double sigma, double truncate) {
  int radius = 0;
  const double* d_taps = nullptr;
  int rc = get_taps(ctx, sigma, truncate, &radius, &d_taps);
  if (rc != MRX_OK) return rc;
  const bool blocked = blocked_mode(ctx, radius);
  const float* d_tapsf = reinterpret_cast<const float*>(d_taps + gauss_tap_slots(2 * radius + 1));
  const size_t lds = (size_t)kXRows * (size_t)((kXCols + 2 * radius + 4 + kAccBlock + 3) & ~3) * sizeof(float);
  if ((rc = raise_lds_cap(ctx)) != MRX_OK) return rc;
  if (lds > kMaxLds)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED,
                    "gaussian radius %d along x exceeds the LDS tile", radius);
  dim3 grid(mrx_ceil_div(nx, kXCols), mrx_ceil_div(ny, kXRows));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "ny too large for one launch");
  if (blocked)
    hipLaunchKernelGGL(gauss_x_kernel<true>, grid, dim3(kBlock), lds, ctx->stream, in, out, ny, nx, d_taps, d_tapsf, radius);
  else
    hipLaunchKernelGGL(gauss_x_kernel<false>, grid, dim3(kBlock), lds, ctx->stream, in, out, ny, nx, d_taps, d_tapsf, radius);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int smooth_axis0(mrx_ctx* ctx, const float* in, float* out, int ny, int nx,
                 double sigma, double truncate) {
  int radius = 0;
  const double* d_taps = nullptr;
  int rc = get_taps(ctx, sigma, truncate, &radius, &d_taps);
  if (rc != MRX_OK) return rc;
  if ((rc = raise_lds_cap(ctx)) != MRX_OK) return rc;
  const float* d_tapsf = reinterpret_cast<const float*>(d_taps + gauss_tap_slots(2 * radius + 1));
  // output rows per tile: up to 64 while the image stays under 40 KiB (4 per CU: with eight outputs a thread a wave has
  // the arithmetic to cover its reads); 32 rows while the image fits at all, 16 (four a thread) for the largest radii
  auto image = [&](int rows, int per_thread) { return (size_t)(rows + 2 * radius + kAccBlock + per_thread) * kYCols * sizeof(float); };
  int tile_rows = 64;
  while (tile_rows > 32 && image(tile_rows, 8) > 40 * 1024) tile_rows /= 2;
  if (image(tile_rows, 8) > kMaxLds) tile_rows = 16;
  const int per_thread = tile_rows >= 32 ? 8 : 4;
  const bool blocked = blocked_mode(ctx, radius);
  const size_t lds = image(tile_rows, per_thread);
  if (lds > kMaxLds)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED,
                    "gaussian radius %d along y exceeds the LDS tile", radius);
  dim3 grid(mrx_ceil_div(nx, kYCols), mrx_ceil_div(ny, tile_rows));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "ny too large for one launch");
#define MRX_LAUNCH_Y(B, R) \
  hipLaunchKernelGGL((gauss_y_kernel<B, R>), grid, dim3(kBlock), lds, ctx->stream, in, out, ny, nx, d_taps, d_tapsf, radius, tile_rows)
  if (per_thread == 8) {
    if (blocked) MRX_LAUNCH_Y(true, 8); else MRX_LAUNCH_Y(false, 8);
  } else {
    if (blocked) MRX_LAUNCH_Y(true, 4); else MRX_LAUNCH_Y(false, 4);
  }
#undef MRX_LAUNCH_Y
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

// Linear resampling of every row onto new column positions (model="3d": a layer's own cross-section
// grid, atmosphere/atmosphere.py:208-219, from the grid its process's volume was generated on).
__global__ __launch_bounds__(256) void resample_columns_kernel(const float* __restrict__ in, int n_e, int n_in, size_t ld_in,
                                                               const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                               const float* __restrict__ scale, int n_out,
                                                               float* __restrict__ out, size_t ld_out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_out) return;
  const int i = min(max(idx[j], 0), n_in - 2);  // a caller's index outside 0 .. n_in - 2 must not read outside the row
  const float wj = w[j], sj = scale[j];
  for (int e = blockIdx.y; e < n_e; e += gridDim.y) {
    const float* row = in + (size_t)e * ld_in + i;
    out[(size_t)e * ld_out + j] = sj * fmaf(wj, row[1] - row[0], row[0]);
  }
}

}  // namespace

extern "C" {

int mrx_resample_columns(mrx_ctx* ctx, const float* d_in, int n_e, int n_in, size_t ld_in, const int32_t* d_idx,
                         const float* d_w, const float* d_scale, int n_out, float* d_out, size_t ld_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, n_e >= 0 && n_out >= 0, "negative size");
  if (n_e == 0 || n_out == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_in && d_idx && d_w && d_scale && d_out, "null pointer");
  MRX_REQUIRE(ctx, n_in >= 2 && ld_in >= (size_t)n_in && ld_out >= (size_t)n_out, "need n_in >= 2 and leading dimensions >= the rows");
  const dim3 grid(mrx_ceil_div(n_out, 256), n_e < 1024 ? n_e : 1024);
  hipLaunchKernelGGL(resample_columns_kernel, grid, dim3(256), 0, ctx->stream, d_in, n_e, n_in, ld_in, d_idx, d_w, d_scale, n_out,
                     d_out, ld_out);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_gauss_smooth2d(mrx_ctx* ctx, const float* d_in, float* d_out,
                       float* d_tmp, int ny, int nx, double sigma_y,
                       double sigma_x, double truncate) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, d_in && d_out && d_tmp, "null pointer");
  MRX_REQUIRE(ctx, ny >= 0 && nx >= 0, "negative size");
  MRX_REQUIRE(ctx, sigma_y >= 0.0 && sigma_x >= 0.0 && truncate > 0.0,
              "sigma must be >= 0 and truncate > 0");
  if (ny == 0 || nx == 0) return MRX_OK;
  const bool do_y = sigma_y > 1e-15, do_x = sigma_x > 1e-15;
  const size_t bytes = (size_t)ny * nx * sizeof(float);
  int rc = MRX_OK;
  // A WIDE stencil along y (radius >= kTransposeRadius): the y pass stages 2 radius + rows-per-tile rows per tile -- ten
  // times the rows it writes at radius 128, in an LDS image that leaves two workgroups to a CU -- where the x pass, whose
  // window is a row segment, stays near its arithmetic: 8192^2, sigma 32: 1.68 ms along y against 0.61 along x.  So the
  // plane is transposed (0.1 ms either way), filtered along x, and transposed back.  d_out serves as the second scratch
  // plane (it is written last); in place (d_in == d_out) the input is consumed by the first transpose.
  const bool wide_y = do_y && (int)(truncate * sigma_y + 0.5) >= kTransposeRadius && ny <= 65535 * 64 && nx <= 65535 * kXRows;
  if (wide_y) {
    const dim3 g_in(mrx_ceil_div(nx, 64), mrx_ceil_div(ny, 64)), g_back(mrx_ceil_div(ny, 64), mrx_ceil_div(nx, 64));
    hipLaunchKernelGGL(transpose_kernel, g_in, dim3(kBlock), 0, ctx->stream, d_in, d_tmp, ny, nx);  // tmp: [nx][ny]
    MRX_CHECK_LAUNCH(ctx);
    rc = smooth_axis1(ctx, d_tmp, d_out, nx, ny, sigma_y, truncate);                                // out: [nx][ny], filtered along y
    if (rc != MRX_OK) return rc;
    hipLaunchKernelGGL(transpose_kernel, g_back, dim3(kBlock), 0, ctx->stream, d_out, d_tmp, nx, ny);  // tmp: [ny][nx]
    MRX_CHECK_LAUNCH(ctx);
    if (do_x) return smooth_axis1(ctx, d_tmp, d_out, ny, nx, sigma_x, truncate);
    MRX_HIP(ctx, hipMemcpyAsync(d_out, d_tmp, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MRX_OK;
  }
  if (do_y && do_x) {
    rc = smooth_axis0(ctx, d_in, d_tmp, ny, nx, sigma_y, truncate);
    if (rc == MRX_OK) rc = smooth_axis1(ctx, d_tmp, d_out, ny, nx, sigma_x, truncate);
  } else if (do_y || do_x) {
    // a single pass cannot run in place: go through d_tmp when in == out
    const float* src = d_in;
    if (d_in == d_out) {
      MRX_HIP(ctx, hipMemcpyAsync(d_tmp, d_in, bytes, hipMemcpyDeviceToDevice,
                                  ctx->stream));
      src = d_tmp;
    }
    rc = do_y ? smooth_axis0(ctx, src, d_out, ny, nx, sigma_y, truncate)
              : smooth_axis1(ctx, src, d_out, ny, nx, sigma_x, truncate);
  } else if (d_in != d_out) {
    MRX_HIP(ctx, hipMemcpyAsync(d_out, d_in, bytes, hipMemcpyDeviceToDevice,
                                ctx->stream));
  }
  return rc;
}

int mrx_map_smooth(mrx_ctx* ctx, const float* d_data, const float* d_weight,
                   float* d_out, float* d_denom_out, float* d_tmp, int ny,
                   int nx, double sigma_y, double sigma_x) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, d_data && d_out && d_tmp, "null pointer");
  MRX_REQUIRE(ctx, ny >= 0 && nx >= 0, "negative size");
  if (ny == 0 || nx == 0) return MRX_OK;
  const size_t n = (size_t)ny * nx;
  float* t0 = d_tmp;      // pass scratch
  float* t1 = d_tmp + n;  // data*weight, then denom when the caller wants none
  const int blocks = (int)((n + kBlock - 1) / kBlock < 4096
                               ? (n + kBlock - 1) / kBlock
                               : 4096);
  int rc;
  if (d_weight) {
    hipLaunchKernelGGL(mul_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream,
                       d_data, d_weight, t1, n);
    MRX_CHECK_LAUNCH(ctx);
    rc = mrx_gauss_smooth2d(ctx, t1, d_out, t0, ny, nx, sigma_y, sigma_x, 4.0);
    if (rc != MRX_OK) return rc;
    float* denom = d_denom_out ? d_denom_out : t1;
    rc = mrx_gauss_smooth2d(ctx, d_weight, denom, t0, ny, nx, sigma_y, sigma_x, 4.0);
    if (rc != MRX_OK) return rc;
    hipLaunchKernelGGL(ratio_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream,
                       d_out, denom, d_out, n);
    MRX_CHECK_LAUNCH(ctx);
  } else {
    // weight == 1: denom = G(1) is 1 -- the taps are normalised in float64 and the reflected boundary keeps a
    // constant constant, so every pixel's float32 denominator is exactly 1.0f and numer / denom is numer: one filter
    // instead of two and no quotient pass (1024^2, sigma 2 px: 0.040 -> 0.017 ms)
    rc = mrx_gauss_smooth2d(ctx, d_data, d_out, t0, ny, nx, sigma_y, sigma_x, 4.0);
    if (rc != MRX_OK) return rc;
    if (d_denom_out) {
      hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, d_denom_out, 1.0f, n);
      MRX_CHECK_LAUNCH(ctx);
    }
  }
  return MRX_OK;
}

}  // extern "C"
