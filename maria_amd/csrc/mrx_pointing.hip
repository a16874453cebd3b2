// Full-rate detector pointing and the linear upsample of the coarse pwv (a14): the two small
// full-rate passes beside the spline writers, on the same TOD tile.
#include "mrx_internal.h"
#include "mrx_tile.h"

namespace {

// Linear interpolation of the coarse pwv (float64, time-major) to the full
// rate: sim/atmosphere.py:30-37.  Same tiling as the cubic kernel without the
// LDS stage; only the optional map/cmb consumers need it.
__global__ __launch_bounds__(kBlock) void linear_upsample_kernel(
    const double* __restrict__ pwv, int D, int n, double ta0, double inv_dta,
    double dta, const double* __restrict__ t, int T, float* __restrict__ out,
    size_t ld) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= T) return;
  const double x = (t[s] - ta0) * inv_dta;
  int jj = (int)floor(fmin(fmax(x, -1.0), 2.0e9));
  jj = min(max(jj, 0), n - 2);
  const double u = (t[s] - (ta0 + (double)jj * dta)) * inv_dta;
  const int d0 = blockIdx.y * kTileDet;
  const int nd = min(kTileDet, D - d0);
  for (int dl = 0; dl < nd; ++dl) {
    const int d = d0 + dl;
    const double y0 = pwv[(size_t)jj * D + d];
    const double y1 = pwv[(size_t)(jj + 1) * D + d];
    out[(size_t)d * ld + s] = (float)(y0 + u * (y1 - y0));
  }
}

// Full-rate detector pointing (coords/coordinates.py:378-386 at the sample rate,
// sim/observation.py:55-58): az/el [D][T] float32 from the boresight and the
// detector offsets, the float32 chain of coords/transforms.py:10-29.  Same tile
// as the TOD writer: 16 rows x 1024 samples, two 16-byte stores per row.
__global__ __launch_bounds__(kBlock) void pointing_broadcast_kernel(
    const float* __restrict__ az, const float* __restrict__ el, int T,
    const float* __restrict__ dxs, const float* __restrict__ dys, int D,
    float* __restrict__ az_out, float* __restrict__ el_out, size_t ld,
    int vec_ok) {
  __shared__ float4 pdet[kTileDet];  // sin(r)cos(p), cos(r), sin(r)sin(p)
  const int d0 = blockIdx.y * kTileDet;
  const int sb = blockIdx.x * kTileSamples + threadIdx.x * kSamplesPerThread;
  const int nd = min(kTileDet, D - d0);
  if ((int)threadIdx.x < nd) {
    const float dx = dxs[d0 + threadIdx.x], dy = dys[d0 + threadIdx.x];
    const float r = sqrtf(dx * dx + dy * dy);
    const float p = atan2f(-dx, -dy);
    const float sr = sinf(r);
    pdet[threadIdx.x] = make_float4(__fmul_rn(sr, cosf(p)), cosf(r), __fmul_rn(sr, sinf(p)), 0.f);
  }
  float ca[kSamplesPerThread], sa[kSamplesPerThread], zz[kSamplesPerThread];
#pragma unroll
  for (int q = 0; q < kSamplesPerThread; ++q) {
    const int s = min(sb + q, T - 1);
    const float a = el[s] - 1.57079637050628662109375f;
    ca[q] = cosf(a);
    sa[q] = sinf(a);
    zz[q] = az[s];
  }
  __syncthreads();
  const bool full = (sb + kSamplesPerThread <= T) && vec_ok;
  for (int dl = 0; dl < nd; ++dl) {
    const float4 c = pdet[dl];
    float oa[kSamplesPerThread], oe[kSamplesPerThread];
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) {
      const float re = __fsub_rn(__fmul_rn(c.x, ca[q]), __fmul_rn(c.y, sa[q]));
      const float im = __fadd_rn(__fmul_rn(c.x, sa[q]), __fmul_rn(c.y, ca[q]));
      oa[q] = __fadd_rn(atan2f(c.z, re), zz[q]);
      oe[q] = asinf(im);
    }
    float* da = az_out + (size_t)(d0 + dl) * ld + sb;
    float* de = el_out + (size_t)(d0 + dl) * ld + sb;
    if (full) {
      const vfloat4 va = {oa[0], oa[1], oa[2], oa[3]};
      const vfloat4 ve = {oe[0], oe[1], oe[2], oe[3]};
      __builtin_nontemporal_store(va, reinterpret_cast<vfloat4*>(da));
      __builtin_nontemporal_store(ve, reinterpret_cast<vfloat4*>(de));
    } else {
#pragma unroll
      for (int q = 0; q < kSamplesPerThread; ++q)
        if (sb + q < T) {
          da[q] = oa[q];
          de[q] = oe[q];
        }
    }
  }
}

}  // namespace

extern "C" {

int mrx_pointing_broadcast(mrx_ctx* ctx, const float* d_az, const float* d_el,
                           int T, const float* d_dx, const float* d_dy, int D,
                           float* d_az_out, float* d_el_out, size_t ld_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_az && d_el && d_dx && d_dy && d_az_out && d_el_out,
              "null pointer");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  dim3 grid(mrx_ceil_div(T, kTileSamples), mrx_ceil_div(D, kTileDet));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  const int vec_ok = (ld_out % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(d_az_out) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(d_el_out) & 15u) == 0);
  hipLaunchKernelGGL(pointing_broadcast_kernel, grid, dim3(kBlock), 0,
                     ctx->stream, d_az, d_el, T, d_dx, d_dy, D, d_az_out,
                     d_el_out, ld_out, vec_ok);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_linear_upsample(mrx_ctx* ctx, const double* d_pwv, int D, int Ta,
                        double ta0, double dta, const double* d_t, int T,
                        float* d_out, size_t ld_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_pwv && d_t && d_out, "null pointer");
  MRX_REQUIRE(ctx, dta > 0.0, "coarse step must be positive");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  MRX_REQUIRE(ctx, Ta >= 2, "linear interpolation needs 2 coarse samples");
  dim3 grid(mrx_ceil_div(T, kBlock), mrx_ceil_div(D, kTileDet));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  hipLaunchKernelGGL(linear_upsample_kernel, grid, dim3(kBlock), 0, ctx->stream,
                     d_pwv, D, Ta, ta0, 1.0 / dta, dta, d_t, T, d_out, ld_out);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

}  // extern "C"
