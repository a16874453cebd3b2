// TOD.to("K_RJ") (tod/tod.py:106-142, calibration/functions.py:73-90) in its three forms: fused into the spline's
// evaluation (spline_upsample_krj_kernel: per sample), on the coarse loading before the spline (coarse_krj_kernel),
// and in place on a finished full-rate field (tod_krj_kernel, both directions).  The shared machinery is mrx_krj.h.
#include <type_traits>

#include "mrx_internal.h"
#include "mrx_krj.h"

namespace {

// the static part of the K_RJ writer's LDS, carved from the dynamic buffer behind the cell table (with a static size the
// compiler derives the occupancy from it and ignores the register bound below, as for the fused writer)
struct KrjWriterLds {
  static constexpr int kMaxKnots = 64, kPitch = kMaxKnots + 1;
  float2 tile[kTileDet * kPitch];
  CalDet cdet[kTileDet];
  float red[12];
  int row_lds[kTileDet];
};

// 5 waves per SIMD = 96 registers: the evaluation loop needs 95; the per-sample instance of the loop (KrjSamples::curved,
// rare) would take the kernel to 98 and a wave per SIMD away (K_RJ writer 2.6 -> 3.0 ms), so it spills what is over
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5))) void spline_upsample_krj_kernel(
    const float2* __restrict__ ym, int D, int n, double ta0, double inv_dta,
    const double* __restrict__ t, int T, const float* __restrict__ scale,
    const int32_t* __restrict__ rows, const float* __restrict__ bore_el,
    const float* __restrict__ dxs, const float* __restrict__ dys,
    const int32_t* __restrict__ band, const float* __restrict__ cal_axis,
    const float* __restrict__ cal_values, int n_el, int n_bands,
    float* __restrict__ out, size_t ld, int vec_ok, int groups) {
  constexpr int kMaxKnots = KrjWriterLds::kMaxKnots;  // 8 KiB image: the arithmetic of this writer wants the occupancy
  constexpr int kPitch = KrjWriterLds::kPitch;
  extern __shared__ __align__(16) float4 cal_cells[];  // [n_bands][n_el - 1], see stage_cal_cells; then KrjWriterLds
  KrjWriterLds& L = *reinterpret_cast<KrjWriterLds*>(cal_cells + (size_t)n_bands * (n_el - 1));
  float2* tile = L.tile;
  CalDet* cdet = L.cdet;
  float* red = L.red;
  int* row_lds = L.row_lds;  // destination rows of the group (see spline_upsample_kernel)
  auto row_of = [&](int dl, int d) -> size_t { return rows ? (size_t)row_lds[dl] : (size_t)d; };

  const int s_tile = blockIdx.x * kTileSamples;
  const int sb = s_tile + threadIdx.x * kSamplesPerThread;

  // per-sample interval, weights and boresight: computed once, reused for `groups` tiles of
  // 16 detector rows each (the float64 prologue is a third of a single tile's instructions)
  SampleWeights w;
  sample_weights(t, sb, T, n, ta0, inv_dta, w);
  const KrjSamples ks = krj_prologue(cal_cells, red, bore_el, T, sb, cal_axis, cal_values, n_el, n_bands);

  const int s_last = min(s_tile + kTileSamples, T) - 1;
  const int jmin = interval_of((t[s_tile] - ta0) * inv_dta, n);
  const int jmax = interval_of((t[s_last] - ta0) * inv_dta, n) + 1;
  const int K = jmax - jmin + 1;
  const bool use_lds = K <= kMaxKnots;
  const bool full = (sb + kSamplesPerThread <= T) && vec_ok;
  int r[kSamplesPerThread];
#pragma unroll
  for (int q = 0; q < kSamplesPerThread; ++q) r[q] = min(max(w.j[q] - jmin, 0), max(K - 2, 0));

  for (int g = 0; g < groups; ++g) {
    const int d0 = (blockIdx.y * groups + g) * kTileDet;
    if (d0 >= D) break;
    const int nd = min(kTileDet, D - d0);
    if (g > 0) __syncthreads();  // the previous group is done with tile[] and cdet[]
    if (use_lds) {
      const int dl = threadIdx.x & (kTileDet - 1);
      const int d = d0 + dl;
      for (int rr = threadIdx.x / kTileDet; rr < K; rr += kBlock / kTileDet) {
        float2 v = make_float2(0.f, 0.f);
        if (d < D) v = ym[(size_t)(jmin + rr) * D + d];
        tile[dl * kPitch + rr] = v;
      }
    }
    krj_stage_rows(cdet, red, dxs, dys, band, scale, n_bands, d0, nd, cal_cells, n_el);
    if (rows && (int)threadIdx.x < nd) row_lds[threadIdx.x] = rows[d0 + threadIdx.x];
    __syncthreads();
    // the loop body is instantiated once per knot source so that each instance
    // addresses one memory space (a runtime select would force flat loads)
    auto body = [&](auto from_lds) {
    for (int dl = 0; dl < nd; ++dl) {
      const CalDet c = cdet[dl];
      const float4* C = cal_cells + c.band * (n_el - 1);
      float o[kSamplesPerThread], sv[kSamplesPerThread];
#pragma unroll
      for (int q = 0; q < kSamplesPerThread; ++q) {
        float2 k0, k1;
        if constexpr (decltype(from_lds)::value) {
          k0 = tile[dl * kPitch + r[q]];
          k1 = tile[dl * kPitch + r[q] + 1];
        } else {
          k0 = ym[(size_t)w.j[q] * D + d0 + dl];
          k1 = ym[(size_t)(w.j[q] + 1) * D + d0 + dl];
        }
        sv[q] = c.scale * spline_eval(w, q, k0, k1);
      }
      krj_row<false>(c, C, n_el, ks, sv, o, bore_el, sb, T, cal_cells, cal_axis);
      float* dst = out + row_of(dl, d0 + dl) * ld + sb;
      if (full) {
        const vfloat4 v = {o[0], o[1], o[2], o[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(dst));
      } else {
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q)
          if (sb + q < T) dst[q] = o[q];
      }
    }
    };
    if (use_lds) body(std::true_type{}); else body(std::false_type{});
  }
}

// TOD.to("K_RJ") on the COARSE grid, before the spline: out[j][d] = loading[j][d] /
// den_band(d)(el_det(d, j)) with the detector elevation of the full formula at coarse step j.
// The reference divides the full-rate spline S[y](t) by g(t) = den(el_det(t)) sample by sample;
// S[y / g] differs from S[y] / g by the spline's interpolation error on g alone (g is smooth in
// time but for the kinks where the elevation crosses a node of the table's axis): the host
// bounds it before choosing this form (DevicePath.coarse_krj_bound) and the TOD is then written
// by the plain pW writer.  Time-major like the sampler's output: lanes are detectors.
constexpr int kCoarseKrjSteps = 32;

// (`loading` and `out` may be the same buffer -- the pipelined run converts a block's coarse loading in place --: no
// __restrict__ on either; every element is read once, by the thread that then writes it)
__global__ __launch_bounds__(kBlock) void coarse_krj_kernel(
    const float* loading, int D, int Ta, const float* __restrict__ bore_el,
    const float* __restrict__ dxs, const float* __restrict__ dys, const int32_t* __restrict__ band,
    const float* __restrict__ cal_axis, const float* __restrict__ cal_values, int n_el, int n_bands,
    float* out, float* __restrict__ tail, int tail_first, size_t ld_tail) {
  extern __shared__ __align__(16) float4 cal_cells[];  // [n_bands][n_el - 1], see stage_cal_cells
  __shared__ float2 trig[kCoarseKrjSteps];  // (cos, sin) of (boresight elevation - pi/2) of the block's steps
  stage_cal_cells(cal_cells, cal_axis, cal_values, n_el, n_bands);
  const int j0 = blockIdx.y * kCoarseKrjSteps, j1 = min(j0 + kCoarseKrjSteps, Ta);
  if ((int)threadIdx.x < j1 - j0) {
    const float a = bore_el[j0 + threadIdx.x] - 1.57079637050628662109375f;
    trig[threadIdx.x] = make_float2(cosf(a), sinf(a));
  }
  __syncthreads();
  const int d = blockIdx.x * kBlock + threadIdx.x;
  if (d >= D) return;
  const CalDet c = make_cal_det(dxs[d], dys[d], min(max(band[d], 0), n_bands - 1), 1.0f);
  const float4* C = cal_cells + c.band * (n_el - 1);
  const float el_first = cal_cells[0].x, el_last = cal_axis[n_el - 1], el_inv = cal_cells[0].z;
  // eight steps' loads in flight per thread (one at a time, the loop was a chain of 32 memory latencies: 65 us for a block
  // of 2 500 rows beside the TOD writer, on the sampler's stream of the pipelined step)
  constexpr int kAhead = 8;
  for (int jb = j0; jb < j1; jb += kAhead) {
    float v[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) v[k] = loading[(size_t)min(jb + k, j1 - 1) * D + d];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      const int j = jb + k;
      if (j >= j1) break;
      // coords/transforms.py:20-28 in float32: im = sin(el_det), el_det = asin(im)
      const float2 cs = trig[j - j0];
      const float im = det_sin_elevation(c.a_re, c.a_im, cs.x, cs.y);
      const float den = den_lookup(asinf(im), C, n_el, el_first, el_last, el_inv);
      out[(size_t)j * D + d] = v[k] * __builtin_amdgcn_rcpf(den);
      if (tail && j >= tail_first) tail[(size_t)(j - tail_first) * ld_tail + d] = v[k];  // (uniform: j is the block's)
    }
  }
}

// TOD.to("K_RJ") of a field that is already at the full rate (noise, map, cmb;
// tod/tod.py:106-142), in place: data[row(d)][s] *= scale_d / den_band(d)(el(d, s)).
// Same tile as the fused writer: 16 detectors x 1024 samples per workgroup, 16-byte
// loads and non-temporal stores; 8 B of HBM traffic per sample.
template <bool kInverse>
__global__ __launch_bounds__(kBlock) void tod_krj_kernel(
    float* __restrict__ data, size_t ld, int D, int T, const float* __restrict__ scale,
    const int32_t* __restrict__ rows, const float* __restrict__ bore_el,
    const float* __restrict__ dxs, const float* __restrict__ dys,
    const int32_t* __restrict__ band, const float* __restrict__ cal_axis,
    const float* __restrict__ cal_values, int n_el, int n_bands, int vec_ok) {
  extern __shared__ __align__(16) float4 cal_cells[];
  __shared__ CalDet cdet[kTileDet];
  __shared__ float red[12];
  const int s_tile = blockIdx.x * kTileSamples;
  const int d0 = blockIdx.y * kTileDet;
  const int sb = s_tile + threadIdx.x * kSamplesPerThread;
  const int nd = min(kTileDet, D - d0);
  __shared__ int row_lds[kTileDet];
  KrjSamples ks = krj_prologue(cal_cells, red, bore_el, T, sb, cal_axis, cal_values, n_el, n_bands);
  krj_stage_rows(cdet, red, dxs, dys, band, scale, n_bands, d0, nd, cal_cells, n_el);
  if ((int)threadIdx.x < nd) row_lds[threadIdx.x] = rows ? rows[d0 + threadIdx.x] : d0 + (int)threadIdx.x;
  __syncthreads();
  if (sb >= T) return;
  const bool full = (sb + kSamplesPerThread <= T) && vec_ok;
  for (int dl = 0; dl < nd; ++dl) {
    const CalDet c = cdet[dl];
    const float4* C = cal_cells + c.band * (n_el - 1);
    float* row = data + (size_t)row_lds[dl] * ld + sb;
    float v[kSamplesPerThread];
    if (full) {
      const vfloat4 x = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(row));
      v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    } else {
#pragma unroll
      for (int q = 0; q < kSamplesPerThread; ++q) v[q] = sb + q < T ? row[q] : 0.0f;
    }
    float sv[kSamplesPerThread];
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) sv[q] = c.scale * v[q];
    krj_row<kInverse>(c, C, n_el, ks, sv, v, bore_el, sb, T, cal_cells, cal_axis);
    if (full) {
      const vfloat4 x = {v[0], v[1], v[2], v[3]};
      __builtin_nontemporal_store(x, reinterpret_cast<vfloat4*>(row));
    } else {
#pragma unroll
      for (int q = 0; q < kSamplesPerThread; ++q)
        if (sb + q < T) row[q] = v[q];
    }
  }
}


}  // namespace

extern "C" {

int mrx_spline_upsample_krj(mrx_ctx* ctx, const float* d_ym, int D, int Ta,
                            double ta0, double dta, const double* d_t, int T,
                            const float* d_scale, const int32_t* d_rows,
                            const float* d_bore_el, const float* d_dx,
                            const float* d_dy, const int32_t* d_band,
                            const float* d_cal_axis_el,
                            const float* d_cal_values, int n_el, int n_bands,
                            float* d_out, size_t ld_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_ym && d_t && d_out, "null pointer");
  MRX_REQUIRE(ctx, d_bore_el && d_dx && d_dy && d_band && d_cal_axis_el &&
                       d_cal_values,
              "null calibration pointer");
  // the cell table lives in LDS as one float4 per (band, cell): 96 KiB beside the kernels' static images
  MRX_REQUIRE(ctx, n_el >= 2 && n_bands >= 1 && (size_t)(n_el - 1) * n_bands <= 6144,
              "calibration tables need 2 <= n_el and (n_el-1)*n_bands <= 6144");
  MRX_REQUIRE(ctx, dta > 0.0, "coarse step must be positive");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  if (Ta < 4)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED,
                    "cubic interpolation needs at least 4 coarse samples");
  // detector tiles per workgroup: the per-sample prologue is shared by all of them
  int groups = ctx->options[MRX_OPT_UPSAMPLE_GROUPS];
  if (groups <= 0) groups = 4;  // measured on atlast_10k: 1 -> 2.69 ms, 2 -> 2.55, 4 -> 2.48
  while (groups > 1 && (long long)mrx_ceil_div(T, kTileSamples) *
                               mrx_ceil_div(D, kTileDet * groups) < 4LL * 256 * 4)
    groups /= 2;  // keep the chip full on small problems
  dim3 grid(mrx_ceil_div(T, kTileSamples), mrx_ceil_div(D, kTileDet * groups));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  const int vec_ok =
      (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0);
  const size_t lds = sizeof(float4) * (size_t)(n_el - 1) * n_bands + sizeof(KrjWriterLds);  // the cell table, the knot image
  MRX_LDS_CAP(ctx, spline_upsample_krj_kernel, lds);
  hipLaunchKernelGGL(spline_upsample_krj_kernel, grid, dim3(kBlock), lds,
                     ctx->stream, reinterpret_cast<const float2*>(d_ym), D, Ta,
                     ta0, 1.0 / dta, d_t, T, d_scale, d_rows, d_bore_el, d_dx,
                     d_dy, d_band, d_cal_axis_el, d_cal_values, n_el, n_bands,
                     d_out, ld_out, vec_ok, groups);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_coarse_to_krj(mrx_ctx* ctx, const float* d_loading, int D, int Ta, const float* d_bore_el_coarse,
                      const float* d_dx, const float* d_dy, const int32_t* d_band,
                      const float* d_cal_axis_el, const float* d_cal_values, int n_el, int n_bands,
                      float* d_out) {
  return mrx_coarse_to_krj_keep_tail(ctx, d_loading, D, Ta, d_bore_el_coarse, d_dx, d_dy, d_band, d_cal_axis_el, d_cal_values,
                                     n_el, n_bands, d_out, nullptr, 0, 0);
}

int mrx_coarse_to_krj_keep_tail(mrx_ctx* ctx, const float* d_loading, int D, int Ta, const float* d_bore_el_coarse,
                                const float* d_dx, const float* d_dy, const int32_t* d_band,
                                const float* d_cal_axis_el, const float* d_cal_values, int n_el, int n_bands,
                                float* d_out, float* d_tail_pw, int tail_knots, size_t ld_tail) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && Ta >= 0, "negative size");
  MRX_REQUIRE(ctx, tail_knots >= 0 && tail_knots <= Ta, "tail_knots must lie in [0, Ta]");
  if (!d_tail_pw || tail_knots == 0) { d_tail_pw = nullptr; tail_knots = 0; }
  MRX_REQUIRE(ctx, !d_tail_pw || ld_tail >= (size_t)D, "ld_tail is shorter than a row of D detectors");
  if (D == 0 || Ta == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_loading && d_out && d_bore_el_coarse && d_dx && d_dy && d_band && d_cal_axis_el && d_cal_values,
              "null pointer");
  // the cell table lives in LDS as one float4 per (band, cell): 96 KiB beside the kernels' static images
  MRX_REQUIRE(ctx, n_el >= 2 && n_bands >= 1 && (size_t)(n_el - 1) * n_bands <= 6144,
              "calibration tables need 2 <= n_el and (n_el-1)*n_bands <= 6144");
  const dim3 grid(mrx_ceil_div(D, kBlock), mrx_ceil_div(Ta, kCoarseKrjSteps));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "Ta too large for one launch");
  const size_t lds = sizeof(float4) * (size_t)(n_el - 1) * n_bands;
  MRX_LDS_CAP(ctx, coarse_krj_kernel, lds);
  hipLaunchKernelGGL(coarse_krj_kernel, grid, dim3(kBlock), lds, ctx->stream, d_loading, D, Ta, d_bore_el_coarse,
                     d_dx, d_dy, d_band, d_cal_axis_el, d_cal_values, n_el, n_bands, d_out, d_tail_pw, Ta - tail_knots, ld_tail);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

static int tod_convert(mrx_ctx* ctx, bool inverse, float* d_data, size_t ld, int D, int T,
                       const float* d_scale, const int32_t* d_rows,
                       const float* d_bore_el, const float* d_dx, const float* d_dy,
                       const int32_t* d_band, const float* d_cal_axis_el,
                       const float* d_cal_values, int n_el, int n_bands) {
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_data && d_bore_el && d_dx && d_dy && d_band && d_cal_axis_el && d_cal_values,
              "null pointer");
  // the cell table lives in LDS as one float4 per (band, cell): 96 KiB beside the kernels' static images
  MRX_REQUIRE(ctx, n_el >= 2 && n_bands >= 1 && (size_t)(n_el - 1) * n_bands <= 6144,
              "calibration tables need 2 <= n_el and (n_el-1)*n_bands <= 6144");
  MRX_REQUIRE(ctx, ld >= (size_t)T, "ld smaller than T");
  dim3 grid(mrx_ceil_div(T, kTileSamples), mrx_ceil_div(D, kTileDet));
  MRX_REQUIRE(ctx, grid.y <= 65535u, "D too large for one launch");
  const int vec_ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_data) & 15u) == 0);
  const size_t lds = sizeof(float4) * (size_t)(n_el - 1) * n_bands;  // the cell table
  if (inverse) MRX_LDS_CAP(ctx, tod_krj_kernel<true>, lds); else MRX_LDS_CAP(ctx, tod_krj_kernel<false>, lds);
  if (inverse)
    hipLaunchKernelGGL(tod_krj_kernel<true>, grid, dim3(kBlock), lds, ctx->stream, d_data, ld, D, T,
                       d_scale, d_rows, d_bore_el, d_dx, d_dy, d_band, d_cal_axis_el,
                       d_cal_values, n_el, n_bands, vec_ok);
  else
    hipLaunchKernelGGL(tod_krj_kernel<false>, grid, dim3(kBlock), lds, ctx->stream, d_data, ld, D, T,
                       d_scale, d_rows, d_bore_el, d_dx, d_dy, d_band, d_cal_axis_el,
                       d_cal_values, n_el, n_bands, vec_ok);
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_tod_to_krj(mrx_ctx* ctx, float* d_data, size_t ld, int D, int T,
                   const float* d_scale, const int32_t* d_rows,
                   const float* d_bore_el, const float* d_dx, const float* d_dy,
                   const int32_t* d_band, const float* d_cal_axis_el,
                   const float* d_cal_values, int n_el, int n_bands) {
  MRX_ENTER(ctx);
  return tod_convert(ctx, false, d_data, ld, D, T, d_scale, d_rows, d_bore_el, d_dx, d_dy, d_band,
                     d_cal_axis_el, d_cal_values, n_el, n_bands);
}

int mrx_tod_from_krj(mrx_ctx* ctx, float* d_data, size_t ld, int D, int T,
                     const float* d_scale, const int32_t* d_rows,
                     const float* d_bore_el, const float* d_dx, const float* d_dy,
                     const int32_t* d_band, const float* d_cal_axis_el,
                     const float* d_cal_values, int n_el, int n_bands) {
  MRX_ENTER(ctx);
  return tod_convert(ctx, true, d_data, ld, D, T, d_scale, d_rows, d_bore_el, d_dx, d_dy, d_band,
                     d_cal_axis_el, d_cal_values, n_el, n_bands);
}

}  // extern "C"
