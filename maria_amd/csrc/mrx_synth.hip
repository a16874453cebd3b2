#include <type_traits>

#include "mrx_internal.h"
#define MRX_PX_CONTRACT_FAST_AFTER  // (the sampler's code keeps its two roundings; the rest of this TU contracts)
#include "mrx_sample_px.h"  // the sampler role of atm_tod_kernel
#include "mrx_spline_tile.h"  // the writer role
#include "mrx_krj.h"          // the K_RJ form's division in the sampler role

namespace {

// ---------------------------------------------------------------------------
// Atmosphere -> TOD in ONE launch (mrx_atm_synthesize): the sampler and the writer as two ROLES of one grid, the
// hand-over on the device.  Replaces, for one observation, screens -> [sampler of block b on a side stream |
// writer of block b behind an event] x blocks: there the writer of block 0 cannot start before the whole first
// block is sampled (0.2 ms of the 2.3-ms step of atlast_10k with HBM idle) and every launch boundary drains and
// refills the chip (4-14 per step).
//
//   * the first `n_sampler_wgs` workgroups (lowest block indices: the dispatcher hands them out first, so they are
//     resident before any writer -- and they never wait for anything, so the grid always drains) run
//     px_sample_items over the detector blocks in order: block b's coarse loading is its own [Ta][rows] array;
//   * the loading leaves the sampler's CUs as write-through (sc1) 16-byte stores, every 128-byte line written whole
//     by one store instruction (px_sample_items<..., kWriteThrough>); after each finished work item every wave
//     drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, and ONE lane adds 1 to done[b]
//     (agent scope, relaxed);
//   * every other workgroup is a writer: it takes tile numbers from a queue (one agent-scope atomic per tile; tiles
//     are numbered block by block, time tile fastest -- the order the two-dimensional grid of
//     spline_upsample_fused_kernel is dispatched in), and before its FIRST tile of a block one lane polls done[b]
//     (global_load_dword sc1, s_sleep between) until all of the block's items are in, then the workgroup meets at a
//     barrier; the tile stages its knots with sc1 loads (past the CU's L1), then fused_writer_tile as in the
//     stand-alone kernel.
// The per-XCD L2s are not coherent and a CU's L1 is not refreshed by other CUs' stores.  Write-through stores +
// drained waves + one agent-scope add per workgroup on the producer's side, an sc1 poll + a workgroup barrier + sc1
// loads on the consumer's, is one of the forms MI355X_MICROARCH.md lists as measured valid on gfx950 (its table of
// hand-offs without fences, third row).  The first version used the fenced form (release fence per work item, acquire
// per writer and block): correct too, but a release writes back the XCD's whole L2 under a streaming writer --
// 0.18 ms of a 1.9-ms step in fences, 0.15-0.3 more in waits.
// A poll gives up after `poll_limit` tries and raises MRX_FLAG_HANDOVER (the host then fails the call): a bound,
// not a path -- it cannot trigger while the sampler role is resident.
// Results: the same bits as mrx_atm_sample + mrx_spline_upsample_fused per block (same bodies, same order of
// operations; tests/test_gpu_synthesize.py).
// kKrj (mrx_atm_synthesize_krj): TOD.to("K_RJ") on the coarse grid, in the sampler role's epilogue -- what
// coarse_krj_kernel does to a finished block between the two calls (same functions, same operands: the same bits),
// the last knots kept aside in pW for the samples past the last knot.
// who samples what: blocks [end[p-1], end[p]) by the first wgs[p] sampler workgroups (wgs descending)
struct SynthPhases {
  int n;
  int end[4];
  int wgs[4];
};

// The calibration of the K_RJ form (DevicePath.set_calibration: the band's denominators on the elevation axis).
struct SynthCal {
  const float* dx;      // [D] detector offsets as the calibration holds them
  const float* dy;
  const float* axis;    // [n_el]
  const float* values;  // [n_bands][n_el]
  int n_el, n_bands;
  float* tail;          // [tail_knots][ld_tail] the last knots in pW, or null
  int tail_first;       // Ta - tail_knots
  size_t ld_tail;
};

// What the sampler role does beside sampling: the K_RJ division per coarse sample and the block counters.
template <bool kKrj>
struct SynthHooks {
  int* ctl;
  const int32_t* band;
  SynthCal cal;
  const float4* cells;  // the staged table (stage_cal_cells)
  float el_first, el_last, el_inv;
  float a_re = 0.0f, a_im = 0.0f;  // of the work item's detector (make_cal_det)
  const float4* C = nullptr;
  __device__ __forceinline__ void item(int, int d) {
    if (kKrj) {
      const CalDet c = make_cal_det(cal.dx[d], cal.dy[d], min(max(band[d], 0), cal.n_bands - 1), 1.0f);
      a_re = c.a_re;
      a_im = c.a_im;
      C = cells + c.band * (cal.n_el - 1);
    }
  }
  // coarse_krj_kernel's arithmetic: im = sin(el_det) from the step's (cos, sin) of (boresight elevation - pi/2)
  __device__ __forceinline__ float value(float v, const float4& bt, int t, int d, bool real) const {
    if (!kKrj) return v;
    const float im = __fadd_rn(__fmul_rn(a_re, bt.y), __fmul_rn(a_im, bt.x));
    const float den = den_lookup(asinf(im), C, cal.n_el, el_first, el_last, el_inv);
    if (cal.tail && real && t >= cal.tail_first) cal.tail[(size_t)(t - cal.tail_first) * cal.ld_tail + d] = v;
    return v * __builtin_amdgcn_rcpf(den);
  }
  __device__ __forceinline__ void done(int blk) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's (write-through) stores of the item are out
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctl + 32 + blk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

template <bool kLdsTables, bool kHasScale, int kMaxKnots, int kG, bool kKrj>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MRX_WRITER_WAVES, MRX_WRITER_WAVES))) void atm_tod_kernel(
    const mrx_layer_fast* __restrict__ fast, const mrx_layer_px* __restrict__ lpx, int n_layers,
    const double2* __restrict__ offpx, const mrx_table_dev* __restrict__ tables, int n_tables,
    const float* __restrict__ table_data, int table_floats, const float* __restrict__ az,
    const float* __restrict__ el, int Ta, const float* __restrict__ dxs, const float* __restrict__ dys,
    const int32_t* __restrict__ band, const float* __restrict__ mueller00, int D, double pwv0,
    float* loading,  // written by the sampler role, read by the writer role: no __restrict__, no const
    uint32_t* __restrict__ flags, int chunk, int nby, int block_rows, int n_blocks, int n_sampler_wgs,
    SynthPhases phases, double ta0, double inv_dta, const double* __restrict__ t, int T,
    const float* __restrict__ scale, const int32_t* __restrict__ rows, float* __restrict__ out, size_t ld, int vec_ok,
    int batches, int* ctl, int poll_limit, SynthCal cal) {
  extern __shared__ __align__(16) unsigned char synth_lds[];
  if ((int)blockIdx.x < n_sampler_wgs) {
    SynthHooks<kKrj> hooks;
    hooks.ctl = ctl;
    hooks.band = band;
    hooks.cal = cal;
    if (kKrj) {
      // the cell table behind the anchors, the band tables and the four turned steps (16-byte aligned)
      float4* cells = reinterpret_cast<float4*>(synth_lds) + 2 * chunk * n_layers + (kLdsTables ? (table_floats + 3) / 4 : 0) + kBlock;
      stage_cal_cells(cells, cal.axis, cal.values, cal.n_el, cal.n_bands);
      __syncthreads();
      hooks.cells = cells;
      hooks.el_first = cells[0].x;
      hooks.el_last = cal.axis[cal.n_el - 1];
      hooks.el_inv = cells[0].z;
    }
    // phases: the first blocks by ALL sampler workgroups (nothing else is resident yet: the chip is theirs), the next
    // ones by fewer and fewer of them -- those that leave make room for writers --, the rest by the first few
    int b0 = 0;
    for (int ph = 0; ph < phases.n; ++ph) {
      const int nw = phases.wgs[ph];
      if ((int)blockIdx.x >= nw) break;
      const int b1 = phases.end[ph];
      if (b1 > b0)
        mrx_px::px_sample_items<kLdsTables, 1, true, true>(
            fast, lpx, n_layers, offpx, tables, n_tables, table_data, table_floats, az, el, Ta, dxs, dys, band, mueller00,
            D, pwv0, nullptr, loading, flags, chunk, nby, block_rows, n_blocks, b0, b1, (int)blockIdx.x, nw,
            reinterpret_cast<float4*>(synth_lds), hooks);
      b0 = max(b0, b1);
    }
    synth_leave(ctl, n_blocks);
    return;
  }
  constexpr int kRows = kTileDet * kG;
  __shared__ int s_next;
  const int nsx = (T + kTileSamples - 1) / kTileSamples;
  const int rows_per_tile = kRows * batches;
  const int last_rows = D - (n_blocks - 1) * block_rows;
  const int tiles_full = nsx * ((block_rows + rows_per_tile - 1) / rows_per_tile);
  const int total = tiles_full * (n_blocks - 1) + nsx * ((last_rows + rows_per_tile - 1) / rows_per_tile);
  int have = -1;  // blocks up to this one are known to be sampled
  for (;;) {
    if (threadIdx.x == 0) s_next = __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();  // (also: the previous tile's readers of the LDS images are done)
    const int tile = s_next;
    if (tile >= total) break;
    const int blk = min(tile / tiles_full, n_blocks - 1);
    const int rem = tile - blk * tiles_full;
    const int by = rem / nsx, sx = rem - by * nsx;
    const int Db = blk == n_blocks - 1 ? last_rows : block_rows;
    if (blk > have) {  // workgroup-uniform
      if (threadIdx.x == 0) {
        const int want = nby * ((Db + mrx_px::kPxBlock - 1) / mrx_px::kPxBlock);
        int tries = 0;
        while (__hip_atomic_load(ctl + 32 + blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          if (++tries > poll_limit) {
            atomicOr(flags, MRX_FLAG_HANDOVER);
            break;
          }
          __builtin_amdgcn_s_sleep(32);
        }
      }
      __syncthreads();  // between the poll and EVERY load of the block's bytes, the polling wave's own too
      have = blk;
    }
    const size_t row0 = (size_t)blk * block_rows;
    fused_writer_tile<kHasScale, kMaxKnots, kG, true>(loading + (size_t)Ta * row0, (Db + 31) & ~31, Db, Ta, ta0, inv_dta, t, T,
                                                       kHasScale ? scale + row0 : nullptr, rows ? rows + row0 : nullptr,
                                                       rows ? out : out + row0 * ld, ld, vec_ok, batches, sx, by, synth_lds);
  }
  synth_leave(ctl, n_blocks);
}

}  // namespace

extern "C" {

static int atm_synthesize(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az, const float* d_el, int Ta,
                          const float* d_dx, const float* d_dy, const int32_t* d_band, const float* d_mueller00, int D,
                          double pwv0, float* d_coarse, int block_rows, int head_rows, uint32_t* d_flags, double ta0,
                          double dta, const double* d_t, int T, const float* d_scale, const int32_t* d_rows, float* d_out,
                          size_t ld_out, const SynthCal* krj) {
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, D >= 0 && Ta >= 0 && T >= 0, "negative size");
  if (D == 0 || T == 0) return MRX_OK;
  MRX_REQUIRE(ctx, plan != nullptr, "plan is null");
  MRX_REQUIRE(ctx, d_az && d_el && d_dx && d_dy && d_band && d_mueller00, "null input pointer");
  MRX_REQUIRE(ctx, d_coarse && d_flags && d_t && d_out, "null pointer");
  MRX_REQUIRE(ctx, plan->n_layers == 0 || Ta == plan->n_t, "Ta differs from the plan's n_t (length of the wind offsets)");
  MRX_REQUIRE(ctx, dta > 0.0, "coarse step must be positive");
  MRX_REQUIRE(ctx, ld_out >= (size_t)T, "ld_out smaller than T");
  if (Ta < 4)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "cubic interpolation needs at least 4 coarse samples (got %d), as scipy "
                    "interp1d(kind='cubic') does", Ta);
  // the one-launch form exists for what the default step runs: every layer on a verified-uniform axis, the
  // reference's default cell rule and pointing, linear band tables (otherwise: mrx_atm_sample + mrx_spline_upsample_fused)
  const bool literal = ctx->options[MRX_OPT_AXIS_LITERAL] != 0 || ctx->options[MRX_OPT_POINTING_CHAIN] != 0;
  if (!plan->all_pixel || literal || plan->any_cubic || plan->n_layers > mrx_px::kMaxAnchors)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "mrx_atm_synthesize: this plan or option set takes the two-call form");
  const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  if (block_rows <= 0) block_rows = D;
  block_rows = std::min(mrx_ceil_div(block_rows, kBlock) * kBlock, mrx_ceil_div(D, 32) * 32);  // whole groups of 256 lanes; rows of whole lines
  const int n_blocks = mrx_ceil_div(D, block_rows);
  MRX_REQUIRE(ctx, n_blocks <= kSynthMaxBlocks, "too many detector blocks");
  MRX_REQUIRE(ctx, (long long)Ta * block_rows * 4 < (1LL << 31), "a block's coarse array must stay below 2 GiB");
  MRX_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(d_coarse) & 127u) == 0, "d_coarse must be 128-byte aligned");
  // ---- writer role: mrx_spline_upsample_fused's choices ----
  const int vec_ok = (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0);
  const double knots_per_tile = (double)kTileSamples * (double)Ta / (double)T;
  const bool small = knots_per_tile + 6.0 <= 64.0;
  const int rows_per_batch = small ? 2 * kTileDet : kTileDet;
  int batches = ctx->options[MRX_OPT_UPSAMPLE_GROUPS];
  if (batches <= 0) batches = 1;
  while (batches > 1 && (long long)mrx_ceil_div(T, kTileSamples) * mrx_ceil_div(D, rows_per_batch * batches) < 4LL * 256 * 4)
    batches /= 2;
  const long long rows_per_tile = (long long)rows_per_batch * batches;
  const long long nsx = mrx_ceil_div(T, kTileSamples);
  const long long n_tiles = nsx * ((long long)(n_blocks - 1) * ((block_rows + rows_per_tile - 1) / rows_per_tile) +
                                   ((D - (long long)(n_blocks - 1) * block_rows) + rows_per_tile - 1) / rows_per_tile);
  MRX_REQUIRE(ctx, n_tiles <= 0x7fffffffLL - 65536, "too many tiles for one launch");
  const size_t lds_w = small ? FusedLds<64, 2>::kBytes : FusedLds<256, 1>::kBytes;
  // ---- sampler role: mrx_atm_sample's choices for a resident grid ----
  int chunk = ctx->options[MRX_OPT_SAMPLE_CHUNK];
  if (chunk <= 0) {
    chunk = mrx_px::kMaxChunk;
    const long long want = 24LL * n_cu;
    while (chunk > 1 && (long long)mrx_ceil_div(D, kBlock) * mrx_ceil_div(Ta, chunk) < want) chunk /= 2;
  }
  chunk = chunk < 1 ? 1 : chunk > mrx_px::kMaxChunk ? mrx_px::kMaxChunk : chunk;
  while (chunk > 1 && chunk * plan->n_layers > mrx_px::kMaxAnchors) chunk /= 2;
  // the launch has ONE dynamic LDS size and a CU's LDS is what bounds its writers: the sampler's anchors stay under
  // the writer's images (16 layers at 64 steps a work item took 37 KB -- and a writer's place on every CU)
  const size_t lds_turn = sizeof(float) * 4 * kBlock;  // four steps of every lane (px_sample_items<..., kWriteThrough>)
  while (chunk > 8 && 2 * sizeof(float4) * (size_t)chunk * plan->n_layers + lds_turn > lds_w) chunk /= 2;
  const int nby = mrx_ceil_div(Ta, chunk);
  const long long n_items = (long long)nby * ((long long)(n_blocks - 1) * mrx_ceil_div(block_rows, kBlock) +
                                              mrx_ceil_div(D - (n_blocks - 1) * block_rows, kBlock));
  MRX_REQUIRE(ctx, n_items <= 0x7fffffffLL, "too many work items for one launch");
  int per_cu = ctx->options[MRX_OPT_SAMPLE_WGS_PER_CU];
  if (per_cu <= 0 || per_cu >= 8) per_cu = 2;
  long long wgs_s = std::min(n_items, (long long)per_cu * n_cu);
  if (wgs_s >= 8) wgs_s &= ~7LL;
  // the head start: the first blocks by a grid that fills the chip (MRX_WRITER_WAVES workgroups per CU)
  const int head_blocks = std::max(0, std::min(mrx_ceil_div(std::max(head_rows, 0), block_rows), n_blocks));
  const long long wgs_full = std::max(wgs_s, std::min(n_items, (long long)MRX_WRITER_WAVES * n_cu) & ~7LL);
  SynthPhases phases = {};
  // (a staircase -- the head's blocks in shares to 5, 4, 3 workgroups per CU, writers entering as each step leaves --
  //  measured no better than one step: 1.98-2.05 against 1.93-1.98 ms)
  if (head_blocks > 0) {
    phases.n = 2;
    phases.end[0] = head_blocks; phases.wgs[0] = (int)wgs_full;
    phases.end[1] = n_blocks;    phases.wgs[1] = (int)wgs_s;
  } else {
    phases.n = 1;
    phases.end[0] = n_blocks; phases.wgs[0] = (int)wgs_s;
  }
  const long long wgs_head = phases.wgs[0];
  const size_t lds_anchor = 2 * sizeof(float4) * (size_t)chunk * plan->n_layers;
  const size_t lds_tables = sizeof(float) * (size_t)((plan->table_floats + 3) / 4 * 4);
  // band tables in LDS only where they fit under the writer's images too
  const size_t lds_cal = krj ? sizeof(float4) * (size_t)(krj->n_el - 1) * krj->n_bands : 0;  // the K_RJ cell table
  const bool lds_tab = plan->table_floats <= mrx_px::kMaxLdsTableFloats && lds_anchor + lds_tables + lds_turn + lds_cal <= lds_w;
  if (lds_anchor + (lds_tab ? lds_tables : 0) + lds_turn + lds_cal > lds_w)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "mrx_atm_synthesize: the calibration table does not fit beside the sampler's LDS: use the two calls");
  const size_t lds = lds_w;
  // writers: as many as fit a CU once the samplers have left (the surplus is dispatched as those exit)
  const long long per_cu_w = std::max<long long>(1, std::min<long long>(MRX_WRITER_WAVES, (long long)(ctx->lds_per_cu > 0 ? ctx->lds_per_cu : 160 * 1024) / (long long)(lds + 1552)));
  const long long wgs_w = std::min(n_tiles, per_cu_w * n_cu);
  int* ctl = nullptr;
  {
    const int rc = mrx_synth_ctl(ctx, &ctl);
    if (rc != MRX_OK) return rc;
  }
  const dim3 grid((unsigned)(wgs_head + wgs_w));
  const int poll_limit = 1 << 22;  // x >= 0.5 us a try: seconds
  const SynthCal cal = krj ? *krj : SynthCal{};
#define MRX_LAUNCH_SYNTH(L, S, K, G, J)                                                                           \
  do {                                                                                                            \
    MRX_LDS_CAP(ctx, (atm_tod_kernel<L, S, K, G, J>), lds);                                                       \
    hipLaunchKernelGGL((atm_tod_kernel<L, S, K, G, J>), grid, dim3(kBlock), lds, ctx->stream, plan->d_fast,       \
                       plan->d_px, plan->n_layers, plan->d_offpx, plan->d_tables, plan->n_tables,                 \
                       plan->d_table_data, plan->table_floats, d_az, d_el, Ta, d_dx, d_dy, d_band, d_mueller00,   \
                       D, pwv0, d_coarse, d_flags, chunk, nby, block_rows, n_blocks, (int)wgs_head, phases, ta0, 1.0 / dta,           \
                       d_t, T, d_scale, d_rows, d_out, ld_out, vec_ok, batches, ctl, poll_limit, cal);           \
  } while (0)
#define MRX_LAUNCH_SYNTH_J(L, S, K, G) do { if (krj) MRX_LAUNCH_SYNTH(L, S, K, G, true); else MRX_LAUNCH_SYNTH(L, S, K, G, false); } while (0)
#define MRX_LAUNCH_SYNTH_S(L, S) do { if (small) MRX_LAUNCH_SYNTH_J(L, S, 64, 2); else MRX_LAUNCH_SYNTH_J(L, S, 256, 1); } while (0)
  if (lds_tab) {
    if (d_scale) MRX_LAUNCH_SYNTH_S(true, true); else MRX_LAUNCH_SYNTH_S(true, false);
  } else {
    if (d_scale) MRX_LAUNCH_SYNTH_S(false, true); else MRX_LAUNCH_SYNTH_S(false, false);
  }
#undef MRX_LAUNCH_SYNTH_S
#undef MRX_LAUNCH_SYNTH_J
#undef MRX_LAUNCH_SYNTH
  MRX_CHECK_LAUNCH(ctx);
  return MRX_OK;
}

int mrx_atm_synthesize(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az, const float* d_el, int Ta,
                       const float* d_dx, const float* d_dy, const int32_t* d_band, const float* d_mueller00, int D,
                       double pwv0, float* d_coarse, int block_rows, int head_rows, uint32_t* d_flags, double ta0,
                       double dta, const double* d_t, int T, const float* d_scale, const int32_t* d_rows, float* d_out,
                       size_t ld_out) {
  MRX_ENTER(ctx);
  return atm_synthesize(ctx, plan, d_az, d_el, Ta, d_dx, d_dy, d_band, d_mueller00, D, pwv0, d_coarse, block_rows, head_rows,
                        d_flags, ta0, dta, d_t, T, d_scale, d_rows, d_out, ld_out, nullptr);
}

int mrx_atm_synthesize_krj(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az, const float* d_el, int Ta,
                           const float* d_dx, const float* d_dy, const int32_t* d_band, const float* d_mueller00, int D,
                           double pwv0, float* d_coarse, int block_rows, int head_rows, uint32_t* d_flags, double ta0,
                           double dta, const double* d_t, int T, const float* d_scale, const int32_t* d_rows, float* d_out,
                           size_t ld_out, const float* d_cal_dx, const float* d_cal_dy, const float* d_cal_axis_el,
                           const float* d_cal_values, int n_el, int n_bands, float* d_tail_pw, int tail_knots, size_t ld_tail) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, d_cal_dx && d_cal_dy && d_cal_axis_el && d_cal_values, "null calibration pointer");
  MRX_REQUIRE(ctx, n_el >= 2 && n_bands >= 1, "calibration tables need 2 <= n_el, 1 <= n_bands");
  MRX_REQUIRE(ctx, tail_knots >= 0 && tail_knots <= Ta && (!d_tail_pw || ld_tail >= (size_t)D), "bad tail window");
  SynthCal cal{d_cal_dx, d_cal_dy, d_cal_axis_el, d_cal_values, n_el, n_bands, tail_knots > 0 ? d_tail_pw : nullptr, Ta - tail_knots, ld_tail};
  return atm_synthesize(ctx, plan, d_az, d_el, Ta, d_dx, d_dy, d_band, d_mueller00, D, pwv0, d_coarse, block_rows, head_rows,
                        d_flags, ta0, dta, d_t, T, d_scale, d_rows, d_out, ld_out, &cal);
}

}  // extern "C"
