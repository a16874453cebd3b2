// The pixel-coordinate sampler (atm_sample_px_kernel's body) and the device-side plan it reads, shared by the
// sampling TU (mrx_sample.hip: the kernel on its own) and the synthesis TU (mrx_synth.hip: the same body as the
// sampler role of the one-launch atmosphere -> TOD kernel).  Arithmetic follows the reference's rounding points
// (see mrx_sample.hip's header): everything here is compiled WITHOUT floating-point contraction, whatever the
// including TU's flags say -- a * b + c rounds twice, like numpy / XLA; fused multiply-adds are written out.
#pragma once

#include "mrx_internal.h"

#pragma clang fp contract(off)

// Device-side layer descriptor (plan-private).
struct mrx_layer_dev {
  const float* values;
  const float* axis_e;
  const float* axis_c;
  double h, r00, r10, r01, r11;
  double hr00, hr10, hr01, hr11;  // h * r: the projection and the layer height in one product
  double e0, de, c0, dc;          // node(i) = float32(e0 + i*de) when uniform_*
  double pe_x, pe_y, pc_x, pc_y;  // pixel coordinates: fe = px*pe_x + py*pe_y + offpx.x, fc likewise
  float e_first, e_inv, e_last;   // axis_e[0], 1/(axis_e[1]-axis_e[0]), axis_e[n-1]
  float c_first, c_inv, c_last;
  float pwv_rms;
  int n_e, n_c;
  int uniform_e, uniform_c;
};

// What the pixel-coordinate path needs of a layer, in one 64-byte line: the whole record is
// one scalar load per layer and wave (the general descriptor above costs a dozen).
struct alignas(64) mrx_layer_fast {
  const float* values;
  double pe_x, pe_y, pc_x, pc_y;
  int n_e, n_c;
  float pwv_rms;
  int pixel;  // both axes verified uniform
};
static_assert(sizeof(mrx_layer_fast) == 64, "one cache line per layer");

// The float32 record of the pixel-coordinate kernel (atm_sample_px_kernel): one scalar load per layer and wave.
struct alignas(64) mrx_layer_px {
  const float* values;   // the screen: the lower corners of a cell
  const float* values1;  // the screen from its second row on: the upper corners, at the same byte offset
  float pe_x, pe_y, pc_x, pc_y;
  int nc4;               // 4 n_c: a row in bytes
  uint32_t bytes;        // 4 n_e n_c (< 4 GiB, plan_finish_layers): the range the hardware checks the gathers against
  uint32_t bytes1;       // bytes - nc4
  float pwv_rms;
  float half_e, half_c;  // (n_e - 1) / 2, (n_c - 1) / 2: a position is on the grid while |position - middle| <= half
  int pad_[2];
};
static_assert(sizeof(mrx_layer_px) == 64, "one cache line per layer");

// Device-side band table descriptor: offsets (in floats) into the packed table
// buffer, which is [values 2*np*ne][axis_pwv np][axis_el ne] per band.
struct mrx_table_dev {
  int off_values, off_pwv, off_el;
  int n_pwv, n_el;
  float w_t;
  int t_oob;
  float p_first, p_inv, e_first, e_inv;
  const double* cubic;  // bicubic cells (mrx_band_table::d_cubic) or null
};

struct mrx_atm_plan {
  mrx_layer_dev* d_layers = nullptr;
  mrx_layer_fast* d_fast = nullptr;
  mrx_layer_px* d_px = nullptr;  // float32 records of the pixel-coordinate kernel
  double2* d_off = nullptr;  // [n_t][n_layers] (off_e, off_c)
  double2* d_offpx = nullptr;  // [n_t][n_layers] ((off_e - e0)/de, (off_c - c0)/dc): the same in pixels
  mrx_table_dev* d_tables = nullptr;
  float* d_table_data = nullptr;
  int n_layers = 0, n_tables = 0, n_t = 0;
  int table_floats = 0;
  bool all_pixel = false;  // every layer passed the uniform check: atm_sample_px_kernel applies
  unsigned long long screen_bytes = 0;  // of all layers' screens together
  bool any_cubic = false;  // a band table carries the bicubic cells of interpolation_method="cubic"
};

namespace mrx_px {

constexpr int kPxBlock = 256;
constexpr int kMaxLdsTableFloats = 12288;  // 48 KiB of band tables in LDS
constexpr int kMaxChunk = 64;              // time steps per workgroup, at most

// float32(pi/2): jax folds the weak-typed python float pi/2 to float32
// (coords/transforms.py:22) and numpy clips float32 elevations to it
// (sim/atmosphere.py:60).
constexpr float kHalfPiF = 1.57079637050628662109375f;

struct Cell {
  int i;
  float w;  // normalised distance to the lower node
  bool oob;
};

// v_cvt_flr_i32_f32: floor and convert in one instruction; saturates, NaN -> 0 (no undefined
// conversion whatever the input)
__device__ __forceinline__ int cvt_flr_i32(float x) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// v_med3_i32: clamp x into [lo, hi] (lo <= hi) in one instruction
__device__ __forceinline__ int med3_i32(int x, int lo, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "s"(hi));
  return r;
}
// v_mad_i32_i24: a * b + c with a, b signed 24-bit (a cell offset of a few pixels times a row pitch below 2^24)
__device__ __forceinline__ int mad_i32_i24(int a, int b, int c) {
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}
typedef float pair4 __attribute__((ext_vector_type(2), aligned(4)));
typedef __attribute__((address_space(1))) const pair4 gpair;

// ---- the pixel-coordinate kernel: float32 in the layer loop ----------------------------------
// For plans whose layers all sit on verified-uniform axes (every generated screen does).  The
// position of a line of sight on layer l at step t, in pixels, is affine in the unit-height
// ground projection (px, py):   f = px pe_x + py pe_y + offpx(t, l).
// At ~2000 pixels from the grid origin a float32 evaluation of that sum would be good to 2e-4
// pixel (the reference's own rounding); round 2 therefore ran it in float64 -- four v_fma_f64,
// four conversions and two v_add_f64 per layer-sample, 103 VALU instructions all told.  Here the
// float64 part is paid once per (work item, step, layer) instead of once per lane:
//   * ANCHOR.  For every step of the item and every layer, the pixel position of the boresight's
//     own line of sight (pc_x, pc_y: float32 numbers, any point near the detectors would do) in
//     float64, split into an integer cell and a float32 fraction in [0, 1): 16 bytes in LDS,
//     computed by one thread each in the item's prologue;
//   * DELTA.  A lane adds its float32 offset from the anchor, (px - pc_x) pe_x + (py - pc_y) pe_y:
//     |delta| is the focal plane's footprint on the layer in pixels (<= ~40 at the top layer of
//     atlast_10k), so its float32 rounding is <= 4e-6 pixel -- 50x below the reference's own
//     coordinate rounding, and (px - pc_x), (py - pc_y) are shared by all layers.
// Cell = anchor cell + floor(fraction + delta); weight = what is left.  The 2x2 blend is two
// lerps along c and one along e (6 operations; jax's four weighted corners sum to the same value
// to rounding), the layer's term is accumulated in float32 on the FLUCTUATION only
// (sum_l rms_l y_l ~ 3 % of pwv0) and added to the float64 pwv0 once per step.
constexpr int kMaxAnchors = 1024;  // (step, layer) pairs of one work item: 32 KiB of LDS at most

#ifndef MRX_PX_STAGES
#define MRX_PX_STAGES 3
#endif
#ifndef MRX_PX_WAVES
#define MRX_PX_WAVES 6
#endif
constexpr int kPxStages = MRX_PX_STAGES;  // layers in the software pipeline of the resident (kPipe) instance
constexpr int kPxWaves = MRX_PX_WAVES;    // its register budget: 512 / kPxWaves

// a / b to within an ulp (correctly rounded but for rare ties) in four instructions instead of the ten of the
// IEEE sequence: v_rcp_f32, the quotient, its residual, one correction.  b is a table step or sin(elevation):
// normal numbers; b = 0, infinite or NaN gives NaN or infinity, which the callers flag.
__device__ __forceinline__ float div_near(float a, float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  const float q = a * r;
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}

// jax's _find_indices on one axis of a band table whose first step predicts the cell (the `am` axes are uniform
// but for the last elevation node, which the clamp absorbs): the guess, its two nodes, and -- only when some lane
// of the wave sits within rounding of a node or the axis is not uniform -- the search of find_cell.
template <typename NodeFn>
__device__ __forceinline__ Cell guess_cell(NodeFn node, int n, float x, float first, float inv, float last) {
  const float f = fminf(fmaxf((x - first) * inv, -1.0f), 2.0e9f);
  Cell c;
  c.i = min(max((int)f, 0), n - 2);
  float lo = node(c.i), hi = node(c.i + 1);
  const bool miss = (c.i < n - 2 && hi < x) || (c.i > 0 && lo >= x);
  if (__builtin_amdgcn_ballot_w64(miss) != 0) {
    while (c.i < n - 2 && hi < x) { ++c.i; lo = hi; hi = node(c.i + 1); }
    while (c.i > 0 && lo >= x) { --c.i; hi = lo; lo = node(c.i); }
  }
  c.w = div_near(x - lo, hi - lo);
  c.oob = !(x >= first && x <= last);  // also true for NaN
  return c;
}

// band_loading's linear branch for the pixel kernel: the same 8-term float32 sum in the reference's order
// (band/band.py:283-286), cells from guess_cell, weights from div_near.
__device__ __forceinline__ float band_loading_px(const mrx_table_dev& tb, const float* __restrict__ tdata, float xp,
                                                 float theta, float m00, bool& table_oob) {
  const float* __restrict__ ax_p = tdata + tb.off_pwv;
  const float* __restrict__ ax_e = tdata + tb.off_el;
  const float* __restrict__ tv = tdata + tb.off_values;
  const int slab = tb.n_pwv * tb.n_el;
  const float xel = fminf(theta, kHalfPiF);  // .clip(max=pi/2), sim/atmosphere.py:60
  const Cell cp_ = guess_cell([=](int i) { return ax_p[i]; }, tb.n_pwv, xp, tb.p_first, tb.p_inv, ax_p[tb.n_pwv - 1]);
  const Cell cl = guess_cell([=](int i) { return ax_e[i]; }, tb.n_el, xel, tb.e_first, tb.e_inv, ax_e[tb.n_el - 1]);
  const float* q = tv + cp_.i * tb.n_el + cl.i;
  float val = 0.0f;
#pragma unroll
  for (int ia = 0; ia < 2; ++ia) {
    const float w1 = 1.0f * (ia ? tb.w_t : 1.0f - tb.w_t);
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      const float w2 = w1 * (ib ? cp_.w : 1.0f - cp_.w);
#pragma unroll
      for (int ic = 0; ic < 2; ++ic) {
        const float w3 = w2 * (ic ? cl.w : 1.0f - cl.w);
        val = val + q[ia * slab + ib * tb.n_el + ic] * w3;
      }
    }
  }
  table_oob = cp_.oob || cl.oob || tb.t_oob;
  return table_oob ? __builtin_nanf("") : m00 * val;
}

// The work of one sampler workgroup, `wg` of `n_wgs`: the body of atm_sample_px_kernel (which calls it with its block
// index and grid size, one detector block, nothing to signal) and the sampler role of the one-launch synthesis
// (atm_tod_kernel, mrx_synth.hip).  The detectors come in `n_blocks` blocks of `block_rows` rows (the last one
// shorter); block b's loading is its own time-major array loading + Ta * (b * block_rows) of [Ta][rows of b], and
// `hooks.item(b, d)` is called by every thread at the start of a work item of block b with its detector's row in the
// whole shard, `hooks.value(loading, (cos, sin, ..) of the step's boresight, step, row, real)` gives what is stored for
// a sample, and `hooks.done(b)` is called by every thread after each finished work item (its stores issued, not yet
// waited for).  PxNoHooks: the loading as it is, nobody to tell.
// kWriteThrough (the synthesis role): the loading leaves the CU as 16-byte write-through (sc1) stores -- four steps of
// a wave are turned in LDS so that a lane holds four neighbouring detectors of one step -- into rows of pitch
// round_up(rows of b, 32) floats, every 128-byte line written whole by one store instruction (lanes past the last
// row write padding): the form MI355X_MICROARCH.md gives for bytes another XCD reads in the same launch without a
// release fence (which would write back the whole L2 of the XCD, per work item, under a streaming writer: measured
// 0.18 ms of a 1.9-ms step).  Otherwise plain 4-byte stores at pitch = rows.
// Parameters are passed one by one, not in a struct: the callers hand their `const __restrict__` kernel arguments
// through, which is what lets the compiler keep the wave-uniform loads (layer records, anchors' inputs) on the scalar unit.
struct PxNoHooks {
  __device__ __forceinline__ void item(int, int) {}
  __device__ __forceinline__ float value(float v, const float4&, int, int, bool) const { return v; }
  __device__ __forceinline__ void done(int) {}
};

template <bool kLdsTables, int kT, bool kPipe, bool kWriteThrough, typename Hooks>
__device__ __forceinline__ void px_sample_items(
    const mrx_layer_fast* __restrict__ fast, const mrx_layer_px* __restrict__ lpx, int n_layers,
    const double2* __restrict__ offpx, const mrx_table_dev* __restrict__ tables, int n_tables,
    const float* __restrict__ table_data, int table_floats, const float* __restrict__ az,
    const float* __restrict__ el, int Ta, const float* __restrict__ dxs_all, const float* __restrict__ dys_all,
    const int32_t* __restrict__ band_all, const float* __restrict__ mueller00_all, int D_all, double pwv0,
    double* __restrict__ pwv_out_all, float* __restrict__ loading_all, uint32_t* __restrict__ flags, int chunk,
    int nby, int block_rows, int n_blocks, int blk_begin, int blk_end, int wg, int n_wgs, float4* lds_px, Hooks& hooks) {
  // [chunk * n_layers anchors of 32 bytes: (fraction e, fraction c, middle e, middle c), byte offset of the anchor's
  //  cell, padding][band tables]
  __shared__ float4 bore[kMaxChunk];   // per step: cos/sin of (el - pi/2), cos/sin of az
  __shared__ float2 borec[kMaxChunk];  // per step: unit-height projection of the boresight itself
  float4* anchor = lds_px;
  float* lds_tables = reinterpret_cast<float*>(lds_px + 2 * chunk * n_layers);
  // kWriteThrough: [4 steps][64 lanes] floats per wave, behind the tables (16-byte aligned: table_floats rounded up)
  float* turn = lds_tables + (kLdsTables ? (table_floats + 3) / 4 * 4 : 0) + (threadIdx.x / 64) * 256;
  static_assert(!kWriteThrough || kT == 1 || kT == 2 || kT == 4, "the write-through form turns four steps: kT must divide 4");
  if (kLdsTables)
    for (int i = threadIdx.x; i < table_floats; i += kPxBlock) lds_tables[i] = table_data[i];
  const float* __restrict__ tdata = kLdsTables ? lds_tables : table_data;
  uint32_t myflags = 0u;

  // Work items = (time chunk, block of 256 detectors), dealt so that the workgroups of one XCD (blockIdx mod 8 under
  // the round-robin placement: a matter of speed only) share their time chunks: XCD x walks the chunks x, x + 8, ...,
  // its workgroups taking the detector blocks of a chunk side by side.  The lines of sight of one chunk meet a few
  // hundred KB of the layer stack; with the items dealt round-robin every XCD's 4 MB L2 saw the footprints of all
  // ~80 chunks in flight at once -- 16 MB for 16 layers of 4096^2, 40 % of its L2 reads missing (profiles/r04_50k_*).
  // Row blocks follow one another in that order: an XCD's items are those of block 0, then block 1, ..., and its
  // workgroups take them in turn across the block boundaries.
  const int n_xcd = (n_wgs & 7) == 0 ? 8 : 1;  // (a grid that is no multiple of 8: one group, the plain order)
  const int xcd = wg % n_xcd, per_xcd = n_wgs / n_xcd;
  const int nbx_full = (block_rows + kPxBlock - 1) / kPxBlock;
  const int last_rows = D_all - (n_blocks - 1) * block_rows;
  const int per_block = ((nby - xcd + n_xcd - 1) / n_xcd) * nbx_full;  // this XCD's items of a full block (> 0: see the break)
  for (int q = wg / n_xcd;; q += per_xcd) {
    if (xcd >= nby) break;  // (fewer time chunks than XCDs: this one has none)
    int rel = q / per_block;
    if (rel >= blk_end - blk_begin) {
      if (blk_end != n_blocks) break;  // (a range that stops short of the last block ends on a full one)
      rel = blk_end - blk_begin - 1;   // the last block of all may be shorter: its own count of items, below
    }
    const int blk = blk_begin + rel;
    const int pair = q - rel * per_block;
    const int D = blk == n_blocks - 1 ? last_rows : block_rows;
    const int nbx = (D + kPxBlock - 1) / kPxBlock;
    const int pitch = kWriteThrough ? (D + 31) & ~31 : D;
    const int by = (pair / nbx) * n_xcd + xcd;
    if (by >= nby) break;  // (only the last block ends the walk)
    const int bx = pair % nbx;
    const size_t row0 = (size_t)blk * block_rows;
    const float* __restrict__ dxs = dxs_all + row0;
    const float* __restrict__ dys = dys_all + row0;
    const int32_t* __restrict__ band = band_all + row0;
    const float* __restrict__ mueller00 = mueller00_all + row0;
    float* __restrict__ loading = loading_all + (size_t)Ta * row0;
    double* __restrict__ pwv_out = pwv_out_all ? pwv_out_all + (size_t)Ta * row0 : nullptr;
    // (kWriteThrough: the block's array as a raw buffer, 4 Ta pitch bytes < 4 GiB -- the launcher checks)
    const __amdgpu_buffer_rsrc_t wt_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)loading, 0, kWriteThrough ? Ta * pitch * 4 : 0, 0x00020000);
    const int t_first = by * chunk;
    uint32_t iflags = 0u;
    __syncthreads();  // the previous item's readers of bore[] and anchor[] are done
    if ((int)threadIdx.x < chunk) {
      const int t = min(t_first + (int)threadIdx.x, Ta - 1);
      const float a = el[t] - kHalfPiF;  // transforms.py:22
      const float z = az[t];
      const float ca = cosf(a), sa = sinf(a), cz = cosf(z), sz = sinf(z);
      bore[threadIdx.x] = make_float4(ca, sa, cz, sz);
      // the detector at the focal-plane centre: re = -sin a, im = cos a
      borec[threadIdx.x] = make_float2((-sa * cz) / ca, (-sa * sz) / ca);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < chunk * n_layers; k += kPxBlock) {
      const int it = k / n_layers, l = k - it * n_layers;
      const int t = min(t_first + it, Ta - 1);
      const mrx_layer_fast lf = fast[l];
      const double2 o = offpx[(size_t)t * n_layers + l];
      const float2 pc = borec[it];
      const double Fe = fma((double)pc.x, lf.pe_x, fma((double)pc.y, lf.pe_y, o.x));
      const double Fc = fma((double)pc.x, lf.pc_x, fma((double)pc.y, lf.pc_y, o.y));
      // (a NaN position gives a finite cell and a NaN fraction, which every lane then reports as off the screen;
      // an anchor far outside wraps its byte offset: any offset is safe, the gathers are range-checked)
      const double ce = floor(fmin(fmax(Fe, -1.0e9), 1.0e9)), cc = floor(fmin(fmax(Fc, -1.0e9), 1.0e9));
      anchor[2 * k] = make_float4((float)(Fe - ce), (float)(Fc - cc), (float)(0.5 * (double)(lf.n_e - 1) - ce),
                                  (float)(0.5 * (double)(lf.n_c - 1) - cc));
      reinterpret_cast<int*>(anchor + 2 * k + 1)[0] = (int)(uint32_t)(((long long)ce * lf.n_c + (long long)cc) * 4ll);
    }

    const int d = bx * kPxBlock + threadIdx.x;
    const bool live = d < D;
    const int dd = live ? d : D - 1;  // keep addresses valid; stores are masked
    hooks.item(blk, (int)row0 + dd);
    // ---- per-detector constants (coords/transforms.py:14-23), float32 ------
    const float dx = dxs[dd], dy = dys[dd];
    const float r = sqrtf(dx * dx + dy * dy);
    const float p = atan2f(-dx, -dy);
    const float sr = sinf(r), cr = cosf(r);
    const float A = sr * cosf(p);  // sin(r) cos(p): real part before the tilt
    const float Y = sr * sinf(p);  // sin(r) sin(p)
    const int b = band[dd];
    const float m00 = mueller00[dd];
    if (live && (b < 0 || b >= n_tables)) iflags |= MRX_FLAG_NAN;
    const mrx_table_dev tb = tables[min(max(b, 0), n_tables - 1)];
    __syncthreads();  // anchors are in place

    for (int it = 0; it < chunk && t_first + it < Ta; it += kT) {
      float theta[kT], dpx[kT], dpy[kT], fl[kT];
      bool outside[kT];  // some layer's position left its grid (the compiler keeps it as a lane mask in scalar registers)
#pragma unroll
      for (int tt = 0; tt < kT; ++tt) {
        // transforms.py:20-28 and the unit-height ground projection (see atm_sample_kernel)
        const float4 bt = bore[it + tt];
        const float2 pc = borec[it + tt];
        const float re = A * bt.x - cr * bt.y;
        const float im = A * bt.y + cr * bt.x;
        theta[tt] = asinf(im);
        const float inv_im = div_near(1.0f, im);
        dpx[tt] = (re * bt.z - Y * bt.w) * inv_im - pc.x;
        dpy[tt] = (Y * bt.z + re * bt.w) * inv_im - pc.y;
        fl[tt] = 0.0f;
        outside[tt] = false;
      }
      // ---- layer stack (atmosphere/atmosphere.py:317-373) ---------------------
      // Per layer and sample: the position relative to the anchor's cell, f = fraction + delta (two fused
      // multiply-adds per axis); cell = floor(f) and weight = fract(f) (one instruction each); on the grid while
      // |f - middle| <= half the grid (a subtraction and a comparison per axis; exact but for the float32 rounding
      // of that difference, 1e-4 pixel at the rim of a 4096-node axis -- the reference's own float32 coordinate is
      // coarser); the byte offset of the cell from the anchor's (two integer multiply-adds); the two rows' corner
      // pairs as two 8-byte BUFFER loads, which the hardware checks against the screen's size -- no clamp, and
      // whatever a degenerate pointing produces reads zeros instead of faulting; two lerps along c and one along e.
      // 21 vector instructions against the 29 of round 3 (which clamped the cell per axis, rebuilt the weight from
      // the clamped cell and tracked the smallest / largest weight).
      // Software-pipelined by hand in the resident (kPipe) instance: the gathers of the next layers are issued
      // before layer l is blended, so that a wave has 2 (kPxStages - 1) kT loads in flight while it computes
      // (at the 3 waves per SIMD this kernel gets beside the TOD writer the compiler's own order -- issue a layer's
      // two loads, wait -- exposed the whole L2 latency once per layer).  sched_barrier pins the order.
      const float4* an = anchor + 2 * it * n_layers;
      struct Stage {  // one layer's gathers in flight, for the thread's kT steps
        pair4 r0[kT], r1[kT];
        float we[kT], wc[kT];
        float rms;
      };
      auto issue = [&](int l, Stage& g) {
        const mrx_layer_px lp = lpx[l];  // wave-uniform: one scalar load
        const int nc4 = lp.nc4;
        // two raw buffers: the screen, and the screen from its second row on -- the upper corners of a cell take the
        // same offset in the second, and each load is checked against its own range
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)lp.values, 0, (int)lp.bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)lp.values1, 0, (int)lp.bytes1, 0x00020000);
        g.rms = lp.pwv_rms;
#pragma unroll
        for (int tt = 0; tt < kT; ++tt) {
          const float4 a4 = an[2 * (tt * n_layers + l)];
          const int a0 = reinterpret_cast<const int*>(an + 2 * (tt * n_layers + l) + 1)[0];
          const float fe = __builtin_fmaf(dpx[tt], lp.pe_x, __builtin_fmaf(dpy[tt], lp.pe_y, a4.x));
          const float fc = __builtin_fmaf(dpx[tt], lp.pc_x, __builtin_fmaf(dpy[tt], lp.pc_y, a4.y));
          g.we[tt] = __builtin_amdgcn_fractf(fe);
          g.wc[tt] = __builtin_amdgcn_fractf(fc);
          outside[tt] |= !(__builtin_fabsf(fe - a4.z) <= lp.half_e) || !(__builtin_fabsf(fc - a4.w) <= lp.half_c);
          const int boff = mad_i32_i24(cvt_flr_i32(fe), nc4, a0) + (cvt_flr_i32(fc) << 2);
#ifdef MRX_PX_WHATIF_NOLOAD  // (timing what-if, scripts/exp/synth_timeline.sh: wrong values, the same arithmetic, no gathers)
          g.r0[tt] = pair4{__int_as_float(boff | 0x3f000000), 0.5f};
          g.r1[tt] = pair4{0.25f, __int_as_float(boff | 0x3f000000)};
#else
          g.r0[tt] = __builtin_bit_cast(pair4, __builtin_amdgcn_raw_buffer_load_b64(rs0, boff, 0, 0));
          g.r1[tt] = __builtin_bit_cast(pair4, __builtin_amdgcn_raw_buffer_load_b64(rs1, boff, 0, 0));
#endif
        }
      };
      auto blend = [&](const Stage& g) {
#pragma unroll
        for (int tt = 0; tt < kT; ++tt) {
          const float y0 = __builtin_fmaf(g.wc[tt], g.r0[tt].y - g.r0[tt].x, g.r0[tt].x);
          const float y1 = __builtin_fmaf(g.wc[tt], g.r1[tt].y - g.r1[tt].x, g.r1[tt].x);
          fl[tt] = __builtin_fmaf(g.rms, __builtin_fmaf(g.we[tt], y1 - y0, y0), fl[tt]);
        }
      };
      Stage ga;
      if (kPipe) {
        // kPxStages stages in a ring: kPxStages - 1 layers' gathers stay in flight while one is blended (blends in
        // layer order, so the float32 sum is the plain loop's bit for bit); fully unrolled, the ring lives in registers
        Stage g[kPxStages];
#pragma unroll
        for (int k = 0; k < kPxStages - 1; ++k)
          if (k < n_layers) issue(k, g[k]);
        for (int l = 0; l < n_layers; l += kPxStages) {
#pragma unroll
          for (int k = 0; k < kPxStages; ++k) {
            if (l + k + kPxStages - 1 < n_layers) issue(l + k + kPxStages - 1, g[(k + kPxStages - 1) % kPxStages]);
            __builtin_amdgcn_sched_barrier(0);
            if (l + k < n_layers) blend(g[k]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {  // alone on the chip at 8 waves per SIMD the occupancy hides the latency: the plain loop is 12 % faster there
        for (int l = 0; l < n_layers; ++l) {
          issue(l, ga);
          blend(ga);
        }
      }
      // ---- band emission (band/band.py:264-300) and Mueller weight -----------
#pragma unroll
      for (int tt = 0; tt < kT; ++tt) {
        if (kT > 1 && it + tt >= chunk) break;  // (a chunk that is no multiple of kT: the steps past it are another item's)
        const int t = t_first + it + tt;
        // a line of sight off a screen is jax's NaN fill (atmosphere.py:359-369); a NaN position makes the sum NaN
        const bool off = outside[tt] || fl[tt] != fl[tt];
        const double pwv = off ? (double)__builtin_nanf("") : pwv0 + (double)fl[tt];
        bool table_oob;
        const float out = band_loading_px(tb, tdata, (float)pwv, theta[tt], m00, table_oob);
        if (t < Ta) iflags |= (off ? MRX_FLAG_SCREEN_OOB : 0u) | (table_oob ? MRX_FLAG_TABLE_OOB : 0u) | (out != out ? MRX_FLAG_NAN : 0u);
        // what is stored: the loading itself, or what the caller's hook makes of it (K_RJ on the coarse grid)
        const float kept = hooks.value(out, bore[it + tt], t, (int)row0 + dd, live && t < Ta);
        if (kWriteThrough) {
          if (pwv_out && live && t < Ta) pwv_out[(size_t)t * D + d] = pwv;  // (plain stores: nobody reads it in this launch)
          const int u = (it + tt) & 3, lane = threadIdx.x & 63;
          turn[u * 64 + lane] = kept;
          __builtin_amdgcn_wave_barrier();  // (the read below takes other lanes' values: keep it behind the write)
          if (u == 3 || it + tt == chunk - 1 || t >= Ta - 1) {
            // lane 4k + j takes step j of the group, detectors 4k .. 4k+3 (same wave: LDS keeps the order)
            const int j = lane & 3, col = bx * kPxBlock + (int)(threadIdx.x & ~63u) + (lane & ~3);
            const float4 v = *reinterpret_cast<const float4*>(turn + j * 64 + (lane & ~3));
            const int row = t - u + j;
            if (j <= u && row < Ta && col < pitch) {
              typedef unsigned u4v __attribute__((ext_vector_type(4)));
              const u4v q = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
              // buffer_store_dwordx4 ... sc1 (aux 16): write-through, the line does not stay in this XCD's L2
              __builtin_amdgcn_raw_buffer_store_b128(q, wt_rsrc, (int)((unsigned)(row * pitch + col) * 4u), 0, 16);
            }
          }
        } else if (live && t < Ta) {
          const size_t o = (size_t)t * D + d;
          loading[o] = kept;
          if (pwv_out) pwv_out[o] = pwv;
        }
      }
    }  // chunk loop
    if (live) myflags |= iflags;
    hooks.done(blk);
  }  // item loop
  if (myflags) atomicOr(flags, myflags);
}

}  // namespace mrx_px

#ifdef MRX_PX_CONTRACT_FAST_AFTER
#pragma clang fp contract(fast)
#endif
