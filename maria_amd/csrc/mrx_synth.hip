#include <type_traits>

#include "mrx_internal.h"
#define MRX_PX_CONTRACT_FAST_AFTER  // (the sampler's code keeps its two roundings; the rest of this TU contracts)
#include "mrx_sample_px.h"  // the sampler role of atm_tod_kernel
#include "mrx_spline_tile.h"  // the writer role
#include "mrx_krj.h"          // the K_RJ form's division in the sampler role

namespace {

#ifndef MRX_SYNTH_PIPE
#define MRX_SYNTH_PIPE 0  // 1: the sampler's layer loop as the three-stage gather ring of rounds 4-5 (with MRX_SYNTH_KT 1)
#endif
#ifndef MRX_SYNTH_KT
#define MRX_SYNTH_KT 2  // coarse steps a sampling thread works on side by side: two, their gathers and LDS reads in flight together
#endif
#ifndef MRX_SYNTH_TILE_BATCH
#define MRX_SYNTH_TILE_BATCH 2  // tiles a writing workgroup draws from the queue at a time (1: A/B)
#endif
#ifndef MRX_SYNTH_ACQUIRE
#define MRX_SYNTH_ACQUIRE 0  // 1: the guide's fallback for the consumer -- an agent acquire per tile, plain loads (A/B)
#endif

// ---------------------------------------------------------------------------
// Atmosphere -> TOD in ONE launch (mrx_atm_synthesize): sampling and writing as two kinds of work that ONE resident
// grid takes from two queues, the hand-over between them on the device, in units of a TIME CHUNK.
//
// The observation's work is
//   * sampler items: (detector block b, time chunk c of `chunk` coarse steps, group of 256 detectors) -- the body of
//     atm_sample_px_kernel (px_sample_items) --, handed out chunk by chunk: block b's coarse loading is its own
//     time-major array [Ta][pitch];
//   * writer tiles: (block b, time tile of 1024 samples, group of 32 rows) -- fused_writer_tile, the body of
//     spline_upsample_fused_kernel --, handed out time tile by time tile.  A tile reads the ~66 knots around its
//     samples (fused_tile_knots), i.e. two or three time chunks of its block.
// Round 4's form handed over whole detector blocks: the writers of block b waited for ALL coarse steps of its 512 rows,
// so the sampler role was the launch's critical path until its last block (it runs at half its speed beside the
// writers), a third of the rows had to be sampled before the first writer started (HBM idle for 0.25 ms of 2.0), and
// every detector block re-read the screens' whole track (5-8 GB a step at 16 x 4096^2).  Handing over time chunks,
// the writers follow the samplers a few chunks behind, all detectors of a chunk share one pass over the screens'
// footprint, and the two kinds of work balance themselves:
//   * the first `n_dedicated` workgroups sample until the item queue is empty, then write;
//   * every other workgroup takes a tile; if a chunk its tile needs is not sampled yet it takes ONE sampler item
//     instead of waiting (whichever chunk is next in the queue: sampling never waits for anything), then looks again.
//     So the launch cannot deadlock whatever is resident -- a single workgroup would finish it alone --, at the start
//     every workgroup samples one item (the head start, as long as the first items take), and where the samplers fall
//     behind (16 layers: as much sampling as writing) the writers make up the difference, an item at a time.
// Hand-over (unchanged from round 4 but for its unit): the loading leaves the sampling CUs as write-through (sc1)
// 16-byte stores, every 128-byte line written whole by one store instruction (px_sample_items<..., kWriteThrough>);
// after a work item every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier and ONE
// lane adds 1 to done[b][c] (agent scope, relaxed).  Before a tile, 16 lanes of the workgroup's first wave read the
// next 16 counters past the workgroup's own watermark (global_load_dword sc1; chunks complete roughly in order),
// the workgroup meets at a barrier, and the tile stages its knots with sc1 loads (past the CU's L1).  The per-XCD
// L2s are not coherent and a CU's L1 is not refreshed by other CUs' stores: write-through stores + drained waves +
// one agent-scope add per workgroup on the producer's side, an sc1 poll + a workgroup barrier + sc1 loads on the
// consumer's, is the third row of MI355X_MICROARCH.md's table of hand-offs measured valid on gfx950 without fences --
// in every cell but one: that row was measured with ONE workgroup per CU and this launch runs up to five.  What
// stands in for that cell: no consumer touches a line of a chunk before its counter matches (first touch is an sc1
// load behind the barrier, so no stale copy of it can sit in this CU's L1 from this launch, and sc1 loads do not
// read the L1 anyway), chunks share no line, and tests/test_gpu_synthesize.py changes the data between launches so
// that a byte left over from an earlier launch is a wrong byte (DESIGN 3.0 has the cost of the guide's fallback,
// an agent acquire per tile).
// A tile that waits with the item queue empty polls (s_sleep between) and gives up after `poll_limit` tries, raising
// MRX_FLAG_HANDOVER (the host then fails the call): a bound, not a path.
// Results: the same bits as mrx_atm_sample + mrx_spline_upsample_fused (same bodies, same order of operations;
// tests/test_gpu_synthesize.py).
// kKrj (mrx_atm_synthesize_krj): TOD.to("K_RJ") on the coarse grid, in the sampler's epilogue -- what
// coarse_krj_kernel does to a finished block between the two calls (same functions, same operands: the same bits),
// the last knots kept aside in pW for the samples past the last knot.

// -DMRX_SYNTH_TRACE (scripts/exp/synth_timeline.sh; never in the shipped library): every workgroup logs what it did and when
// -- (kind, number, start, end) on the 100 MHz wall clock: 1 = a writer tile, 2 = a sampler item, 3 = a wait with nothing to do --
// into a table in device memory that mrx_debug_synth_trace copies out.
#ifdef MRX_SYNTH_TRACE
constexpr int kTraceWgs = 2048, kTraceEv = 384;
__device__ uint4 g_trace[kTraceWgs * kTraceEv];
__device__ int g_trace_n[kTraceWgs];
#define MRX_TRACE(kind, id, t0)                                                                                      \
  do {                                                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < kTraceWgs) {                                                                \
      const int k__ = g_trace_n[blockIdx.x];                                                                         \
      if (k__ < kTraceEv) {                                                                                          \
        g_trace[blockIdx.x * kTraceEv + k__] = make_uint4((unsigned)(kind), (unsigned)(id), (unsigned)(t0), (unsigned)wall_clock64()); \
        g_trace_n[blockIdx.x] = k__ + 1;                                                                             \
      }                                                                                                              \
    }                                                                                                                \
  } while (0)
#define MRX_TRACE_RAW(a, b, c, d)                                                                                   \
  do {                                                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < kTraceWgs) {                                                                \
      const int k__ = g_trace_n[blockIdx.x];                                                                         \
      if (k__ < kTraceEv) {                                                                                          \
        g_trace[blockIdx.x * kTraceEv + k__] = make_uint4((unsigned)(a), (unsigned)(b), (unsigned)(c), (unsigned)(d)); \
        g_trace_n[blockIdx.x] = k__ + 1;                                                                             \
      }                                                                                                              \
    }                                                                                                                \
  } while (0)
#define MRX_TRACE_NOW() wall_clock64()
#else
#define MRX_TRACE_RAW(a, b, c, d) do {} while (0)
#define MRX_TRACE(kind, id, t0) do { (void)(t0); } while (0)
#define MRX_TRACE_NOW() 0ull
#endif

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
  unsigned long long tr_item = 0ull;  // (trace build: when the item's prologue -- boresight, anchors -- was through)
  __device__ __forceinline__ void item(int, int d) {
    tr_item = MRX_TRACE_NOW();
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
    const float im = det_sin_elevation(a_re, a_im, bt.x, bt.y);
    const float den = den_lookup(asinf(im), C, cal.n_el, el_first, el_last, el_inv);
    if (cal.tail && real && t >= cal.tail_first) cal.tail[(size_t)(t - cal.tail_first) * cal.ld_tail + d] = v;
    return v * __builtin_amdgcn_rcpf(den);
  }
  int slot = 0;  // the hand-over unit of the work item in progress: block * nby + time chunk
  __device__ __forceinline__ void done(int) {
    const unsigned long long tr_loop = MRX_TRACE_NOW();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's (write-through) stores of the item are out
    const unsigned long long tr_drain = MRX_TRACE_NOW();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctl + kCtlDone + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    MRX_TRACE_RAW(5, tr_loop - tr_item, tr_drain - tr_loop, tr_item);  // steps' loop, the drain, when the prologue ended
  }
};

// Waves per SIMD the launch is compiled for (= its resident workgroups per CU, LDS permitting): the fused writer's 5, i.e. a
// budget of 96 registers, which this kernel -- sampler and writer in one -- exceeds by 25-48 (spilled: DESIGN 3.0).
#ifndef MRX_SYNTH_WAVES
#define MRX_SYNTH_WAVES MRX_WRITER_WAVES
#endif
constexpr int kTileBatch = MRX_SYNTH_TILE_BATCH;
template <bool kLdsTables, bool kHasScale, int kMaxKnots, int kG, bool kKrj>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MRX_SYNTH_WAVES, MRX_SYNTH_WAVES))) void atm_tod_kernel(
    const mrx_layer_fast* __restrict__ fast, const mrx_layer_px* __restrict__ lpx, int n_layers,
    const double2* __restrict__ offpx, const mrx_table_dev* __restrict__ tables, int n_tables,
    const float* __restrict__ table_data, int table_floats, const float* __restrict__ az,
    const float* __restrict__ el, int Ta, const float* __restrict__ dxs, const float* __restrict__ dys,
    const int32_t* __restrict__ band, const float* __restrict__ mueller00, int D, double pwv0,
    double* __restrict__ pwv_out,  // the zenith-scaled pwv per coarse sample, block b as [Ta][rows of b], or null
    float* loading,  // written by sampler items, read by writer tiles: no __restrict__, no const
    uint32_t* __restrict__ flags, int chunk, int nby, int block_rows, int n_blocks, int n_dedicated,
    double ta0, double inv_dta, const double* __restrict__ t, int T,
    const float* __restrict__ scale, const int32_t* __restrict__ rows, float* __restrict__ out, size_t ld, int vec_ok,
    int batches, int tile_order, int* ctl, int poll_limit, int acquire, SynthCal cal) {
  extern __shared__ __align__(16) unsigned char synth_lds[];
  __shared__ int s_word[8];  // what the first wave found out for the workgroup: [0] tile / item, [1] watermark, [2] the tile's last unit, [3..5] its (block, time tile, row group)
  SynthHooks<kKrj> hooks;
  hooks.ctl = ctl;
  hooks.band = band;
  hooks.cal = cal;
  // the cell table of the K_RJ form behind the anchors, the band tables and the four turned steps (16-byte aligned)
  float4* const cells = reinterpret_cast<float4*>(synth_lds) + 2 * chunk * n_layers + (kLdsTables ? (table_floats + 3) / 4 : 0) + kBlock;
  if (kKrj) {  // (what stage_cal_cells puts into cells[0].x and .z)
    hooks.cells = cells;
    hooks.el_first = cal.axis[0];
    hooks.el_last = cal.axis[cal.n_el - 1];
    hooks.el_inv = 1.0f / (cal.axis[1] - cal.axis[0]);
  }
  // ---- the two kinds of work, as numbers ----
  constexpr int kRows = kTileDet * kG;
  const int rows_per_tile = kRows * batches;
  const int last = n_blocks - 1;
  const int last_rows = D - last * block_rows;
  const int nsx = (T + kTileSamples - 1) / kTileSamples;
  const int nrg_full = (block_rows + rows_per_tile - 1) / rows_per_tile, nrg_last = (last_rows + rows_per_tile - 1) / rows_per_tile;
  const int tiles_full = nsx * nrg_full;
  const int n_tiles = tiles_full * last + nsx * nrg_last;
  const int nbx_full = (block_rows + mrx_px::kPxBlock - 1) / mrx_px::kPxBlock, nbx_last = (last_rows + mrx_px::kPxBlock - 1) / mrx_px::kPxBlock;
  const int items_full = nby * nbx_full;
  const int n_items = items_full * last + nby * nbx_last;
  const int n_slots = n_blocks * nby;

  bool sampling = (int)blockIdx.x < n_dedicated;  // (workgroup-uniform, like everything that steers the loop)
  bool items_left = true;
  int tile = -1;   // the tile in hand, not yet written
  int tile_end = 0, ready_end = 0;  // the batch in hand is tile .. tile_end - 1; tiles below ready_end have their chunks in
  int seen = 0;    // the last ticket this workgroup drew (how far the queue has got, roughly)
  // (block, time tile, row group) of tile `tl`
  auto tile_of = [&](int tl, int& blk, int& sx, int& rg) {
    blk = last == 0 ? 0 : min(tl / tiles_full, last);
    const int rem = tl - blk * tiles_full;
    const int nrg = blk == last ? nrg_last : nrg_full;
    if (tile_order == 0) {  // time tile by time tile, the block's row groups side by side
      sx = rem / nrg;
      rg = rem - sx * nrg;
    } else {  // row group by row group, its time tiles in a row
      rg = rem / nsx;
      sx = rem - rg * nsx;
    }
  };
  int have = 0;    // hand-over units 0 .. have - 1 are known to be sampled (blocks in order, chunks in order)
  int tries = 0;
  for (;;) {
    __syncthreads();  // everybody is done with the previous turn's LDS: the images, the tables, s_word
    bool take_item = sampling;
    if (!sampling) {
      // ---- tiles: a batch of tickets if none is in hand, then whether the batch's chunks are sampled ----
      // A workgroup takes kTileBatch consecutive tiles from the queue at a time -- neighbouring row groups of one time
      // tile (or that tile's last and the next one's first): they read the same chunks -- and asks once for all of them,
      // about the last one (the units complete in order: what the last tile reads is in when everything before it is).
      // Between two tiles of a batch nothing is asked of memory.  One ticket, one look at t[], one poll a TILE were three
      // dependent round trips behind the draining stores of the tile just written: 2.5 us before every 13-16-us tile
      // (profiles/r06_synth_timeline.txt).  Near the end of the queue a workgroup takes single tiles again (the tail).
      if (tile < 0 || tile >= ready_end) {
        if (threadIdx.x < 64) {  // the first wave
          int tl = tile, te = tile_end, need = 0;
          const unsigned long long tg0 = MRX_TRACE_NOW();
          if (tl < 0) {
            const int want = seen + 2 * kTileBatch * (int)gridDim.x >= n_tiles ? 1 : kTileBatch;
            if (threadIdx.x == 0) tl = __hip_atomic_fetch_add(ctl + kCtlTiles, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tl = __builtin_amdgcn_readfirstlane(tl);
            te = min(tl + want, n_tiles);
          }
          const unsigned long long tg1 = MRX_TRACE_NOW();
          unsigned long long tg2 = tg1;
          int hv = have;
          if (tl < n_tiles) {
            // (the integer divisions of this decode are a few dozen vector instructions each -- there is no scalar one)
            int blk, sx, rg;
            tile_of(te - 1, blk, sx, rg);
            int lo, hi;
            fused_tile_knots(t, T, Ta, ta0, inv_dta, sx, lo, hi);
            need = blk * nby + hi / chunk;  // the last unit the batch reads (a tile's knots lie in one block)
            tg2 = MRX_TRACE_NOW();
            // the watermark: 16 counters a look, on while all 16 are complete and the batch's are not reached
            while (hv <= need) {
              const int sl = hv + (int)threadIdx.x;
              bool ok = false;
              if (threadIdx.x < 16 && sl < n_slots) {
                const int want = sl >= last * nby ? nbx_last : nbx_full;
                ok = __hip_atomic_load(ctl + kCtlDone + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
              }
              const unsigned long long m = __builtin_amdgcn_ballot_w64(ok) & 0xffffull;
              const int adv = m == 0xffffull ? 16 : __builtin_ctzll(~m);
              hv += adv;
              if (adv < 16) break;
            }
          }
          MRX_TRACE_RAW(4, tg1 - tg0, tg2 - tg1, MRX_TRACE_NOW() - tg2);  // ticket, decode + knots, poll
          if (threadIdx.x == 0) {
            s_word[0] = tl;
            s_word[1] = hv;
            s_word[2] = need;
            s_word[3] = te;
          }
          // MRX_OPT_SYNTH_ACQUIRE (round 6: a RUNTIME switch; the build switch MRX_SYNTH_ACQUIRE = 1 also turns the tile's sc1
          // loads into plain ones): the polling wave, once its poll has matched, runs the agent-scope acquire -- it invalidates
          // this CU's L1 -- and waits for it; the barrier below then holds every other wave behind it.  With the producer's
          // sc1 stores and drained waves that is one of MI355X_MICROARCH.md's always-valid hand-offs, whatever the number of
          // workgroups a CU holds; the tile's loads stay sc1 (past the L1 anyway), so the switch only adds the fence.
          if ((MRX_SYNTH_ACQUIRE || acquire) && hv > need) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
        __syncthreads();  // between the poll and EVERY load of the chunks' bytes, the polling wave's own too
        tile = s_word[0];
        have = s_word[1];
        const int need = s_word[2];
        tile_end = s_word[3];
        if (tile >= n_tiles) break;  // (the queue is empty: nothing in hand, nothing left)
        seen = tile;
        bool ready = have > need;
        if (!ready && !items_left) {  // the chunk is being sampled by others: wait for it
          if (++tries > poll_limit) {
            if (threadIdx.x == 0) atomicOr(flags, MRX_FLAG_HANDOVER);
            ready = true;
          } else {
            const unsigned long long tw0 = MRX_TRACE_NOW();
            __builtin_amdgcn_s_sleep(32);
            MRX_TRACE(3, tile, tw0);
            continue;
          }
        }
        if (ready) ready_end = tile_end;
      }
      if (tile < ready_end) {
        int blk, sx, by;
        tile_of(tile, blk, sx, by);  // (every wave for itself: no word to pass, no barrier)
        const int Db = blk == last ? last_rows : block_rows;
        const size_t row0 = (size_t)blk * block_rows;
        const unsigned long long tt0 = MRX_TRACE_NOW();
        fused_writer_tile<kHasScale, kMaxKnots, kG, MRX_SYNTH_ACQUIRE == 0>(loading + (size_t)Ta * row0, (Db + 31) & ~31, Db, Ta, ta0, inv_dta, t, T,
                                                           kHasScale ? scale + row0 : nullptr, rows ? rows + row0 : nullptr,
                                                           rows ? out : out + row0 * ld, ld, vec_ok, batches, sx, by, synth_lds);
        MRX_TRACE(1, tile, tt0);
        if (++tile == tile_end) tile = -1;
        tries = 0;
        continue;
      }
      take_item = true;  // instead of waiting: one item of whatever is next to be sampled
      __syncthreads();   // (s_word is read: the item's number goes there next)
    }
    if (take_item) {
      if (threadIdx.x == 0) s_word[0] = __hip_atomic_fetch_add(ctl + kCtlItems, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int item = s_word[0];
      if (item >= n_items) {  // everything is sampled or being sampled: from here on this workgroup writes
        items_left = false;
        sampling = false;
        continue;
      }
      // items: block by block, time chunk by time chunk, the chunk's detector groups side by side
      const int blk = min(item / items_full, last);
      const int rem = item - blk * items_full;
      const int nbx = blk == last ? nbx_last : nbx_full;
      const int by = rem / nbx, bx = rem - by * nbx;
      hooks.slot = blk * nby + by;
      const unsigned long long ti0 = MRX_TRACE_NOW();
      if (kKrj) stage_cal_cells(cells, cal.axis, cal.values, cal.n_el, cal.n_bands);  // (the item starts with a barrier)
      // px_sample_items walks the items of "workgroup w of W" in its own order -- XCD w mod 8 takes the chunks
      // w mod 8, + 8, ... -- and W = 2^30 makes that walk exactly ONE item long: the one numbered (by, bx)
      mrx_px::px_sample_items<kLdsTables, MRX_SYNTH_KT, MRX_SYNTH_PIPE != 0, true>(
          fast, lpx, n_layers, offpx, tables, n_tables, table_data, table_floats, az, el, Ta, dxs, dys, band, mueller00,
          D, pwv0, pwv_out, loading, flags, chunk, nby, block_rows, n_blocks, blk, blk + 1, ((by >> 3) * nbx + bx) * 8 + (by & 7),
          1 << 30, reinterpret_cast<float4*>(synth_lds), hooks);
      MRX_TRACE(2, item, ti0);
    }
  }
  synth_leave(ctl, n_slots);
}

}  // namespace

extern "C" {

// Rows per block of the coarse array (and of the hand-over): the caller's number in whole groups of 256 lanes, or the
// library's choice.
static int synth_block_rows(mrx_ctx* ctx, const mrx_atm_plan* plan, int D, int Ta, int block_rows, int* out) {
  // one block where the block's coarse array stays below 2 GiB (the sampler addresses it as a raw buffer)
  {
    const long long cap = ((1LL << 31) - 4096) / (4LL * Ta) / kBlock * kBlock;  // rows of whole groups of 256 lanes
    MRX_REQUIRE(ctx, cap >= kBlock, "Ta too large: 256 rows of the coarse loading must stay below 2 GiB");
    if (block_rows <= 0) {
      // The library's choice.  The tiles of a block are written time tile by time tile, all its row groups side by side;
      // cut into blocks of about 5 000 rows the launch measured 3-7 % faster than in one (atlast_10k's shape at 7 000,
      // 8 000, 10 000 and 20 000 rows: 1.41 / 1.61 / 1.93 / 3.88 ms against 1.50 / 1.68 / 2.08 / 3.98; blocks of 2 048 to
      // 6 912 rows alike) -- but every block walks the screens' track again, which is free only while the screens stay
      // in the 256 MiB Infinity Cache (8 x 2048^2: 134 MB): 16 x 4096^2 (1.07 GB) in blocks of 3 328 rows took 9.45 ms
      // against 9.22 in one block, 12 500 rows of that shape 17.2 / 17.4 in two / three blocks against 17.0 in one
      const long long all = (long long)mrx_ceil_div(D, kBlock) * kBlock;
      long long want = all;
      if (plan->screen_bytes <= (192ull << 20) && D > 6400) want = (long long)mrx_ceil_div(mrx_ceil_div(D, mrx_ceil_div(D, 5120)), kBlock) * kBlock;
      block_rows = (int)std::min(cap, want);
    } else if (block_rows > cap) {
      block_rows = (int)cap;
    }
  }
  *out = std::min(mrx_ceil_div(block_rows, kBlock) * kBlock, mrx_ceil_div(D, 32) * 32);  // whole groups of 256 lanes; rows of whole lines
  return MRX_OK;
}

static int atm_synthesize(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az, const float* d_el, int Ta,
                          const float* d_dx, const float* d_dy, const int32_t* d_band, const float* d_mueller00, int D,
                          double pwv0, float* d_coarse, int block_rows, int sampler_wgs, uint32_t* d_flags, double ta0,
                          double dta, const double* d_t, int T, const float* d_scale, const int32_t* d_rows, float* d_out,
                          size_t ld_out, const SynthCal* krj, double* d_pwv) {
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
  {
    const int rc = synth_block_rows(ctx, plan, D, Ta, block_rows, &block_rows);
    if (rc != MRX_OK) return rc;
  }
  const int n_blocks = mrx_ceil_div(D, block_rows);
  MRX_REQUIRE(ctx, (long long)Ta * block_rows * 4 < (1LL << 31), "a block's coarse array must stay below 2 GiB");
  MRX_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(d_coarse) & 127u) == 0, "d_coarse must be 128-byte aligned");
  // ---- writer tiles: mrx_spline_upsample_fused's choices ----
  const int vec_ok = (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_out) & 15u) == 0);
  const double knots_per_tile = (double)kTileSamples * (double)Ta / (double)T;
  const bool small = knots_per_tile + 6.0 <= (double)kSmallKnots;
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
  const size_t lds_w = small ? FusedLds<kSmallKnots, 2>::kBytes : FusedLds<256, 1>::kBytes;
  // ---- sampler items: the time chunk is the unit of the hand-over ----
  int chunk = ctx->options[MRX_OPT_SAMPLE_CHUNK];
  if (chunk <= 0) {
    // 32 steps where that makes a few work items per resident workgroup (atlast_10k: 7 520 items, 2.00 ms against 2.04 at
    // 64 steps and 2.08 at 16), shorter chunks for small shards, whose few detector groups would otherwise leave most
    // of the chip without an item while the first chunks are sampled (2 512 rows: 0.54 ms at 16 steps, 0.58 at 32;
    // 1 264 rows: 0.32 against 0.37)
    chunk = 32;
    while (chunk > 8 && (long long)mrx_ceil_div(Ta, chunk) * mrx_ceil_div(D, kBlock) < 5LL * MRX_SYNTH_WAVES * n_cu / 2) chunk /= 2;
  }
  chunk = chunk < 1 ? 1 : chunk > mrx_px::kMaxChunk ? mrx_px::kMaxChunk : chunk;
  while (chunk > 1 && chunk * plan->n_layers > mrx_px::kMaxAnchors) chunk /= 2;
  // the launch has ONE dynamic LDS size and a CU's LDS is what bounds its workgroups: the sampler's anchors stay under
  // the writer's images (16 layers at 64 steps a work item took 37 KB -- and a workgroup's place on every CU)
  const size_t lds_turn = sizeof(float) * 4 * kBlock;  // four steps of every lane (px_sample_items<..., kWriteThrough>)
  while (chunk > 8 && 2 * sizeof(float4) * (size_t)chunk * plan->n_layers + lds_turn > lds_w) chunk /= 2;
  while (chunk < mrx_px::kMaxChunk && (long long)n_blocks * mrx_ceil_div(Ta, chunk) > kSynthMaxSlots) chunk *= 2;
  const int nby = mrx_ceil_div(Ta, chunk);
  MRX_REQUIRE(ctx, (long long)n_blocks * nby <= kSynthMaxSlots, "too many (detector block, time chunk) units for one launch");
  const long long n_items = (long long)nby * ((long long)(n_blocks - 1) * mrx_ceil_div(block_rows, kBlock) +
                                              mrx_ceil_div(D - (n_blocks - 1) * block_rows, kBlock));
  MRX_REQUIRE(ctx, n_items <= 0x7fffffffLL - 65536, "too many work items for one launch");
  const size_t lds_anchor = 2 * sizeof(float4) * (size_t)chunk * plan->n_layers;
  const size_t lds_tables = sizeof(float) * (size_t)((plan->table_floats + 3) / 4 * 4);
  // band tables in LDS only where they fit under the writer's images too
  const size_t lds_cal = krj ? sizeof(float4) * (size_t)(krj->n_el - 1) * krj->n_bands : 0;  // the K_RJ cell table
  const bool lds_tab = plan->table_floats <= mrx_px::kMaxLdsTableFloats && lds_anchor + lds_tables + lds_turn + lds_cal <= lds_w;
  if (lds_anchor + (lds_tab ? lds_tables : 0) + lds_turn + lds_cal > lds_w)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "mrx_atm_synthesize: the calibration table does not fit beside the sampler's LDS: use the two calls");
  const size_t lds = lds_w;
  // ---- the grid: as many workgroups as a CU holds, all resident, each taking tiles and items until both queues are
  // empty; the first `dedicated` only sample while items remain (MRX_OPT_SAMPLE_WGS_PER_CU per CU, or the caller's number)
  long long per_cu = std::max<long long>(1, std::min<long long>(MRX_SYNTH_WAVES, (long long)(ctx->lds_per_cu > 0 ? ctx->lds_per_cu : 160 * 1024) / (long long)(lds + 1600)));
  if (ctx->options[MRX_OPT_SYNTH_WGS_PER_CU] > 0) per_cu = std::min<long long>(per_cu, ctx->options[MRX_OPT_SYNTH_WGS_PER_CU]);
  const long long wgs = std::min(n_tiles + n_items, per_cu * n_cu);
  long long dedicated = sampler_wgs;
  if (dedicated <= 0) {
    int s_per_cu = ctx->options[MRX_OPT_SAMPLE_WGS_PER_CU];
    if (s_per_cu <= 0 || s_per_cu >= 8) s_per_cu = 2;
    dedicated = (long long)s_per_cu * n_cu;
  }
  dedicated = std::min(dedicated, std::max(0LL, wgs - 1));  // (somebody has to write while items remain: or all would sample first)
  if (ctx->options[MRX_OPT_SAMPLE_WGS_PER_CU] >= 8) dedicated = 0;  // (A/B: nobody only samples)
  int* ctl = nullptr;
  {
    const int rc = mrx_synth_ctl(ctx, &ctl);
    if (rc != MRX_OK) return rc;
  }
  const dim3 grid((unsigned)wgs);
  const int poll_limit = 1 << 22;  // x >= 0.5 us a try: seconds
  const SynthCal cal = krj ? *krj : SynthCal{};
#define MRX_LAUNCH_SYNTH(L, S, K, G, J)                                                                           \
  do {                                                                                                            \
    MRX_LDS_CAP(ctx, (atm_tod_kernel<L, S, K, G, J>), lds);                                                       \
    hipLaunchKernelGGL((atm_tod_kernel<L, S, K, G, J>), grid, dim3(kBlock), lds, ctx->stream, plan->d_fast,       \
                       plan->d_px, plan->n_layers, plan->d_offpx, plan->d_tables, plan->n_tables,                 \
                       plan->d_table_data, plan->table_floats, d_az, d_el, Ta, d_dx, d_dy, d_band, d_mueller00,   \
                       D, pwv0, d_pwv, d_coarse, d_flags, chunk, nby, block_rows, n_blocks, (int)dedicated, ta0, 1.0 / dta, \
                       d_t, T, d_scale, d_rows, d_out, ld_out, vec_ok, batches, ctx->options[MRX_OPT_SYNTH_TILE_ORDER], ctl, poll_limit,       \
                       ctx->options[MRX_OPT_SYNTH_ACQUIRE] != 0 ? 1 : 0, cal);                                    \
  } while (0)
#define MRX_LAUNCH_SYNTH_J(L, S, K, G) do { if (krj) MRX_LAUNCH_SYNTH(L, S, K, G, true); else MRX_LAUNCH_SYNTH(L, S, K, G, false); } while (0)
#define MRX_LAUNCH_SYNTH_S(L, S) do { if (small) MRX_LAUNCH_SYNTH_J(L, S, kSmallKnots, 2); else MRX_LAUNCH_SYNTH_J(L, S, 256, 1); } while (0)
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

int mrx_atm_synthesize_block_rows(mrx_ctx* ctx, const mrx_atm_plan* plan, int D, int Ta, int block_rows, int* rows_out) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, plan && rows_out && D > 0 && Ta > 0, "need a plan, D > 0, Ta > 0 and an output");
  return synth_block_rows(ctx, plan, D, Ta, block_rows, rows_out);
}

int mrx_atm_synthesize(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az, const float* d_el, int Ta,
                       const float* d_dx, const float* d_dy, const int32_t* d_band, const float* d_mueller00, int D,
                       double pwv0, float* d_coarse, int block_rows, int sampler_wgs, uint32_t* d_flags, double ta0,
                       double dta, const double* d_t, int T, const float* d_scale, const int32_t* d_rows, float* d_out,
                       size_t ld_out, double* d_pwv) {
  MRX_ENTER(ctx);
  return atm_synthesize(ctx, plan, d_az, d_el, Ta, d_dx, d_dy, d_band, d_mueller00, D, pwv0, d_coarse, block_rows, sampler_wgs,
                        d_flags, ta0, dta, d_t, T, d_scale, d_rows, d_out, ld_out, nullptr, d_pwv);
}

int mrx_atm_synthesize_krj(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az, const float* d_el, int Ta,
                           const float* d_dx, const float* d_dy, const int32_t* d_band, const float* d_mueller00, int D,
                           double pwv0, float* d_coarse, int block_rows, int sampler_wgs, uint32_t* d_flags, double ta0,
                           double dta, const double* d_t, int T, const float* d_scale, const int32_t* d_rows, float* d_out,
                           size_t ld_out, const float* d_cal_dx, const float* d_cal_dy, const float* d_cal_axis_el,
                           const float* d_cal_values, int n_el, int n_bands, float* d_tail_pw, int tail_knots, size_t ld_tail,
                           double* d_pwv) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, d_cal_dx && d_cal_dy && d_cal_axis_el && d_cal_values, "null calibration pointer");
  MRX_REQUIRE(ctx, n_el >= 2 && n_bands >= 1, "calibration tables need 2 <= n_el, 1 <= n_bands");
  MRX_REQUIRE(ctx, tail_knots >= 0 && tail_knots <= Ta && (!d_tail_pw || ld_tail >= (size_t)D), "bad tail window");
  SynthCal cal{d_cal_dx, d_cal_dy, d_cal_axis_el, d_cal_values, n_el, n_bands, tail_knots > 0 ? d_tail_pw : nullptr, Ta - tail_knots, ld_tail};
  return atm_synthesize(ctx, plan, d_az, d_el, Ta, d_dx, d_dy, d_band, d_mueller00, D, pwv0, d_coarse, block_rows, sampler_wgs,
                        d_flags, ta0, dta, d_t, T, d_scale, d_rows, d_out, ld_out, &cal, d_pwv);
}

}  // extern "C"

#ifdef MRX_SYNTH_TRACE
// events: [2048][384] x 16 bytes, counts: [2048] ints (host buffers); reset != 0 clears the counts afterwards
extern "C" int mrx_debug_synth_trace(void* events, void* counts, int reset) {
  if (hipMemcpyFromSymbol(events, HIP_SYMBOL(g_trace), sizeof(uint4) * kTraceWgs * kTraceEv) != hipSuccess) return MRX_ERR_HIP;
  if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_trace_n), sizeof(int) * kTraceWgs) != hipSuccess) return MRX_ERR_HIP;
  if (reset) {
    static int zeros[kTraceWgs];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), zeros, sizeof(zeros)) != hipSuccess) return MRX_ERR_HIP;
  }
  return MRX_OK;
}
#endif
