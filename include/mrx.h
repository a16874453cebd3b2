/*
 * mrx.h -- C ABI of libmrx, the MI355X (gfx950) implementation of maria's
 * atmospheric-turbulence + time-ordered-data (TOD) synthesis hot path.
 *
 * The reference (thomaswmorris/maria) is pure Python and has no FFI for this
 * path; the entry points below are what a ctypes binding placed at the
 * reference's own Python seams would call.  Each function cites the reference
 * code (file:line under maria/) whose arithmetic it replaces.
 *
 * Conventions
 *  - every function returns 0 (MRX_OK) or a negative mrx_status; nothing throws.
 *    mrx_last_error(ctx) returns a human-readable message for the last failure.
 *  - the CALLER owns every buffer.  Pointers named d_* are DEVICE pointers
 *    (e.g. torch-ROCm tensor.data_ptr()); everything else is host memory that
 *    is only read during the call.
 *  - all work is enqueued on the HIP stream bound with mrx_set_stream()
 *    (default: the null stream); no call synchronises the device except
 *    mrx_read_flags() and mrx_synchronize().
 *  - one mrx_ctx per host thread per device.
 *  - coarse (atmosphere-rate) arrays are TIME-MAJOR: index [t * D + d] with
 *    D = detector rows of this shard; the full-rate TOD is DETECTOR-MAJOR
 *    [d * ld + s], the reference's (ndet, nt) layout (sim/atmosphere.py:72-82).
 */
#ifndef MRX_H_
#define MRX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRX_VERSION 150 /* 0.2.0 (120): mrx_spline_upsample_fused, mrx_allgather_tod_p2p, mrx_exchange_screens,
                           mrx_resample_columns; MRX_OPT_SAMPLE_TILES retired.  130: mrx_screen_amplitudes,
                           mrx_screen_desc.d_amp, mrx_screen_generate_3d(..., d_amp), mrx_streams_concurrent.
                           131: mrx_coarse_to_krj_keep_tail.  140: mrx_screen_desc.periodic_beam,
                           mrx_noise_generate_krj, the noise generator's two-rate form.  141: mrx_atm_synthesize,
                           MRX_FLAG_HANDOVER.  142: mrx_atm_synthesize_krj, MRX_OPT_WRITER_PER_TILE.  150: mrx_map_cal.steps_per_tile
                           (the struct's reserved word), MRX_OPT_GAUSS_ACCUM, MRX_OPT_SYNTH_ACQUIRE; mrx_map_sample reads a map
                           of less than 2 GiB through a context-owned copy */

typedef enum mrx_status {
  MRX_OK = 0,
  MRX_ERR_INVALID = -1,  /* bad argument (null pointer, non-positive size ...) */
  MRX_ERR_HIP = -2,      /* a HIP runtime call failed; see mrx_last_error    */
  MRX_ERR_NO_DEVICE = -3,
  MRX_ERR_UNSUPPORTED = -4, /* shape outside what the kernels are built for  */
  MRX_ERR_ALLOC = -5
} mrx_status;

/* bits of the device flag word written by the sampling kernel */
#define MRX_FLAG_SCREEN_OOB 1u /* a line of sight left a layer's screen: the
                                  reference raises RuntimeError "introduced nans"
                                  (atmosphere/atmosphere.py:368-369)            */
#define MRX_FLAG_TABLE_OOB 2u  /* (pwv, el) left the emission table: jax fill
                                  value NaN (band/band.py:283-286)              */
#define MRX_FLAG_NAN 4u        /* a NaN reached the output                      */
#define MRX_FLAG_HANDOVER 8u   /* mrx_atm_synthesize: a writer gave up waiting for the sampler (its bound of
                                  ~seconds; the call's TOD is then invalid)     */

typedef struct mrx_ctx mrx_ctx;
typedef struct mrx_atm_plan mrx_atm_plan;
typedef struct mrx_comm mrx_comm;

/* ---- context ------------------------------------------------------------ */

int mrx_version(void);
/* Creates a context on HIP device `device`.  Fails with MRX_ERR_NO_DEVICE when
 * no gfx950 device is visible: there is no CPU fallback in this library. */
int mrx_init(int device, mrx_ctx** ctx);
int mrx_destroy(mrx_ctx* ctx);
/* Bind the hipStream_t all later calls enqueue on (pass NULL for the null
 * stream).  The stream stays owned by the caller. */
int mrx_set_stream(mrx_ctx* ctx, void* hip_stream);
/* Do kernels on the context's stream and on `other_stream` (a hipStream_t) run side by side?  HIP spreads
 * the streams of a process round-robin over a few hardware queues (four by default); two streams on one
 * queue take turns, whatever the events between them say.  A caller that pipelines work over two streams
 * (DevicePath.run: the sampler of block b+1 beside the writer of block b) asks this once and takes another
 * stream when the answer is 0 -- one new stream in four shares the current stream's queue, and the
 * pipelined step then runs as slowly as the serial one (2.9 instead of 2.1 ms).  Two 150 us spin kernels, timed;
 * synchronises both streams.  The library's own side streams (mrx_noise_generate) are chosen the same way. */
int mrx_streams_concurrent(mrx_ctx* ctx, void* other_stream, int* concurrent);
int mrx_synchronize(mrx_ctx* ctx);
/* Options (default 0).
 *  MRX_OPT_POINTING_CHAIN = 1: mrx_atm_sample follows the reference's float32
 *    chain literally (atan2 -> phi, asin -> theta, tan/cos/sin of those).  The
 *    default evaluates the same ground projection directly from the unit
 *    line-of-sight vector: equal up to the float32 rounding noise of the chain,
 *    better conditioned near the zenith, and ~3x fewer instructions.
 *  MRX_OPT_AXIS_LITERAL = 1: mrx_atm_sample finds cell and weight on every screen axis with
 *    jax's float32 rule (searchsorted on the float32 nodes, weight (x - lo)/(hi - lo)), also on
 *    uniform axes.  The default computes the position in pixels on axes whose uniform hint
 *    verified (a float64 anchor per step and layer + the lane's float32 offset from it, good to
 *    ~4e-6 pixel): equal to the literal rule up to the float32 rounding of the reference's own
 *    coordinates (~2e-4 pixel, ~1e-8 of the loading), at a third of the instructions.
 *    MRX_OPT_POINTING_CHAIN implies the literal rule.
 *  MRX_OPT_SAMPLE_TIMES = 1|2|4: coarse time steps per thread in mrx_atm_sample
 *    (tuning; 0 = library default). */
enum {
  MRX_OPT_POINTING_CHAIN = 0,
  MRX_OPT_AXIS_LITERAL = 1,
  MRX_OPT_SAMPLE_TIMES = 2,
  MRX_OPT_SAMPLE_CHUNK = 3, /* time steps per workgroup (tuning; 0 = automatic) */
  MRX_OPT_UPSAMPLE_GROUPS = 4, /* 16-row detector tiles per workgroup of the TOD
                                  writer (tuning; 0 = automatic) */
  MRX_OPT_NOISE_GENERIC = 5, /* bit 0: the LDS second pass even where the register one applies;
                                bit 1: the Stockham first pass even where the radix-16 register
                                one applies; bit 2: the lanes' first batches of equal length
                                (tests, A/B runs) */
  MRX_OPT_SAMPLE_WGS_PER_CU = 6, /* mrx_atm_sample runs as a resident grid of this many workgroups
                                    per CU that walk the work items (tuning; 0 = default, 8) */
  MRX_OPT_WRITER_PER_TILE = 7, /* 1: mrx_spline_upsample_fused launches one workgroup per tile instead of a resident
                                  grid over a tile queue (the default, 6 % faster alone): for a launch that shares the
                                  chip with kernels launched after it on another stream -- a resident grid never makes
                                  room for them.  (Slot of the retired MRX_OPT_SAMPLE_TILES.) */
  MRX_OPT_NOISE_LANES = 8, /* streams mrx_noise_generate spreads its batches over (1..4; 0 = automatic:
                              up to 4, each with at least 128 detectors of the work buffer) */
  MRX_OPT_SCREEN_STOCKHAM = 9, /* 1: the screen generator's transforms as LDS Stockham passes even
                                  where the register transforms apply (tests, A/B runs) */
  MRX_OPT_SYNTH_WGS_PER_CU = 10, /* mrx_atm_synthesize: resident workgroups per CU of its one grid (1..5; 0 = as many as
                                    fit, 5).  Timing sweeps, and the tests' way to vary who runs beside whom */
  MRX_OPT_SYNTH_TILE_ORDER = 11, /* mrx_atm_synthesize: 0 = the tiles of a detector block time tile by time tile (all its row groups
                                    side by side: the writers follow the samplers a few chunks behind); 1 = row group by row group, the
                                    time tiles of a row group in a row (the order the chip stores fastest; a block's tiles then wait
                                    for the whole block: give block_rows) */
  MRX_OPT_GAUSS_ACCUM = 12, /* mrx_gauss_smooth2d / mrx_map_smooth: 0 = float32 products summed over 16 taps and those sums
                               in float64 from radius 16 on (<= 1e-6 of scipy's float64 sums, twice as fast), scipy's float64
                               arithmetic below; 1 = float64 always ("exact": <= 2.5e-7, float32 roundings of the result);
                               2 = the blocked sums at every radius */
  MRX_OPT_SYNTH_ACQUIRE = 13, /* mrx_atm_synthesize: 1 = a writer tile's workgroup runs an agent-scope acquire (and waits for it)
                                 once the chunks it needs are in, before it loads them -- the hand-over form that is valid
                                 whatever a CU holds; 0 (default) = write-through stores, drained waves and sc1 loads alone,
                                 measured valid and 2 % faster */
  MRX_OPT_COUNT = 14
};
int mrx_set_option(mrx_ctx* ctx, int option, int value);
const char* mrx_last_error(const mrx_ctx* ctx);
/* Device properties the host side sizes launches with. */
int mrx_device_info(const mrx_ctx* ctx, int* n_cu, int* lds_bytes_per_cu,
                    size_t* hbm_bytes, char* name, int name_len);

/* HIP-event timer on the bound stream (bench.py's live kernel timing; a
 * torch.cuda.Event only sees torch's own current stream). */
int mrx_timer_start(mrx_ctx* ctx);
int mrx_timer_stop(mrx_ctx* ctx, float* elapsed_ms); /* synchronises */

/* ---- turbulent layer stack + emission tables (the "plan") ---------------- */

/* One turbulent layer = one smoothed screen on a rectilinear grid in the
 * process frame (atmosphere/atmosphere.py:317-373).  A line of sight with
 * unit-height ground projection (px, py) (coords/coordinates.py:333-349,
 * z = 1) hits the layer at
 *     e = h*(px*r00 + py*r10) + d_off_e[t]
 *     c = h*(px*r01 + py*r11) + d_off_c[t]
 * where (r..) is process.transform (utils/rotations.py:45-77) and
 * d_off_*[t] = (cumsum(timestep*(vx,vy,0)) + (0,0,h)) @ transform, columns 0/1
 * (atmosphere/atmosphere.py:318-319,346-347), evaluated by the host in f64. */
typedef struct mrx_layer {
  const float* d_values; /* [n_e][n_c] f32, smoothed screen (:341-344)        */
  const float* d_axis_e; /* [n_e] f32 grid nodes = float32(process.extrusion)  */
  const float* d_axis_c; /* [n_c] f32 = float32(process.cross_section[mask,0]) */
  const double* d_off_e; /* [Ta] f64                                           */
  const double* d_off_c; /* [Ta] f64                                           */
  int32_t n_e, n_c;
  double h;              /* layer.h, metres                                    */
  double r00, r10, r01, r11;
  /* Optional uniform-axis hint: the float64 grids behind the two axes are
   * extrusion[i] = e0 + i*de (np.arange, atmosphere.py:241-245) and
   * cross_section[i] = c0 + i*dc (np.linspace, :208-219).  When
   * float32(e0 + i*de) reproduces every node of d_axis_e to one float32 ulp (checked on
   * the device at plan creation; the float64 grid is itself rounded, so a node in ~1e5 parts
   * from the model by an ulp) the kernel works in pixel coordinates, (x - e0)/de in float64,
   * instead of searching the axis (see MRX_OPT_AXIS_LITERAL).  Set de / dc to 0 when unknown.
   * Screens of 2^22 nodes a side or 4 GiB and more take the general kernel. */
  double e0, de, c0, dc;
  float pwv_rms;         /* float32(layer.pwv_rms), extrusion.py:100-105       */
  int32_t reserved;
} mrx_layer;

/* Band-integrated emission table (band/band.py:264-286).  The reference does
 * a trilinear lookup at (T0, pwv, el) with scalar T0; the host passes the two
 * temperature slabs that bracket T0 and T0's normalised distance between them
 * so the kernel reproduces the reference's 8-term float32 sum term by term. */
typedef struct mrx_band_table {
  const float* d_values;   /* [2][n_pwv][n_el] f32: slab iT then slab iT+1    */
  const float* d_axis_pwv; /* [n_pwv] f32, spectrum.side_zenith_pwv (mm)      */
  const float* d_axis_el;  /* [n_el]  f32, spectrum.side_elevation (rad)      */
  int32_t n_pwv, n_el;
  float w_t;               /* float32 (T0 - T[iT]) / (T[iT+1] - T[iT])        */
  int32_t t_oob;           /* 1 if T0 lies outside the table (result NaN)     */
  /* interpolation_method="cubic" (band/band.py:288-300): the table interpolated linearly to
   * T0 (scipy interp1d, float64) and then scipy's RegularGridInterpolator(method="cubic") on
   * (pwv, el), i.e. the tensor-product not-a-knot cubic spline, float64.  The host expands
   * that spline into one bicubic polynomial per grid cell:
   *   d_cubic = [n_pwv float64 pwv nodes][n_el float64 el nodes]
   *             [(n_pwv-1)*(n_el-1) cells, row-major (pwv, el)][16]: c[4*k + m] multiplies
   *             (pwv - pwv_i)^m (el - el_j)^k.
   * NULL selects the linear (jax, float32) lookup above.  A sample outside the grid sets
   * MRX_FLAG_TABLE_OOB (scipy raises ValueError there). */
  const double* d_cubic;
} mrx_band_table;

/* Builds the device-resident plan from the layer and table descriptors (host
 * structs holding device pointers): packs the wind offsets of all layers
 * [n_t][n_layers], copies the band tables into one buffer (staged in LDS by the
 * kernel) and verifies the uniform-axis hints.  The screens (d_values) and axis
 * arrays must stay alive while the plan is in use; the screens may be
 * rewritten in place between calls (a new realisation), the rest may not.
 * n_t = length of every d_off_e / d_off_c array = Ta of mrx_atm_sample. */
int mrx_atm_plan_create(mrx_ctx* ctx, const mrx_layer* layers, int n_layers,
                        const mrx_band_table* tables, int n_tables, int n_t,
                        mrx_atm_plan** plan);
int mrx_atm_plan_destroy(mrx_ctx* ctx, mrx_atm_plan* plan);
/* Diagnostics: number of layer axes (0..2*n_layers) that passed the uniform
 * check, and whether the band tables fit the kernel's LDS stage. */
int mrx_atm_plan_info(mrx_ctx* ctx, const mrx_atm_plan* plan, int* uniform_axes,
                      int* tables_in_lds);

/* ---- hot path ------------------------------------------------------------- */

/* Fused coarse-rate sampling: detector pointing (coords/transforms.py:10-29 via
 * coordinates.py:378-386), ground projection (coordinates.py:333-349),
 * wind-advected bilinear gather through the layer stack
 * (atmosphere/atmosphere.py:346-373, jax RegularGridInterpolator "linear",
 * float32), emission lookup (band/band.py:264-286) and Mueller weight
 * (sim/atmosphere.py:64-65, array/array.py:204-218).
 *
 *  d_az, d_el     [Ta]  float32 coarse boresight (coordinates.py:286-304)
 *  d_dx, d_dy     [D]   float32 detector offsets (radians)
 *  d_band         [D]   table index of each detector row, 0 <= band < n_tables
 *  d_mueller00    [D]   mueller()[:,0,0]
 *  pwv0                 weather.pwv (mm)  (atmosphere.py:309)
 *  d_pwv          [Ta*D] f64 zenith-scaled pwv, time-major; may be NULL
 *  d_loading      [Ta*D] f32 band power in pW, time-major
 *  d_flags        one uint32 device word, OR-ed with MRX_FLAG_* (never cleared
 *                 by the kernel; clear it with mrx_clear_flags)
 */
int mrx_atm_sample(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az,
                   const float* d_el, int Ta, const float* d_dx,
                   const float* d_dy, const int32_t* d_band,
                   const float* d_mueller00, int D, double pwv0, double* d_pwv,
                   float* d_loading, uint32_t* d_flags);

int mrx_clear_flags(mrx_ctx* ctx, uint32_t* d_flags);
int mrx_read_flags(mrx_ctx* ctx, const uint32_t* d_flags, uint32_t* host_flags);

/* Not-a-knot cubic spline through the coarse samples of every detector:
 * scipy interp1d(kind="cubic") == make_interp_spline(k=3) as called at
 * sim/atmosphere.py:72-82.  Knots are uniform (coordinates.py:292,
 * np.arange).  Writes for every knot the pair (y, m) with m = h^2/6 * S''(x),
 * solved in float64, so that on [x_j, x_j+1], u = (x - x_j)/h:
 *   S = (1-u) y_j + u y_j+1 + ((1-u)^3 - (1-u)) m_j + (u^3 - u) m_j+1.
 *  d_y  [Ta*D] f32 time-major;  d_ym [Ta*D][2] f32 time-major.  Ta >= 4. */
int mrx_spline_prepare(mrx_ctx* ctx, const float* d_y, int D, int Ta,
                       float* d_ym);

/* Evaluates the spline at the full-rate sample times and writes the TOD
 * (sim/atmosphere.py:72-82, cast to float32), optionally scaled per detector
 * (gain error, sim/simulation.py:239-247; pW->K_RJ, tod/tod.py:106-142).
 * Beyond the last knot the last polynomial piece is extended, which is what
 * fill_value="extrapolate" does.
 *  ta0, dta       first coarse time and coarse step (s)
 *  d_t     [T]    f64 full-rate sample times, ascending
 *  d_scale [D]    f32 per-detector factor, or NULL
 *  d_rows  [D]    destination row of each detector, or NULL for the identity:
 *                 a caller that keeps its detectors in a locality order (see
 *                 maria_amd/pipeline.py) still gets the TOD in its own row order
 *  d_out          f32, element (d, s) at d_out[d * ld_out + s] */
int mrx_spline_upsample(mrx_ctx* ctx, const float* d_ym, int D, int Ta,
                        double ta0, double dta, const double* d_t, int T,
                        const float* d_scale, const int32_t* d_rows,
                        float* d_out, size_t ld_out);

/* mrx_spline_prepare + mrx_spline_upsample in ONE kernel (sim/atmosphere.py:72-82): the
 * writer reads the raw coarse samples d_y [Ta*D] f32 time-major -- mrx_atm_sample's
 * d_loading as it is -- and solves the second derivatives of the knots its own tile of
 * 1024 samples touches in the tile prologue (the same twisted factorisation, started
 * 16 knots outside the tile or at the true not-a-knot end).  No (y, m) buffer, no solve
 * launch.  Same result as the two-call form to float32 rounding of m (<= 1e-7 of y).
 * Arguments as mrx_spline_upsample.  A tile whose samples span more knots than the LDS
 * image holds (upsampling ratios T/Ta below ~4) is walked in segments: correct at any
 * ratio, fastest from ~18 up (maria's ratios: 5 at 50 Hz, 40 at 400 Hz). */
int mrx_spline_upsample_fused(mrx_ctx* ctx, const float* d_y, int D, int Ta,
                              double ta0, double dta, const double* d_t, int T,
                              const float* d_scale, const int32_t* d_rows,
                              float* d_out, size_t ld_out);

/* Atmosphere -> TOD for one observation in ONE launch: mrx_atm_sample followed by
 * mrx_spline_upsample_fused -- the reference's _simulate_atmosphere from the layer loop to the
 * interpolation at the sample rate (atmosphere/atmosphere.py:317-373, sim/atmosphere.py:43-82) --,
 * with the hand-over from the sampling to the writing on the device, TIME CHUNK by time chunk: one
 * resident grid takes sampler work items (a time chunk of 256 detectors) and TOD tiles (1024 samples
 * of 32 rows) from two queues; a tile is written once the two or three chunks around its samples are
 * in (a counter per chunk; write-through stores, an sc1 poll, a workgroup barrier, sc1 loads), and a
 * workgroup whose tile is not ready samples one work item instead of waiting, so the two kinds of
 * work balance themselves and the launch cannot stall whatever is resident.  The first TOD tile
 * starts after the first chunks instead of after a whole sampling launch, all detectors of a chunk
 * share one pass over the screens' footprint, and no launch boundary or stream event separates
 * anything.  Bit-identical to the two calls.
 *  block_rows        detectors per block of the coarse array, rounded up to a multiple of 256; <= 0:
 *                    the library's choice (mrx_atm_synthesize_block_rows tells) -- blocks of about 5 000
 *                    rows while the plan's screens fit the Infinity Cache, one block beyond (every block
 *                    walks the screens' track again), never more than 4 * Ta * rows < 2^31 allows (a block's
 *                    coarse array is addressed with 32-bit offsets)
 *  sampler_wgs       workgroups that ONLY sample while work items remain (they write afterwards);
 *                    <= 0: MRX_OPT_SAMPLE_WGS_PER_CU per CU (unset: 2).  The others write and sample
 *                    where they would wait
 *  d_coarse          f32, out (scratch the caller may read), 128-byte aligned, Ta * round_up(D, 32)
 *                    floats: the coarse loading in pW, block b (rows b*block_rows ...) as its own
 *                    time-major array at d_coarse + Ta * b * block_rows, [Ta][pitch] with
 *                    pitch = its rows rounded up to 32 (the columns past the last row: padding)
 *  d_pwv             f64, out, or NULL: the zenith-scaled pwv of every coarse sample (what mrx_atm_sample's
 *                    d_pwv receives: atmosphere.py:373; the map mixin's calibration reads it,
 *                    sim/map.py:117-135), block b as its own time-major array [Ta][rows of b] at
 *                    d_pwv + Ta * b * block_rows -- one [Ta][D] array where the rows are one block
 *  other arguments   as mrx_atm_sample (d_az ... pwv0, d_flags) and mrx_spline_upsample
 *                    (ta0 ... ld_out); MRX_OPT_SAMPLE_CHUNK: coarse steps per time chunk (unset: by size)
 * MRX_ERR_UNSUPPORTED (nothing launched) for plans or options the pixel-coordinate sampler
 * does not take -- a layer on a non-uniform axis, MRX_OPT_AXIS_LITERAL, MRX_OPT_POINTING_CHAIN,
 * bicubic band tables --: use the two calls there.  MRX_FLAG_HANDOVER in d_flags: see above. */
int mrx_atm_synthesize(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az, const float* d_el,
                       int Ta, const float* d_dx, const float* d_dy, const int32_t* d_band,
                       const float* d_mueller00, int D, double pwv0, float* d_coarse, int block_rows,
                       int sampler_wgs, uint32_t* d_flags, double ta0, double dta, const double* d_t, int T,
                       const float* d_scale, const int32_t* d_rows, float* d_out, size_t ld_out, double* d_pwv);

/* The rows per block mrx_atm_synthesize lays d_coarse (and d_pwv) out in for `block_rows` (<= 0: the library's
 * choice, which depends on the plan's screens: see the call) -- for a caller that reads the coarse arrays back. */
int mrx_atm_synthesize_block_rows(mrx_ctx* ctx, const mrx_atm_plan* plan, int D, int Ta, int block_rows, int* rows_out);

/* mrx_atm_synthesize with TOD.to("K_RJ") applied on the coarse grid (tod/tod.py:106-142 before the spline, as
 * mrx_coarse_to_krj does between the two calls -- same functions, same operands, same bits): the sampler role divides
 * every coarse sample by den_band(d)(el_det(d, j)) before it stores it, and the writer role then writes K_RJ at the
 * pW writer's cost.  Use it where mrx_coarse_to_krj applies (the caller bounds the form's error:
 * DevicePath.coarse_krj_bound).
 *  d_cal_dx, d_cal_dy [D]   detector offsets as the calibration holds them (mrx_coarse_to_krj's d_dx, d_dy)
 *  d_cal_axis_el [n_el], d_cal_values [n_bands][n_el], n_el, n_bands   as mrx_coarse_to_krj; the cell table
 *                           (16 (n_el - 1) n_bands bytes) must fit the launch's LDS beside the sampler's:
 *                           MRX_ERR_UNSUPPORTED otherwise
 *  d_tail_pw, tail_knots, ld_tail   as mrx_coarse_to_krj_keep_tail: the last tail_knots knots of every detector IN
 *                           pW, element (k, d) at d_tail_pw[k * ld_tail + d] (d: the row in this call's D), for the
 *                           samples past the last knot, which take the per-sample form; NULL / 0: not kept
 *  T                        the samples the writer role writes: those up to the last knot when a tail is kept
 *  d_coarse                 holds K_RJ afterwards */
int mrx_atm_synthesize_krj(mrx_ctx* ctx, const mrx_atm_plan* plan, const float* d_az, const float* d_el,
                           int Ta, const float* d_dx, const float* d_dy, const int32_t* d_band,
                           const float* d_mueller00, int D, double pwv0, float* d_coarse, int block_rows,
                           int sampler_wgs, uint32_t* d_flags, double ta0, double dta, const double* d_t, int T,
                           const float* d_scale, const int32_t* d_rows, float* d_out, size_t ld_out,
                           const float* d_cal_dx, const float* d_cal_dy, const float* d_cal_axis_el,
                           const float* d_cal_values, int n_el, int n_bands, float* d_tail_pw, int tail_knots,
                           size_t ld_tail, double* d_pwv);

/* mrx_spline_upsample fused with TOD.to("K_RJ") (tod/tod.py:106-142): each sample
 * is divided by den_b(el) = (0.5 if polarized else 1) * k_B * 1e12 *
 * Int tau_b(nu) exp(-opacity(nu)) dnu (calibration/functions.py:73-90,
 * band/band.py:235-255), looked up on the elevation axis at the detector's own
 * full-rate elevation, which the kernel recomputes from the full-rate boresight
 * elevation and the (rolled) detector offsets (sim/observation.py:55-58,
 * coords/transforms.py:14-28, float32).  The host collapses each band's
 * transmission-integral table at the observation's scalar base temperature and
 * zenith pwv (tod.py:94-97) onto the elevation axis:
 *  d_bore_el     [T]   float32 full-rate boresight elevation
 *  d_dx, d_dy    [D]   float32 offsets of observation.coords (rolled), radians
 *  d_band        [D]   band index of each detector row
 *  d_cal_axis_el [n_el] float32 elevation axis (rad); d_cal_values [n_bands][n_el]
 * A detector elevation outside the axis gives NaN, as jax's interpolator does. */
int mrx_spline_upsample_krj(mrx_ctx* ctx, const float* d_ym, int D, int Ta,
                            double ta0, double dta, const double* d_t, int T,
                            const float* d_scale, const int32_t* d_rows,
                            const float* d_bore_el, const float* d_dx,
                            const float* d_dy, const int32_t* d_band,
                            const float* d_cal_axis_el,
                            const float* d_cal_values, int n_el, int n_bands,
                            float* d_out, size_t ld_out);

/* TOD.to("K_RJ") applied to the COARSE loading before the spline: d_out[j * D + d] =
 * d_loading[j * D + d] / den_band(d)(el_det(d, j)), time-major like mrx_atm_sample's output,
 * with the detector elevation of coarse step j from the coarse boresight elevation (the same
 * float32 formula as mrx_spline_upsample_krj).  mrx_spline_prepare + mrx_spline_upsample of
 * d_out then give the K_RJ TOD at the pW writer's cost (HBM-bound instead of arithmetic-bound).
 * It is S[y / g] in place of the reference's S[y] / g (S the spline, g = den(el_det(t))): the
 * two differ by the spline's interpolation error on g alone -- g is smooth in time but for the
 * kinks where a detector's elevation crosses a node of the table's axis, where the error is at
 * most 0.25 x (jump of the relative slope of den at the node) x (elevation step per coarse
 * sample).  The caller bounds that on the host and takes mrx_spline_upsample_krj when it is
 * not far below the tolerance (maria_amd/pipeline.py: DevicePath.coarse_krj_bound).
 * d_out may be d_loading (conversion in place). */
int mrx_coarse_to_krj(mrx_ctx* ctx, const float* d_loading, int D, int Ta, const float* d_bore_el_coarse,
                      const float* d_dx, const float* d_dy, const int32_t* d_band,
                      const float* d_cal_axis_el, const float* d_cal_values, int n_el, int n_bands,
                      float* d_out);

/* The same, and the loading in pW of the last `tail_knots` coarse steps kept aside as it is read:
 * d_tail_pw[k * ld_tail + d] = d_loading[(Ta - tail_knots + k) * D + d].  The samples past the last
 * coarse knot are not written in the coarse form (an extrapolated spline of y / g is not the
 * reference's extrapolated spline of y, divided; maria/sim/atmosphere.py:72-82 extrapolates): the caller
 * converts them one by one from these knots (mrx_spline_prepare + mrx_spline_upsample_krj on the window),
 * and d_out may be d_loading.  d_tail_pw = NULL or tail_knots = 0: mrx_coarse_to_krj. */
int mrx_coarse_to_krj_keep_tail(mrx_ctx* ctx, const float* d_loading, int D, int Ta, const float* d_bore_el_coarse,
                                const float* d_dx, const float* d_dy, const int32_t* d_band,
                                const float* d_cal_axis_el, const float* d_cal_values, int n_el, int n_bands,
                                float* d_out, float* d_tail_pw, int tail_knots, size_t ld_tail);

/* TOD.to("K_RJ") of a field that is already at the full rate -- the noise (and later map /
 * cmb) fields, tod/tod.py:106-142 -- in place: d_data[row(d) * ld + s] *= d_scale[d] /
 * den_band(d)(el(d, s)), with the arguments of mrx_spline_upsample_krj. */
int mrx_tod_to_krj(mrx_ctx* ctx, float* d_data, size_t ld, int D, int T,
                   const float* d_scale, const int32_t* d_rows,
                   const float* d_bore_el, const float* d_dx, const float* d_dy,
                   const int32_t* d_band, const float* d_cal_axis_el,
                   const float* d_cal_values, int n_el, int n_bands);

/* The way back, TOD.to("pW") of a K_RJ field in place: d_data *= d_scale[d] * den(el(d, s));
 * same arguments.  (tests/noise/test_noise.py:14 converts the default-unit TOD this way.) */
int mrx_tod_from_krj(mrx_ctx* ctx, float* d_data, size_t ld, int D, int T,
                     const float* d_scale, const int32_t* d_rows,
                     const float* d_bore_el, const float* d_dx, const float* d_dy,
                     const int32_t* d_band, const float* d_cal_axis_el,
                     const float* d_cal_values, int n_el, int n_bands);

/* Full-rate detector pointing: Coordinates.broadcast at the sample rate
 * (coords/coordinates.py:378-386 via transforms.py:10-29, float32;
 * sim/observation.py:55-58).  d_az, d_el [T] float32 boresight; d_dx, d_dy [D];
 * outputs [D][ld_out] float32 azimuth and elevation (radians). */
int mrx_pointing_broadcast(mrx_ctx* ctx, const float* d_az, const float* d_el,
                           int T, const float* d_dx, const float* d_dy, int D,
                           float* d_az_out, float* d_el_out, size_t ld_out);

/* Linear upsample of the coarse pwv to the full rate
 * (sim/atmosphere.py:30-37, interp1d linear + extrapolate); only the map/cmb
 * mixins consume it.  d_pwv [Ta*D] f64 time-major -> d_out [D][ld_out] f32. */
int mrx_linear_upsample(mrx_ctx* ctx, const double* d_pwv, int D, int Ta,
                        double ta0, double dta, const double* d_t, int T,
                        float* d_out, size_t ld_out);

/* ---- screens -------------------------------------------------------------- */

/* Separable Gaussian filter with scipy.ndimage.gaussian_filter semantics
 * (order 0, mode="reflect", radius int(truncate*sigma+0.5), axis 0 then axis 1,
 * float64 accumulation, float32 storage between the passes).  Serves the
 * per-layer screen smoothing (atmosphere/atmosphere.py:341-344) and
 * ProjectionMap.smooth (map/projection.py:485-504).  A sigma <= 1e-15 skips
 * that axis as scipy does.  d_tmp is ny*nx floats of scratch; in == out is
 * allowed. */
int mrx_gauss_smooth2d(mrx_ctx* ctx, const float* d_in, float* d_out,
                       float* d_tmp, int ny, int nx, double sigma_y,
                       double sigma_x, double truncate);

/* Linear resampling of every row of a screen onto new column positions:
 *   d_out[e * ld_out + j] = d_scale[j] * ((1 - d_w[j]) * d_in[e * ld_in + d_idx[j]] + d_w[j] * d_in[e * ld_in + d_idx[j] + 1]).
 * model="3d": every layer of the one process has its own cross-section grid, coarser with height
 * (atmosphere/atmosphere.py:208-219, extrusion.py:20-22), while the layers are planes of one volume
 * generated on one grid; the host gives the bracketing column, the weight and a variance factor
 * (1 / sqrt((1-w)^2 + w^2 + 2 w (1-w) rho(step)): a linear blend of two unit-variance samples has
 * less than unit variance) per output column.  0 <= d_idx[j] <= n_in - 2. */
int mrx_resample_columns(mrx_ctx* ctx, const float* d_in, int n_e, int n_in, size_t ld_in, const int32_t* d_idx,
                         const float* d_w, const float* d_scale, int n_out, float* d_out, size_t ld_out);

/* ProjectionMap.smooth (map/projection.py:485-504): numer = G(data*weight),
 * denom = G(weight), out = denom > 0 ? numer/denom : 0; d_weight may be NULL
 * (weight == 1).  d_denom_out may be NULL.  d_tmp: 2*ny*nx floats. */
int mrx_map_smooth(mrx_ctx* ctx, const float* d_data, const float* d_weight,
                   float* d_out, float* d_denom_out, float* d_tmp, int ny,
                   int nx, double sigma_y, double sigma_x);

/* Turbulent screen generator: Philox-4x32-10 normals on the Hermitian half of k
 * space, times the square root of the Matern / von Karman spectrum
 *     PSD(k) ~ (k0^2 + |k|^2)^-(nu + 1),  k0 = sqrt(2 nu)/r0
 * (functions/__init__.py:30-39 is the covariance this is the transform of),
 * then a 2-D complex-to-real inverse FFT scaled to unit variance.  Replaces the
 * autoregressive generator (atmosphere/process.py:191-209) with a different
 * algorithm of the same target covariance; parity is statistical.
 * Spectrum cell (ky, kx), 0 <= kx <= nx/2, is amp(k) (a + i b)/sqrt 2 with (a, b) the
 * Box-Muller pair of Philox words (0,1) [ky < ny/2] or (2,3) [ky >= ny/2] of
 * counter (kx, ky mod ny/2, stream, 0), key = seed; the columns kx = 0 and nx/2 are
 * made Hermitian in ky from their cells ky < ny/2 (cells ky = 0, ny/2: amp a).
 * ny, nx powers of two in [64, 8192].
 *
 * mrx_screen_generate_batch makes n_screens screens that share one FFT domain (the
 * layers of an atmosphere) in two launches, with the beam smoothing of
 * atmosphere/atmosphere.py:341-344 (scipy.ndimage.gaussian_filter: reflect, truncate 4)
 * folded in: along y on the half spectra between the two FFT passes, along x on the
 * finished row; float32 accumulation (equal to mrx_gauss_smooth2d of the unsmoothed
 * screen to ~1e-6 of its rms) -- or, per screen (periodic_beam), as a factor of the spectrum.  Only the top-left out_ny x out_nx block of the
 * periodic domain is smoothed and written (reflection at ITS edges): a layer may use a
 * larger FFT domain than its grid so that opposite edges decorrelate. */
typedef struct mrx_screen_desc {
  float* d_out;            /* [out_ny][ld_out] f32                                  */
  uint32_t stream;         /* Philox stream of this screen (the layer index)        */
  int32_t out_ny, out_nx;  /* written block; 0 = the whole domain                   */
  int32_t periodic_beam;   /* 0: the beam as scipy applies it to the written block (reflection at its
                              edges).  1: the same truncated, normalised taps as a convolution on the
                              PERIODIC FFT domain, folded into the spectrum (the two stencils, a quarter
                              of the generator's arithmetic, disappear): pixels at least the stencil
                              radius int(4 sigma + 0.5) from every edge of the written block are the
                              same to rounding, nearer ones see the field beyond the edge instead of its
                              mirror image.  For callers whose lines of sight keep that distance.      */
  size_t ld_out;           /* row pitch in floats; 0 = out_nx                        */
  double dy, dx;           /* grid steps (m)                                         */
  double r0, nu;           /* outer scale (m), Matern smoothness                     */
  double sigma_y, sigma_x; /* beam sigma in pixels along y / x; 0 = no smoothing     */
  const float* d_amp;      /* amplitude table of mrx_screen_amplitudes for this FFT domain
                              (dy, dx, r0, nu are then ignored), or NULL: the power law */
} mrx_screen_desc;
/* floats of scratch for n_screens screens: 2 * n_screens * (nx/2 + 1) * (ny + 16) */
int mrx_screen_work_floats(int ny, int nx, int n_screens, size_t* floats);
int mrx_screen_generate_batch(mrx_ctx* ctx, uint64_t seed, int ny, int nx,
                              const mrx_screen_desc* screens, int n_screens,
                              float* d_work, size_t work_floats);
/* model="3d" (atmosphere/atmosphere.py:28,141-279, extrusion.py:69-77): ONE process holds many
 * layers whose turbulence is correlated vertically -- a Matern(nu = 1/3, r0) field in three
 * dimensions, PSD(k) ~ (k0^2 + |k|^2)^-(nu + 3/2), sampled on the layers' heights.  Three transform
 * passes on a periodic nh x ny x nx domain (powers of two; nh in [8, 2048]): along h per (ky, kx)
 * cell, keeping only the requested height planes (linear interpolation between the two FFT planes
 * around plane_pos[p], in units of dh from plane 0, times plane_scale[p] -- the caller's variance
 * correction 1/sqrt((1-w)^2 + w^2 + 2 w (1-w) rho(dh)), or NULL); then the two passes of
 * mrx_screen_generate_batch per plane, with the beam smoothing folded in.  `planes[p]` gives each
 * plane's output block and beam (d_out, out_ny, out_nx, ld_out, sigma_y, sigma_x; the other fields are
 * ignored).  Cell (kz, ky, kx) uses Philox counter (kx, ky, stream << 16 | kz mod nh/2, 0x33440000). */
int mrx_screen3d_work_floats(int nh, int ny, int nx, int n_planes, size_t* floats);
int mrx_screen_generate_3d(mrx_ctx* ctx, uint64_t seed, uint32_t stream, int nh, int ny, int nx,
                           double dh, double dy, double dx, double r0, double nu,
                           const double* plane_pos, const double* plane_scale,
                           const mrx_screen_desc* planes, int n_planes, float* d_work,
                           size_t work_floats, const float* d_amp);

/* Covariance-matched amplitudes (circulant embedding) for the generators above.  The power law is
 * the continuous transform of the Matern covariance cut at the grid's Nyquist wavenumber; for rough
 * fields (nu = 1/3 in three dimensions, atmosphere/atmosphere.py:249) most of the one-pixel structure
 * lies beyond it.  This table holds sqrt(max(lambda, 0)) with lambda the eigenvalues of the
 * covariance ITSELF on the periodic nh x ny x nx grid (nh = 0: two dimensions),
 *     lambda[k] = sum_d rho_per(d) cos(2 pi k.d / N),   rho_per(d) = sum over images n of rho(|d + n L| / r0)
 * (images summed, not the distance wrapped: the periodic covariance stays positive definite; images
 * beyond x_cut outer scales -- where rho < 1e-10 -- are skipped), so that the structure function of the
 * screens is Matern's from one pixel up.  The header of the table normalises the screens by
 * sum(lambda) / rho_per(0): their variance is rho_per(0) = 1 + the images' share (1.01 at L = 5 r0) and
 * their structure function 2 (rho_per(0) - rho_per(d)), in which the images cancel to second order.
 * rho is the exact Matern correlation (functions/__init__.py:30-39) read from the caller's HOST tables
 * log_cov[i] = log rho(x_i), log_sf[i] = log(1 - rho(x_i)) at log x_i = log_first + i log_step, x = r / r0
 * (8192 nodes from 1e-6 to 1e3: four-point Lagrange interpolation in log x is then good to 1e-11; finite
 * values only): rho(0) = 1 and, with x = max(|r| / r0, x_first), t = 1/(1 + x^2), rho = t (1 - exp(LSF(x))) +
 * (1 - t) exp(LCOV(x)), the blend of functions/__init__.py:70-73.  (The reference's
 * approximate_normalized_matern itself -- 1024 nodes, linear, good to 1e-5 -- is no longer positive
 * definite on a grid at that accuracy: its clipped eigenvalues come back as 50 % too much structure at
 * one pixel.)  float64 on the device, direct
 * cosine sums over the even half axes: a set-up step, once per geometry (8 x 2048^2 layers of two
 * outer scales: milliseconds; a 64 x 4096 x 512 volume: about a second).  d_table
 * (mrx_screen_amp_floats' table_floats; 16-byte aligned) is then passed as mrx_screen_desc.d_amp
 * (nh = 0) or mrx_screen_generate_3d's d_amp. */
int mrx_screen_amp_floats(int nh, int ny, int nx, int n_radial, size_t* table_floats, size_t* work_floats);
int mrx_screen_amplitudes(mrx_ctx* ctx, int nh, int ny, int nx, double dh, double dy, double dx, double r0,
                          const double* log_cov, const double* log_sf, int n_radial, double log_first,
                          double log_step, double x_cut, float* d_table, float* d_work, size_t work_floats);

/* One unsmoothed screen over the whole domain.  d_work: mrx_screen_work_floats(ny, nx, 1)
 * floats (a buffer of 2*ny*nx float2, the size earlier versions asked for, is ample). */
int mrx_screen_generate(mrx_ctx* ctx, uint64_t seed, uint32_t stream, int ny,
                        int nx, double dy, double dx, double r0, double nu,
                        float* d_out, float* d_work);

/* Sum over the FFT grid of the un-normalised PSD, reduced on the device
 * (float64); mrx_screen_generate uses it internally, exposed for tests. */
int mrx_screen_psd_sum(mrx_ctx* ctx, int ny, int nx, double dy, double dx,
                       double r0, double nu, double* host_sum);

/* ---- multi-GPU --------------------------------------------------------- */

/* Detectors shard across the GPUs of a node in equal blocks of contiguous rows (the last may
 * be short); the data path itself needs no collective.  The epilogue the reference's seam
 * implies -- one [ndet, nt] array per observation, sim/simulation.py:266-272 -- is ONE RCCL
 * all-gather over xGMI: the TOD is detector-major, so the gathered array is the concatenation
 * of the shards and nothing is packed or staged.
 *
 * mrx_comm_unique_id (rank 0; the caller hands the MRX_COMM_ID_BYTES bytes to every rank by
 * whatever channel it has: torch.distributed, MPI, a file) + mrx_comm_create = ncclGetUniqueId +
 * ncclCommInitRank on the context's device; mrx_comm_wrap adopts an ncclComm_t the caller
 * already owns.  RCCL is loaded at first use (dlopen), not at library load. */
#define MRX_COMM_ID_BYTES 128
int mrx_comm_unique_id(mrx_ctx* ctx, void* id_out);
int mrx_comm_create(mrx_ctx* ctx, const void* id, int world, int rank, mrx_comm** comm);
int mrx_comm_wrap(mrx_ctx* ctx, void* nccl_comm, int world, int rank, mrx_comm** comm);
int mrx_comm_destroy(mrx_ctx* ctx, mrx_comm* comm);
/* d_full[r * count + i] = rank r's d_shard[i], for every rank, enqueued on the context's
 * stream.  count = rows_per_rank * ld floats: every rank passes the same count (pad the last
 * shard's rows).  In place when d_shard == d_full + rank * count, i.e. when the writer put the
 * shard straight into its slot of the full TOD (mrx_spline_upsample's d_out / ld_out). */
int mrx_allgather_tod(mrx_ctx* ctx, mrx_comm* comm, const float* d_shard, float* d_full,
                      size_t count);

/* The same gather as world - 1 direct sends + world - 1 receives in one RCCL group instead of
 * one ncclAllGather: every pair of the node's GPUs has its own xGMI link, so each rank pushes
 * its shard to all peers at once (shard / link rate, against (world - 1) x that around a ring;
 * SURVEY section 5).  Same arguments and result as mrx_allgather_tod; which of the two is
 * faster on a given node is for the caller to measure (bench.py --gather-algo). */
int mrx_allgather_tod_p2p(mrx_ctx* ctx, mrx_comm* comm, const float* d_shard, float* d_full,
                          size_t count);

/* Layer-sharded screen generation (strong scaling: the screens are replicated work): rank
 * l % world generated layer l (mrx_screen_generate_batch with its own layers only); one
 * ncclBroadcast per layer from its owner, all in one group, fills the other ranks' buffers in
 * place.  Screens are functions of (seed, layer), so the result equals what every rank would
 * have generated itself (atmosphere/process.py:191-209 has no counterpart: the reference is
 * one process).  d_screens [n_layers] device pointers (host array), counts [n_layers] floats. */
int mrx_exchange_screens(mrx_ctx* ctx, mrx_comm* comm, float* const* d_screens, const size_t* counts,
                         int n_layers);

/* ---- map sampling --------------------------------------------- */

/* A celestial map as MapMixin._sample_maps sees it after smoothing, conversion to K_RJ and
 * the parity flip (sim/map.py:72-73,104-109): */
typedef struct mrx_sky_map {
  const float* d_values;  /* [n_channels][n_stokes][n_eta][n_xi] float32, K_RJ */
  int n_channels, n_stokes, n_eta, n_xi;
  double eta0, deta;      /* eta[i] = eta0 + i * deta, rad (np.linspace, projection.py:122-123;
                             deta < 0 after the parity flip) */
  double xi0, dxi;        /* xi[j] = xi0 + j * dxi */
  double center_phi, center_theta; /* map centre in the map's frame, rad */
  int bilinear;           /* 1: bilinear sampling, 0: nearest pixel (map_kwargs) */
  int reserved;
} mrx_sky_map;

/* K_RJ -> pW of each channel (sim/map.py:117-135, band/band.py:235-255): with an atmosphere
 * the channel's transmission integral, collapsed by the host at the scalar base temperature
 * onto (zenith pwv, elevation), is looked up per sample at the detector's elevation and its
 * zenith-scaled pwv -- the coarse series of mrx_atm_sample, interpolated linearly
 * (sim/atmosphere.py:30-37); without one, a scalar per channel. */
typedef struct mrx_map_cal {
  const float* d_table;    /* [n_channels][n_pwv][n_el] float32, or NULL */
  const float* d_axis_pwv; /* [n_pwv] */
  const float* d_axis_el;  /* [n_el] */
  int n_pwv, n_el;
  const double* d_pwv;     /* [Ta][D] coarse zenith-scaled pwv, time-major */
  int Ta;
  int steps_per_tile;      /* the most coarse steps of the pwv series that 1024 consecutive samples meet (ceil(1025 dt / dta) + 1), or 0 if the
                            * caller does not know: sizes the LDS table of the calibration's interval form (a tile that meets more
                            * takes the per-sample form: correct, slower) */
  double ta0, dta;         /* first coarse time and coarse step (s) */
  const double* d_t;       /* [T] full-rate sample times */
  const double* d_scalar;  /* [n_channels] Int passband dnu (device), used when d_table is NULL */
} mrx_map_cal;

/* obs.loading["map"] for the D detectors of one band (sim/map.py:76-172): detector pointing
 * from the boresight and offsets (float32 chain of coords/transforms.py:10-29), rotation
 * into the map's frame by the per-sample 3x3 (coords/coordinates.py:184-236; NULL for a map
 * in the az/el frame), offsets from the map centre (transforms.py:36-53), the pointing-matrix
 * row of utils/linalg.py:9-58 with the Stokes weights of map/projection.py:134-179, K_RJ -> pW
 * per channel, float32 accumulation over channels, and the [0.25, 0.5, 0.25] convolution
 * along time (scipy reflect mode, map.py:170), in one pass.
 *  d_az, d_el   [T] float32 full-rate boresight
 *  d_transform  [T][3][3] float64 transform stack (row vector times matrix), or NULL
 *  d_dx, d_dy   [D] float32 offsets of observation.coords (rolled), radians
 *  d_stokes_w   [D][n_stokes] float64: Mueller[d, 0, stokes] (array/array.py:204-221)
 *  d_out        [D][ld_out] float32, pW
 * The map (all channels and Stokes planes) must be smaller than 2 GiB: the sampler reads it through a copy with its
 * row pairs interleaved -- a cell's four corners in 16 contiguous bytes, one gather a sample and plane -- which every call
 * builds into memory the context owns (twice the map's bytes, kept between calls; calls on one context are ordered).
 * With MRX_OPT_POINTING_CHAIN the planes are read as they are (below 4 GiB). */
int mrx_map_sample(mrx_ctx* ctx, const mrx_sky_map* map, const mrx_map_cal* cal,
                   const float* d_az, const float* d_el, int T, const double* d_transform,
                   const float* d_dx, const float* d_dy, const double* d_stokes_w, int D,
                   float* d_out, size_t ld_out);

/* mrx_map_sample with the gain and TOD.to("K_RJ") (tod/tod.py:106-142) on its store:
 *   d_out[d][s] = S(d, s) * d_scale[d] / den_band(d)(el(d, s)),
 * S the pW field mrx_map_sample writes, the division exactly as mrx_tod_to_krj performs it on a finished field (the same
 * per-tile elevation model from the same 1024 samples, the same lookup) -- the same bits as mrx_map_sample followed by
 * mrx_tod_to_krj with that d_scale, without the second pass over the field.
 *  d_scale                      [D] float32 or NULL (1): the gain error (sim/simulation.py:239-247)
 *  d_bore_el                    [T] float32 full-rate boresight elevation
 *  d_krj_dx, d_krj_dy, d_band   [D] float32 offsets and int32 band index of this call's rows, as mrx_tod_to_krj takes them
 *  d_cal_axis_el, d_cal_values  [n_el], [n_bands][n_el] float32: the bands' collapsed transmission integrals
 * MRX_ERR_UNSUPPORTED (nothing launched) with MRX_OPT_POINTING_CHAIN: sample in pW and convert there. */
int mrx_map_sample_krj(mrx_ctx* ctx, const mrx_sky_map* map, const mrx_map_cal* cal,
                       const float* d_az, const float* d_el, int T, const double* d_transform,
                       const float* d_dx, const float* d_dy, const double* d_stokes_w, int D,
                       const float* d_scale, const float* d_bore_el, const float* d_krj_dx, const float* d_krj_dy,
                       const int32_t* d_band, const float* d_cal_axis_el, const float* d_cal_values, int n_el, int n_bands,
                       float* d_out, size_t ld_out);

/* BinMapper.run for one TOD (mappers/bin_mapper.py:84-120): the transpose of the pointing
 * matrix of mrx_map_sample, map_sum += (W * D) @ P and map_wgt += W @ |P|, as float64 atomic
 * adds into the caller's (zeroed or running) maps; the map itself is sum / wgt.  `map` gives
 * the grid (d_values is not read; bilinear = 0 is the mapper's default).
 *  d_tod     [D][ld_tod] float32 signal (the sum of the TOD's fields)
 *  d_weight  [D][ld_weight] float32 sample weights, or NULL for ones (tod.weight's default)
 *  d_channel [D] map channel of each detector (the nu plane whose frequency is the
 *            detector's band centre, map/projection.py:152-155), or NULL for 0
 *  d_sum, d_wgt  [n_stokes][n_channels][n_eta][n_xi] float64
 * Pointing arguments as mrx_map_sample.  Summation order is not fixed (atomics): results
 * agree with a serial sum to float64 rounding. */
int mrx_bin_map(mrx_ctx* ctx, const mrx_sky_map* map, const float* d_tod, size_t ld_tod,
                const float* d_weight, size_t ld_weight, const float* d_az, const float* d_el, int T,
                const double* d_transform, const float* d_dx, const float* d_dy,
                const double* d_stokes_w, const int32_t* d_channel, int D, double* d_sum,
                double* d_wgt);

/* The same without global atomics on scattered pixels -- they execute at the memory side at a
 * tenth of the contiguous rate, and the focal plane is sparser than the map, so nothing merges on
 * the way.  The samples are routed to the map instead: every sample's pixel(s) are computed
 * once, a tile's contributions (nearest pixel: 16 detectors x 1024 samples, one each; bilinear:
 * 8 x 256, up to four each) are sorted by map region (64 x 32 pixels of one plane) into the
 * tile's slot of the work buffer, and each region's segments are then summed in LDS (float64)
 * and added to the map as whole rows.  Same arguments and same result to float64 rounding (the
 * order of the sums differs).  MRX_ERR_UNSUPPORTED for maps of more than 2048 regions
 * (n_channels * ceil(n_eta / 32) * ceil(n_xi / 64)): call mrx_bin_map.
 *  d_work   16-byte aligned; mrx_bin_map_work_bytes gives the least size (one column of tiles:
 *           all detectors x one tile of samples) and the size that takes all T samples in one
 *           go (16 bytes per contribution: 16 per sample nearest, 64 bilinear, + 4 bytes per
 *           (region, tile)); with less the call walks the time axis in chunks. */
int mrx_bin_map_work_bytes(const mrx_sky_map* map, int D, int T, size_t* min_bytes, size_t* full_bytes);
int mrx_bin_map_bucketed(mrx_ctx* ctx, const mrx_sky_map* map, const float* d_tod, size_t ld_tod,
                         const float* d_weight, size_t ld_weight, const float* d_az, const float* d_el, int T,
                         const double* d_transform, const float* d_dx, const float* d_dy,
                         const double* d_stokes_w, const int32_t* d_channel, int D, double* d_sum,
                         double* d_wgt, void* d_work, size_t work_bytes);

/* ---- TOD pre-processing for the mappers (tod/processing.py:91-204) --------------------- */

/* remove_slope (D -= linspace(D[:, 0], D[:, -1], T), processing.py:99-105) and / or window
 * (D *= w, processing.py:139-146) on a [D][ld] float32 TOD in place, each step rounded to
 * float32 as the reference's in-place numpy operations do.
 *  d_window [T] float64 window (scipy.signal.windows.*, host) or NULL
 *  d_work   2 * D doubles of scratch (only read when remove_slope is set) */
int mrx_tod_detrend_window(mrx_ctx* ctx, float* d_data, size_t ld, int D, int T, int remove_slope,
                           const double* d_window, double* d_work);

/* scipy.signal.sosfilt along time (utils/signal/filters.py:46-69; zero initial state,
 * transposed direct form II, float64) of every row, optionally after remove_slope
 * (processing.py:151): the low-pass and the high-pass of process_tod are one cascade.  The
 * recursion is time-parallel: chunks of mrx_sosfilt_chunk() samples are run from a zero state,
 * chained through the chunk transition matrix and run again from their true initial state; the
 * result equals the serial loop up to float64 rounding.
 *  sos            [n_sections][6] host array (b0 b1 b2 a0 a1 a2), n_sections <= 8
 *  d_chunk_matrix [2 n][2 n] float64, device: the state transition of the cascade over one
 *                 chunk, state order (z0, z1) per section (maria_amd/tod_processing.py builds it)
 *  d_in, d_out    [D][ld] float32; may be the same buffer
 *  d_work         mrx_sosfilt_work_doubles(D, T, n_sections) doubles */
int mrx_sosfilt_chunk(void);
int mrx_sosfilt_work_doubles(int D, int T, int n_sections, size_t* doubles);
int mrx_sosfilt(mrx_ctx* ctx, const double* sos, int n_sections, const double* d_chunk_matrix,
                const float* d_in, size_t ld_in, int D, int T, int remove_slope, float* d_out,
                size_t ld_out, double* d_work);

/* Test hook for the in-LDS inverse FFT both generators are built on: `rows` independent rows
 * of n << interleave_log2 complex float32 values, each holding 2^interleave_log2 interleaved
 * sequences of length n (a power of two >= 4; at most 8192 values per row); unnormalised
 * (numpy.fft.ifft(x) * n).  interleave_log2 = -1: the 64-point register transform, n = 64;
 * -2: the 4096-point workgroup transform (three radix-16 register passes), n = 4096;
 * -3: its 16 x RB x 16 form for n = 1024, 2048, 4096 (16 / RB rows per workgroup). */
int mrx_fft_rows(mrx_ctx* ctx, const float* d_in, int rows, int n, int interleave_log2,
                 float* d_out);

/* ---- detector noise ----------------------------------------- */

/* White + 1/f noise with spatially correlated modes: sim/noise.py:18-63 and
 * noise/generation.py:11-51,
 *   noise[d,t] = scale_d (sqrt(fs) w[d,t] + sqrt(c) sum_m B[d,m] M_m[t] + sqrt(1-c) p_d[t]),
 * w white N(0,1); p_d independent pink series with two-sided spectrum (knee/2)/|f| (equal
 * to the white level at f = knee); M_m = sqrt(fs) w'_m + P_m the modes (white + pink, the
 * generator applied to itself, generation.py:41-43); c = corr_prop.  The reference
 * filters white noise with a length-T FFT per detector; here all three parts of a detector
 * are synthesised in the frequency domain on a power-of-two period N >= T (one draw per cell
 * of variance fs/N + (1-c) knee/|k|; a four-step FFT whose real and imaginary parts serve two
 * detectors; the modes enter as tabulated Hermitian spectra) and cut to T samples (any T of
 * the N samples of white noise of period N are independent): same spectrum, different realisation and
 * period -- statistical parity, like the screens.  As in the reference, whose period is the
 * TOD itself, the pink and correlated parts carry no power below fs/T and have zero mean over
 * the T samples (the cells below ceil(N/T) are dropped, the window mean is subtracted).  With knee = 0 the
 * output is white only and neither the basis nor the work buffer is used.
 *  det_offset   global index of row 0 (even): the draws of detector det_offset + d depend on
 *               (seed, det_offset + d) only, the modes on the seed only -- shards of one
 *               band generated on different GPUs share their modes and nothing else
 *  d_basis [D][n_modes] spatial basis (utils/linalg.py:105-126, host), or NULL with n_modes 0
 *  d_scale [D]  1e12 * NEP per detector (noise.py:62), or NULL
 *  d_loading    [D][ld_loading] float32 total optical loading (pW) or NULL: the amplitude of
 *               sample (d,t) is d_scale[d] + per_loading * d_loading[d,t], per_loading =
 *               1e12 * NEP_per_loading (noise.py:35-37); must not alias an accumulating d_out
 *  d_out        [D][ld_out] float32, written (accumulate = 0) or added to (accumulate = 1:
 *               noise straight into an existing TOD)
 *  d_work       16-byte aligned scratch of work_floats floats; mrx_noise_work_floats(T,
 *               n_modes, batch) gives the size that holds `batch` detectors in flight
 *               (4 N + 8 bytes per detector + 8 N per mode).  From 256 detectors up the batches
 *               are spread over up to four streams (the context's stream forks and joins: the call
 *               is ordered on it like any other), each with an equal share of the buffer:
 *               1024 detectors in flight is a good size (MRX_OPT_NOISE_LANES).
 * mrx_noise_period: N = n1 * n2 for T samples (T <= 2^23). */
int mrx_noise_period(int T, int* n1, int* n2);
int mrx_noise_work_floats(int T, int n_modes, int batch, size_t* floats);
int mrx_noise_generate(mrx_ctx* ctx, uint64_t seed, int D, int det_offset, int T,
                       double sample_rate, double knee, double corr_prop,
                       const float* d_basis, int n_modes, const float* d_scale,
                       const float* d_loading, size_t ld_loading, double per_loading,
                       float* d_out, size_t ld_out, int accumulate,
                       float* d_work, size_t work_floats);
/* Two-rate form (chosen by the library: fs, knee and T decide).  Where the pink part at a quarter (half) of the
 * Nyquist frequency has fallen below 2 % of the white level -- 2 rate knee / fs <= 0.02: rate 4 at 400 Hz and a knee
 * of 1 Hz -- and T >= 32768, the pink and correlated-pink parts are synthesised at fs / rate (a period, a scratch
 * round trip and an arithmetic rate times smaller), interpolated with the four-point Catmull-Rom cubic, and the white
 * parts -- the detectors' own and the modes', which reach the Nyquist frequency -- are drawn per sample as the field is
 * written.  What changes against the one-rate form: no pink power between fs / (2 rate) and fs / 2 (< 2 % of the
 * spectrum there) and the interpolation's roll-off of the pink part in the octave below (smaller still); white level,
 * modes' covariance, zero pink mean over the TOD, shard / batch independence as before.  MRX_OPT_NOISE_GENERIC bit 8
 * keeps the one-rate form.
 *
 * mrx_noise_generate_krj: the field in K_RJ (tod/tod.py:106-142): every sample divided by den_band(d)(el(d, t)) as
 * mrx_tod_to_krj does (same elevation model, same lookup), with d_bore_el [T] and d_dx, d_dy, d_band [D] indexed like
 * the rows of d_out.  In the two-rate form the division rides on the writer's store (the field is written once);
 * otherwise the call is mrx_noise_generate followed by mrx_tod_to_krj. */
int mrx_noise_generate_krj(mrx_ctx* ctx, uint64_t seed, int D, int det_offset, int T,
                           double sample_rate, double knee, double corr_prop,
                           const float* d_basis, int n_modes, const float* d_scale,
                           const float* d_loading, size_t ld_loading, double per_loading,
                           float* d_out, size_t ld_out,
                           float* d_work, size_t work_floats,
                           const float* d_bore_el, const float* d_dx, const float* d_dy, const int32_t* d_band,
                           const float* d_cal_axis_el, const float* d_cal_values, int n_el, int n_bands);

/* Philox-4x32-10 standard normals, the generator behind the screens, exposed
 * so tests can check the stream against the published known-answer vectors.
 * counter = (i, 0, stream, 0), key = seed; d_out[i] for i < n. */
int mrx_philox_normal(mrx_ctx* ctx, uint64_t seed, uint32_t stream, size_t n,
                      float* d_out);
int mrx_philox_raw(mrx_ctx* ctx, uint64_t seed, uint32_t c0, uint32_t c1,
                   uint32_t c2, uint32_t c3, uint32_t host_out[4]);

#ifdef __cplusplus
}
#endif
#endif /* MRX_H_ */
