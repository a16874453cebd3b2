// Internal declarations shared by the libmrx translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "mrx.h"

struct mrx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  int n_cu = 0;
  int lds_per_cu = 0;
  size_t hbm_bytes = 0;
  char name[128] = {0};
  char err[512] = {0};
  // scratch owned by the context (grown on demand, never in a hot call that
  // already has what it needs)
  // Gaussian taps (f64) cached per (sigma, truncate): a repeated smoothing
  // call then needs no host->device copy and no synchronisation
  static constexpr int kTapSlots = 32;
  struct TapSlot {
    double sigma = -1.0, truncate = 0.0;
    int radius = 0;
    double* d_taps = nullptr;
  } taps[kTapSlots];
  int taps_next = 0;
  // float32 taps of the fused screen smoothing, zero-padded to a fixed length.  A batch of the screen
  // generator pins the slots it has taken (pinned == ftaps_batch) so that a later miss of the SAME batch
  // cannot evict them before the batch is launched; 128 slots hold the ~2 x 60 distinct taps of a
  // model="3d" layer table without thrashing.
  static constexpr int kFTapSlots = 128;
  struct FTapSlot {
    double sigma = -1.0;
    int radius = 0;
    float* d_taps = nullptr;
    unsigned long long pinned = 0;
  } ftaps[kFTapSlots];
  int ftaps_next = 0;
  unsigned long long ftaps_batch = 0;
  // transfer functions of those taps on a periodic axis of n nodes (the beam folded into the spectrum,
  // mrx_screen_desc.periodic_beam): n/2 + 1 floats per (sigma, n); pinned per batch like the taps
  static constexpr int kFRespSlots = 64;
  struct FRespSlot {
    double sigma = -1.0;
    int n = 0, radius = 0;
    float* d_resp = nullptr;
    unsigned long long pinned = 0;
  } fresp[kFRespSlots];
  int fresp_next = 0;
  int options[MRX_OPT_COUNT] = {0};
  // map sampling: the map's row pairs interleaved (mrx_map.hip: map_pairs_kernel), rebuilt by every call from the caller's
  // planes; the event orders the rebuild behind the sampler that last read the copy, whatever stream the context has since
  float* d_map_pairs = nullptr;
  size_t map_pairs_cap = 0;
  hipEvent_t map_pairs_read = nullptr;
  uint32_t* d_bin_order = nullptr;  // routed binning: contributions per region + the regions by falling total (mrx_map.hip)
  // tile queues + per-block counters of the launches that take their work from a queue (mrx_spline.hip: mrx_synth_ctl)
  static constexpr int kSynthCtlSlots = 8;
  int* d_synth_ctl = nullptr;
  int synth_ctl_next = 0;
  // screen normalisations (sum of the PSD over the FFT grid), one device double
  // per distinct (grid, spectrum)
  static constexpr int kPsdSlots = 64;
  struct PsdKey {
    bool valid = false;
    int ny = 0, nx = 0;
    double dy = 0, dx = 0, r0 = 0, nu = 0;
    int nh = 0;     // 0: a 2-D spectrum
    double dh = 0;
  } psd[kPsdSlots];
  int psd_next = 0;
  double* d_reduce = nullptr;  // kPsdSlots doubles
  size_t reduce_cap = 0;
  // dynamic-LDS caps already raised on THIS context's device: the attribute is per
  // device, so the bookkeeping lives here and not in process-wide statics
  static constexpr int kLdsSlots = 48;
  struct LdsCap {
    const void* fn = nullptr;
    size_t bytes = 0;
  } lds_caps[kLdsSlots];
  // side streams + fork/join events of entry points that spread independent batches over
  // several streams (mrx_noise_generate), made on demand by mrx_side_streams
  static constexpr int kSideStreams = 3;
  hipStream_t side_streams[kSideStreams] = {nullptr};
  hipEvent_t side_ev[kSideStreams + 1] = {nullptr};  // [0]: fork, [1 + i]: join of side stream i
  hipStream_t side_probed = nullptr;  // the context stream the side streams were last checked against
  bool side_checked = false;
};

// the first n side streams (n <= kSideStreams) and their events exist after this returns MRX_OK,
// each on a hardware queue of its own beside the context's stream (mrx_context.hip)
int mrx_side_streams(mrx_ctx* ctx, int n);
// do kernels on streams a and b run side by side?  (HIP spreads streams round-robin over a few hardware
// queues -- four by default -- and two streams that land on one queue take turns.)  Synchronises both.
int mrx_probe_concurrent(mrx_ctx* ctx, hipStream_t a, hipStream_t b, bool* concurrent);

inline int mrx_fail(mrx_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define MRX_HIP(ctx, call)                                                   \
  do {                                                                       \
    hipError_t e__ = (call);                                                 \
    if (e__ != hipSuccess)                                                   \
      return mrx_fail((ctx), MRX_ERR_HIP, "%s failed: %s (%s:%d)", #call,    \
                      hipGetErrorString(e__), __FILE__, __LINE__);           \
  } while (0)

#define MRX_REQUIRE(ctx, cond, msg)                                          \
  do {                                                                       \
    if (!(cond)) return mrx_fail((ctx), MRX_ERR_INVALID, "%s: %s", __func__, \
                                 (msg));                                     \
  } while (0)

#define MRX_CHECK_LAUNCH(ctx)                                                \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess)                                                   \
      return mrx_fail((ctx), MRX_ERR_HIP, "kernel launch failed: %s (%s:%d)", \
                      hipGetErrorString(e__), __FILE__, __LINE__);           \
  } while (0)

// Entry points that launch or allocate run on the context's own device, whatever the
// calling thread's current device is; the caller's device is restored on return.
struct mrx_device_guard {
  int prev = -1;
  explicit mrx_device_guard(const mrx_ctx* ctx) {
    if (ctx && hipGetDevice(&prev) == hipSuccess && prev != ctx->device)
      (void)hipSetDevice(ctx->device);
    else
      prev = -1;
  }
  ~mrx_device_guard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define MRX_ENTER(ctx) mrx_device_guard mrx_guard__(ctx)

// Raise kernel `fn`'s dynamic-LDS cap to at least `bytes` on the context's device, once.
inline int mrx_lds_cap(mrx_ctx* ctx, const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return MRX_OK;  // the default cap
  mrx_ctx::LdsCap* slot = nullptr;
  for (auto& c : ctx->lds_caps) {
    if (c.fn == fn) { slot = &c; break; }
    if (!c.fn && !slot) slot = &c;
  }
  if (slot && slot->fn == fn && slot->bytes >= bytes) return MRX_OK;
  MRX_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (slot) {
    slot->fn = fn;
    slot->bytes = bytes;
  }
  return MRX_OK;
}
#define MRX_LDS_CAP(ctx, kernel, bytes)                                                    \
  do {                                                                                     \
    int rc__ = mrx_lds_cap((ctx), reinterpret_cast<const void*>(kernel), (size_t)(bytes)); \
    if (rc__ != MRX_OK) return rc__;                                                       \
  } while (0)

// The last stage of the two-rate noise generator (mrx_noise.hip builds the slow part, the writer sits behind it in the same file
// next to the K_RJ machinery it shares with mrx_tod_to_krj): see noise_two_rate_kernel.
struct mrx_two_rate_args {
  const float* lo;          // [rows of this launch][ld_lo]: pink + correlated pink parts at fs / rate; sample t' at lo[t' + 1]
  size_t ld_lo;
  int rate;                 // 2 or 4
  const float* mode_white;  // [n_modes][ld_mw] unit white series of the modes, or null
  size_t ld_mw;
  int n_modes;
  const float* basis;       // [rows of the call][n_modes] (indexed by absolute row, like scale)
  float w_corr;             // sqrt(correlated proportion)
  float sqrt_fs;
  const float* scale;       // [rows of the call] or null
  const float* loading;     // [rows of the call][ld_loading] or null
  size_t ld_loading;
  float per_loading;
  float* out;               // [rows of the call][ld]
  size_t ld;
  int row0, rows;           // this launch: rows row0 .. row0 + rows - 1
  uint32_t id0;             // white-noise id of row 0 of the call
  int T;
  int accumulate;
  // TOD.to("K_RJ") in the same pass (arrays indexed by the call's rows), or bore_el = null
  const float* bore_el;
  const float* dx;
  const float* dy;
  const int32_t* band;
  const float* cal_axis;
  const float* cal_values;
  int n_el, n_bands;
  uint32_t key0, key1;
};
int mrx_noise_two_rate_write(mrx_ctx* ctx, hipStream_t stream, const mrx_two_rate_args& a);
// four white normals for samples 4 q .. 4 q + 3 of row `id`: the counter every white draw of the noise generator uses
constexpr uint32_t kMrxTagWhite = 0x57484954u;      // 'WHIT'
constexpr uint32_t kMrxTagModeWhite = 0x4d57484du;  // 'MWHM': the modes' white series (two-rate form)

// threads per workgroup of every tiled kernel (one entity, whichever header brings the name in)
namespace mrx_dev_common {
constexpr int kBlock = 256;
}

static inline int mrx_ceil_div(long long a, long long b) {
  return (int)((a + b - 1) / b);
}
