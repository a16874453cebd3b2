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
  int options[MRX_OPT_COUNT] = {0};
  // screen normalisations (sum of the PSD over the FFT grid), one device double
  // per distinct (grid, spectrum)
  static constexpr int kPsdSlots = 64;
  struct PsdKey {
    bool valid = false;
    int ny = 0, nx = 0;
    double dy = 0, dx = 0, r0 = 0, nu = 0;
  } psd[kPsdSlots];
  int psd_next = 0;
  double* d_reduce = nullptr;  // kPsdSlots doubles
  size_t reduce_cap = 0;
};

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

static inline int mrx_ceil_div(long long a, long long b) {
  return (int)((a + b - 1) / b);
}
