// Context, stream binding, flags and the HIP-event timer of libmrx.
#include "mrx_internal.h"

namespace {

// busy for `ticks` of the 100 MHz constant clock; every wave leaves once the time has passed
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace

int mrx_probe_concurrent(mrx_ctx* ctx, hipStream_t a, hipStream_t b, bool* concurrent) {
  *concurrent = true;
  if (a == b) {
    *concurrent = false;
    return MRX_OK;
  }
  constexpr long long kTicks = 15000;  // 150 us
  struct Events {  // released on every way out
    hipEvent_t e0 = nullptr, e1 = nullptr, eb = nullptr;
    ~Events() {
      for (hipEvent_t e : {e0, e1, eb})
        if (e) (void)hipEventDestroy(e);
    }
  } ev;
  hipEvent_t &e0 = ev.e0, &e1 = ev.e1, &eb = ev.eb;
  MRX_HIP(ctx, hipEventCreate(&e0));
  MRX_HIP(ctx, hipEventCreate(&e1));
  MRX_HIP(ctx, hipEventCreateWithFlags(&eb, hipEventDisableTiming));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {  // the best of three: the first launch on a new stream pays for its queue
    MRX_HIP(ctx, hipStreamSynchronize(a));
    MRX_HIP(ctx, hipStreamSynchronize(b));
    MRX_HIP(ctx, hipEventRecord(e0, a));
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, kTicks);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, kTicks);
    MRX_HIP(ctx, hipEventRecord(eb, b));
    MRX_HIP(ctx, hipStreamWaitEvent(a, eb, 0));
    MRX_HIP(ctx, hipEventRecord(e1, a));
    MRX_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    MRX_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  *concurrent = best < 1.6f * (float)kTicks * 1e-5f;  // one spin is 0.15 ms, two in a row 0.30
  return MRX_OK;
}

int mrx_side_streams(mrx_ctx* ctx, int n) {
  if (n > mrx_ctx::kSideStreams) return mrx_fail(ctx, MRX_ERR_INVALID, "at most %d side streams", mrx_ctx::kSideStreams);
  if (!ctx->side_ev[0]) MRX_HIP(ctx, hipEventCreateWithFlags(&ctx->side_ev[0], hipEventDisableTiming));
  // (re)check the streams against the context's current stream: one that shares its hardware queue would
  // run its batches behind the main lane's instead of beside them
  const bool recheck = !ctx->side_checked || ctx->side_probed != ctx->stream;
  for (int i = 0; i < n; ++i) {
    if (!ctx->side_ev[1 + i]) MRX_HIP(ctx, hipEventCreateWithFlags(&ctx->side_ev[1 + i], hipEventDisableTiming));
    if (ctx->side_streams[i] && !recheck) continue;
    for (int attempt = 0; attempt < 6; ++attempt) {
      if (!ctx->side_streams[i]) MRX_HIP(ctx, hipStreamCreateWithFlags(&ctx->side_streams[i], hipStreamNonBlocking));
      bool ok = true;
      const int rc = mrx_probe_concurrent(ctx, ctx->stream, ctx->side_streams[i], &ok);
      if (rc != MRX_OK) return rc;
      if (ok || attempt == 5) break;  // (a device with a single queue: keep what there is)
      (void)hipStreamDestroy(ctx->side_streams[i]);
      ctx->side_streams[i] = nullptr;
    }
  }
  ctx->side_probed = ctx->stream;
  ctx->side_checked = true;
  return MRX_OK;
}

extern "C" {

int mrx_version(void) { return MRX_VERSION; }

int mrx_streams_concurrent(mrx_ctx* ctx, void* other_stream, int* concurrent) {
  MRX_ENTER(ctx);
  if (!ctx || !concurrent) return MRX_ERR_INVALID;
  bool ok = true;
  const int rc = mrx_probe_concurrent(ctx, ctx->stream, (hipStream_t)other_stream, &ok);
  *concurrent = ok ? 1 : 0;
  return rc;
}

int mrx_init(int device, mrx_ctx** out) {
  if (!out) return MRX_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0 || device < 0 || device >= n)
    return MRX_ERR_NO_DEVICE;
  mrx_ctx* ctx = new (std::nothrow) mrx_ctx();
  if (!ctx) return MRX_ERR_ALLOC;
  ctx->device = device;
  // the calling thread's current device is the caller's business (torch's, in a multi-GPU
  // process): switch for the duration of the call only
  mrx_device_guard guard(ctx);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete ctx;
    return MRX_ERR_HIP;
  }
  ctx->n_cu = prop.multiProcessorCount;
  ctx->lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  ctx->hbm_bytes = prop.totalGlobalMem;
  snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name,
           prop.gcnArchName);
  if (hipEventCreate(&ctx->ev_start) != hipSuccess ||
      hipEventCreate(&ctx->ev_stop) != hipSuccess) {
    delete ctx;
    return MRX_ERR_HIP;
  }
  *out = ctx;
  return MRX_OK;
}

int mrx_destroy(mrx_ctx* ctx) {
  if (!ctx) return MRX_ERR_INVALID;
  {
  mrx_device_guard guard(ctx);
  for (auto& slot : ctx->taps)
    if (slot.d_taps) (void)hipFree(slot.d_taps);
  for (auto& slot : ctx->ftaps)
    if (slot.d_taps) (void)hipFree(slot.d_taps);
  for (auto& slot : ctx->fresp)
    if (slot.d_resp) (void)hipFree(slot.d_resp);
  if (ctx->d_reduce) (void)hipFree(ctx->d_reduce);
  if (ctx->d_synth_ctl) (void)hipFree(ctx->d_synth_ctl);
  if (ctx->d_bin_order) (void)hipFree(ctx->d_bin_order);
  if (ctx->d_map_pairs) (void)hipFree(ctx->d_map_pairs);
  if (ctx->map_pairs_read) (void)hipEventDestroy(ctx->map_pairs_read);
  for (hipStream_t st : ctx->side_streams)
    if (st) (void)hipStreamDestroy(st);
  for (hipEvent_t ev : ctx->side_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
  if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
  }
  delete ctx;
  return MRX_OK;
}

int mrx_set_stream(mrx_ctx* ctx, void* hip_stream) {
  if (!ctx) return MRX_ERR_INVALID;
  ctx->stream = (hipStream_t)hip_stream;
  return MRX_OK;
}

int mrx_set_option(mrx_ctx* ctx, int option, int value) {
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, option >= 0 && option < MRX_OPT_COUNT, "unknown option");
  ctx->options[option] = value;
  return MRX_OK;
}

int mrx_synchronize(mrx_ctx* ctx) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MRX_OK;
}

const char* mrx_last_error(const mrx_ctx* ctx) {
  return ctx ? ctx->err : "null context";
}

int mrx_device_info(const mrx_ctx* ctx, int* n_cu, int* lds_bytes_per_cu,
                    size_t* hbm_bytes, char* name, int name_len) {
  if (!ctx) return MRX_ERR_INVALID;
  if (n_cu) *n_cu = ctx->n_cu;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = ctx->lds_per_cu;
  if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
  if (name && name_len > 0) {
    strncpy(name, ctx->name, (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  return MRX_OK;
}

int mrx_timer_start(mrx_ctx* ctx) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
  return MRX_OK;
}

int mrx_timer_stop(mrx_ctx* ctx, float* elapsed_ms) {
  MRX_ENTER(ctx);
  if (!ctx || !elapsed_ms) return MRX_ERR_INVALID;
  MRX_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
  MRX_HIP(ctx, hipEventSynchronize(ctx->ev_stop));
  MRX_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev_start, ctx->ev_stop));
  return MRX_OK;
}

int mrx_clear_flags(mrx_ctx* ctx, uint32_t* d_flags) {
  MRX_ENTER(ctx);
  if (!ctx || !d_flags) return MRX_ERR_INVALID;
  MRX_HIP(ctx, hipMemsetAsync(d_flags, 0, sizeof(uint32_t), ctx->stream));
  return MRX_OK;
}

int mrx_read_flags(mrx_ctx* ctx, const uint32_t* d_flags,
                   uint32_t* host_flags) {
  MRX_ENTER(ctx);
  if (!ctx || !d_flags || !host_flags) return MRX_ERR_INVALID;
  MRX_HIP(ctx, hipMemcpyAsync(host_flags, d_flags, sizeof(uint32_t),
                              hipMemcpyDeviceToHost, ctx->stream));
  MRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MRX_OK;
}

}  // extern "C"
