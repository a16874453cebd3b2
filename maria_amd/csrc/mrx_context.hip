// Context, stream binding, flags and the HIP-event timer of libmrx.
#include "mrx_internal.h"

extern "C" {

int mrx_version(void) { return MRX_VERSION; }

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
  if (ctx->d_reduce) (void)hipFree(ctx->d_reduce);
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
