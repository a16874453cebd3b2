// Multi-GPU epilogue of libmrx: one RCCL all-gather of the TOD over xGMI
// (BASELINE north star).  Detector shards are equal blocks of contiguous rows
// of the detector-major [ndet][nt] TOD, so the gathered array IS the concatenation of the
// shards: one ncclAllGather, in place when the shard already sits in its slot of the full
// buffer -- no staging copy, no pack/unpack kernel.
//
// RCCL is bound at first use with dlopen (librccl.so.1: the copy torch has already loaded
// when the caller is a torch process, else ROCm's), so libmrx.so itself loads on machines
// without it and single-GPU users never touch it.
#include <dlfcn.h>

#include <mutex>

#include "mrx_internal.h"

struct mrx_comm {
  void* nccl = nullptr;  // ncclComm_t
  int world = 1, rank = 0;
  bool owned = false;  // created here (destroyed by mrx_comm_destroy) or wrapped
};

namespace {

struct UniqueId {
  char internal[MRX_COMM_ID_BYTES];
};
static_assert(sizeof(UniqueId) == 128, "ncclUniqueId is 128 bytes");

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

constexpr int kNcclFloat = 7;  // ncclFloat32 (rccl.h ncclDataType_t)

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.handle, "ncclBroadcast"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.handle, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.handle, "ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.handle, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.handle, "ncclGroupEnd"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Broadcast && r.Send && r.Recv &&
           r.GroupStart && r.GroupEnd;
  });
  return r;
}

int need_rccl(mrx_ctx* ctx) {
  if (!rccl().ok)
    return mrx_fail(ctx, MRX_ERR_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded: %s",
                    rccl().handle ? "missing symbols" : dlerror());
  return MRX_OK;
}

int nccl_fail(mrx_ctx* ctx, const char* what, int rc) {
  return mrx_fail(ctx, MRX_ERR_HIP, "%s failed: %s", what,
                  rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
}

}  // namespace

extern "C" {

int mrx_comm_unique_id(mrx_ctx* ctx, void* id_out) {
  if (!ctx || !id_out) return MRX_ERR_INVALID;
  int rc = need_rccl(ctx);
  if (rc != MRX_OK) return rc;
  UniqueId id;
  const int nrc = rccl().GetUniqueId(&id);
  if (nrc != 0) return nccl_fail(ctx, "ncclGetUniqueId", nrc);
  memcpy(id_out, id.internal, sizeof(id.internal));
  return MRX_OK;
}

int mrx_comm_create(mrx_ctx* ctx, const void* id, int world, int rank, mrx_comm** out) {
  MRX_ENTER(ctx);
  if (!ctx || !out) return MRX_ERR_INVALID;
  *out = nullptr;
  MRX_REQUIRE(ctx, id != nullptr, "null unique id");
  MRX_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "need 0 <= rank < world");
  int rc = need_rccl(ctx);
  if (rc != MRX_OK) return rc;
  UniqueId uid;
  memcpy(uid.internal, id, sizeof(uid.internal));
  mrx_comm* c = new (std::nothrow) mrx_comm();
  if (!c) return mrx_fail(ctx, MRX_ERR_ALLOC, "out of host memory");
  const int nrc = rccl().CommInitRank(&c->nccl, world, uid, rank);
  if (nrc != 0) {
    delete c;
    return nccl_fail(ctx, "ncclCommInitRank", nrc);
  }
  c->world = world;
  c->rank = rank;
  c->owned = true;
  *out = c;
  return MRX_OK;
}

int mrx_comm_wrap(mrx_ctx* ctx, void* nccl_comm, int world, int rank, mrx_comm** out) {
  if (!ctx || !out) return MRX_ERR_INVALID;
  *out = nullptr;
  MRX_REQUIRE(ctx, nccl_comm != nullptr, "null ncclComm_t");
  MRX_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "need 0 <= rank < world");
  int rc = need_rccl(ctx);
  if (rc != MRX_OK) return rc;
  mrx_comm* c = new (std::nothrow) mrx_comm();
  if (!c) return mrx_fail(ctx, MRX_ERR_ALLOC, "out of host memory");
  c->nccl = nccl_comm;
  c->world = world;
  c->rank = rank;
  c->owned = false;
  *out = c;
  return MRX_OK;
}

int mrx_comm_destroy(mrx_ctx* ctx, mrx_comm* comm) {
  MRX_ENTER(ctx);
  if (!ctx || !comm) return MRX_ERR_INVALID;
  int nrc = 0;
  if (comm->owned && comm->nccl) nrc = rccl().CommDestroy(comm->nccl);
  delete comm;
  return nrc == 0 ? MRX_OK : nccl_fail(ctx, "ncclCommDestroy", nrc);
}

int mrx_allgather_tod(mrx_ctx* ctx, mrx_comm* comm, const float* d_shard, float* d_full,
                      size_t count) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, comm != nullptr && comm->nccl != nullptr, "null communicator");
  if (count == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_shard && d_full, "null pointer");
  const float* slot = d_full + (size_t)comm->rank * count;
  MRX_REQUIRE(ctx, d_shard == slot || d_shard + count <= d_full ||
                       d_full + (size_t)comm->world * count <= d_shard,
              "d_shard must be this rank's slot of d_full (in place) or disjoint from it");
  const int nrc = rccl().AllGather(d_shard, d_full, count, kNcclFloat, comm->nccl, ctx->stream);
  if (nrc != 0) return nccl_fail(ctx, "ncclAllGather", nrc);
  return MRX_OK;
}

int mrx_allgather_tod_p2p(mrx_ctx* ctx, mrx_comm* comm, const float* d_shard, float* d_full,
                          size_t count) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, comm != nullptr && comm->nccl != nullptr, "null communicator");
  if (count == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_shard && d_full, "null pointer");
  float* slot = d_full + (size_t)comm->rank * count;
  MRX_REQUIRE(ctx, d_shard == slot || d_shard + count <= d_full ||
                       d_full + (size_t)comm->world * count <= d_shard,
              "d_shard must be this rank's slot of d_full (in place) or disjoint from it");
  // world - 1 sends and world - 1 receives in one group: every pair of GPUs has its own xGMI
  // link, so all of them move data at once (shard / link rate, against
  // (world - 1) x that for a ring)
  int nrc = rccl().GroupStart();
  if (nrc != 0) return nccl_fail(ctx, "ncclGroupStart", nrc);
  for (int k = 1; k < comm->world && nrc == 0; ++k) {
    const int to = (comm->rank + k) % comm->world, from = (comm->rank - k + comm->world) % comm->world;
    nrc = rccl().Send(d_shard, count, kNcclFloat, to, comm->nccl, ctx->stream);
    if (nrc == 0) nrc = rccl().Recv(d_full + (size_t)from * count, count, kNcclFloat, from, comm->nccl, ctx->stream);
  }
  const int erc = rccl().GroupEnd();
  if (nrc != 0) return nccl_fail(ctx, "ncclSend/ncclRecv", nrc);
  if (erc != 0) return nccl_fail(ctx, "ncclGroupEnd", erc);
  if (d_shard != slot)
    MRX_HIP(ctx, hipMemcpyAsync(slot, d_shard, count * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  return MRX_OK;
}

int mrx_exchange_screens(mrx_ctx* ctx, mrx_comm* comm, float* const* d_screens, const size_t* counts,
                         int n_layers) {
  MRX_ENTER(ctx);
  if (!ctx) return MRX_ERR_INVALID;
  MRX_REQUIRE(ctx, comm != nullptr && comm->nccl != nullptr, "null communicator");
  MRX_REQUIRE(ctx, n_layers >= 0, "negative layer count");
  if (n_layers == 0) return MRX_OK;
  MRX_REQUIRE(ctx, d_screens && counts, "null pointer");
  for (int l = 0; l < n_layers; ++l) MRX_REQUIRE(ctx, d_screens[l] != nullptr || counts[l] == 0, "null screen pointer");
  int nrc = rccl().GroupStart();
  if (nrc != 0) return nccl_fail(ctx, "ncclGroupStart", nrc);
  for (int l = 0; l < n_layers && nrc == 0; ++l)
    if (counts[l] > 0)
      nrc = rccl().Broadcast(d_screens[l], d_screens[l], counts[l], kNcclFloat, l % comm->world, comm->nccl, ctx->stream);
  const int erc = rccl().GroupEnd();
  if (nrc != 0) return nccl_fail(ctx, "ncclBroadcast", nrc);
  if (erc != 0) return nccl_fail(ctx, "ncclGroupEnd", erc);
  return MRX_OK;
}

}  // extern "C"
