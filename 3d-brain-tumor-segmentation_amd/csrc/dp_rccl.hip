// Data-parallel exchange entry points of the C ABI: the three collectives a per-sample sharded train step needs, over RCCL
// on device buffers, for hosts that bind include/bts_hip.h without Python (bts_amd.parallel issues the same collectives through
// torch.distributed, whose 'nccl' backend IS RCCL).  The reference is single-device (train.py:138); sharding train.py:140-152 by sample needs
//   C1  sum of the flat fp32 gradient buffer over the ranks, in a few large buckets (bts_dp_allreduce_buckets),
//   C2  the parameters of rank `root` on every rank after initialisation / load (bts_dp_broadcast_params),
//   C3  sum of the raw loss sums / Dice table, fp64 (bts_dp_allreduce_small: util.py:11,18-20 sum those over the batch axis).
// RCCL is bound at the first call by dlopen -- the copy the process already holds (a host that made the communicator has one; under Python
// torch's), else the system one -- so libbts_hip.so itself carries no link-time dependency on it and loads on boxes without RCCL.
// The communicator is the CALLER's (ncclComm_t passed as void*); bts_dp_comm_* are thin conveniences over ncclGetUniqueId /
// ncclCommInitRank / ncclCommDestroy for hosts that do not want to link RCCL themselves.  Every call is enqueued on the caller's stream.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
#include "bts_internal.h"
#include "../../include/bts_hip.h"

namespace {
struct Rccl {
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;
void load_rccl() {
  void* h = nullptr;
  const char* loaded[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : loaded) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy this process already holds
  const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : fresh) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return;
  Rccl r;
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
  r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(h, "ncclBroadcast"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));
  r.ok = r.AllReduce && r.Broadcast && r.GroupStart && r.GroupEnd && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount;
  g_rccl = r;
}
const Rccl* rccl() {
  std::call_once(g_once, load_rccl);
  return g_rccl.ok ? &g_rccl : nullptr;
}
inline int status(ncclResult_t r) { return r == ncclSuccess ? BTS_OK : BTS_ERR_RCCL_BASE - (int)r; }
}  // namespace

extern "C" int bts_dp_available(void) { return rccl() ? 1 : 0; }
extern "C" long bts_dp_unique_id_bytes(void) { return (long)sizeof(ncclUniqueId); }
extern "C" int bts_dp_comm_unique_id(void* id_out) {
  const Rccl* r = rccl();
  if (!r) return BTS_ERR_UNSUPPORTED;
  if (!id_out) return BTS_ERR_ALIGN;
  return status(r->GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out)));
}
extern "C" int bts_dp_comm_init(void** comm_out, int nranks, const void* id, int rank) {
  const Rccl* r = rccl();
  if (!r) return BTS_ERR_UNSUPPORTED;
  if (!comm_out || !id) return BTS_ERR_ALIGN;
  if (nranks <= 0 || rank < 0 || rank >= nranks) return BTS_ERR_SHAPE;
  ncclUniqueId u = *reinterpret_cast<const ncclUniqueId*>(id);
  ncclComm_t c = nullptr;
  const int s = status(r->CommInitRank(&c, nranks, u, rank));
  *comm_out = c;
  return s;
}
extern "C" int bts_dp_comm_destroy(void* comm) {
  const Rccl* r = rccl();
  if (!r) return BTS_ERR_UNSUPPORTED;
  if (!comm) return BTS_ERR_ALIGN;
  return status(r->CommDestroy(reinterpret_cast<ncclComm_t>(comm)));
}
// C1.  flat_grads: the model's flat fp32 gradient buffer (device); bucket b covers [off[b], off[b] + len[b]) elements (host arrays; any
// order -- pass them in the order the backward completes them to exchange a bucket while the rest is still being written: one call per
// finished bucket with nbuckets = 1 is the overlapped form).  In-place sum over the ranks; all buckets of a call form one RCCL group.
extern "C" int bts_dp_allreduce_buckets(void* comm, float* flat_grads, const long* bucket_off, const long* bucket_len, int nbuckets,
                                        hipStream_t stream) {
  const Rccl* r = rccl();
  if (!r) return BTS_ERR_UNSUPPORTED;
  if (!comm || !flat_grads || !bucket_off || !bucket_len) return BTS_ERR_ALIGN;
  if (nbuckets <= 0) return BTS_ERR_SHAPE;
  for (int b = 0; b < nbuckets; ++b)
    if (bucket_off[b] < 0 || bucket_len[b] <= 0) return BTS_ERR_SHAPE;
  ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
  int s = status(r->GroupStart());
  if (s != BTS_OK) return s;
  for (int b = 0; b < nbuckets && s == BTS_OK; ++b)
    s = status(r->AllReduce(flat_grads + bucket_off[b], flat_grads + bucket_off[b], (size_t)bucket_len[b], ncclFloat32, ncclSum, c, stream));
  const int e = status(r->GroupEnd());
  return s != BTS_OK ? s : e;
}
// C2.  every rank ends up with rank `root`'s n fp32 values (the flat parameter buffer; also Adam moments after a resume)
extern "C" int bts_dp_broadcast_params(void* comm, float* flat_params, long n, int root, hipStream_t stream) {
  const Rccl* r = rccl();
  if (!r) return BTS_ERR_UNSUPPORTED;
  if (!comm || !flat_params) return BTS_ERR_ALIGN;
  int nr = 0;
  ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
  const int s = status(r->CommCount(c, &nr));
  if (s != BTS_OK) return s;
  if (n <= 0 || root < 0 || root >= nr) return BTS_ERR_SHAPE;
  return status(r->Broadcast(flat_params, flat_params, (size_t)n, ncclFloat32, root, c, stream));
}
// C3.  in-place sum of n fp64 values (bts_loss_sums' raw sums, the Dice metric's table)
extern "C" int bts_dp_allreduce_small(void* comm, double* sums, int n, hipStream_t stream) {
  const Rccl* r = rccl();
  if (!r) return BTS_ERR_UNSUPPORTED;
  if (!comm || !sums) return BTS_ERR_ALIGN;
  if (n <= 0) return BTS_ERR_SHAPE;
  return status(r->AllReduce(sums, sums, (size_t)n, ncclFloat64, ncclSum, reinterpret_cast<ncclComm_t>(comm), stream));
}
