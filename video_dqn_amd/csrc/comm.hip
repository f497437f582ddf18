// The data-parallel exchange of the C ABI: comm_init + allreduce_bucket over RCCL, for a caller that binds
// include/vdqn.h without PyTorch.  The reference has no distributed code (single GPU: train_q_network.py:255-259,275); the one
// exchange the N > 1 path needs is a SUM all-reduce of the flat f32 gradient, issued per backward stage
// (vdqn_net_stage_range / vdqn_net_grad_stream).  The Python host (video_dqn_amd/dist.py) uses torch.distributed's "nccl"
// backend = the same RCCL; these entries are the equivalent for C callers.
//
// RCCL is bound at run time (dlopen + dlsym), not at link time: a process that already loaded an RCCL (torch ships its own
// librccl.so) must not get a second copy with its own topology state, and a single-GPU user of libvdqn.so must not need RCCL at
// all.  The handle of an already-loaded librccl.so is reused (RTLD_NOLOAD first); VDQN_RCCL_LIB names another file.
#include <dlfcn.h>
#include <stdlib.h>

#include <mutex>

#include "common.h"

namespace {

// the part of rccl.h this file needs (ABI-stable NCCL 2.x types)
struct UniqueId { char internal[VDQN_COMM_UID_BYTES]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclBfloat16 = 9;

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
  bool tried = false;
};
Rccl g_rccl;
std::mutex g_mu;

const Rccl* rccl() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rccl.tried) return g_rccl.handle ? &g_rccl : nullptr;
  g_rccl.tried = true;
  const char* names[] = {getenv("VDQN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names)  // an RCCL this process already holds (torch's) wins
    if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const char* n : names)
    if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return nullptr;
  g_rccl.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
  g_rccl.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
  g_rccl.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
  g_rccl.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) {
    dlclose(h);
    return nullptr;
  }
  g_rccl.handle = h;
  return &g_rccl;
}

int fail(const Rccl* r, const char* what, int code) {
  vdqn_set_error("%s: RCCL error %d (%s)", what, code, (r && r->error_string) ? r->error_string(code) : "?");
  return VDQN_ERR_LAUNCH;
}

}  // namespace

struct vdqn_comm {
  Comm comm;
  int rank, nranks, device;
};

extern "C" {

int vdqn_comm_unique_id(void* uid_host) {
  VDQN_CHECK(uid_host != nullptr, "vdqn_comm_unique_id: uid_host is NULL");
  const Rccl* r = rccl();
  VDQN_CHECK(r != nullptr, "vdqn_comm_unique_id: no RCCL library found (librccl.so; set VDQN_RCCL_LIB)");
  UniqueId id;
  const int e = r->get_unique_id(&id);
  if (e != 0) return fail(r, "ncclGetUniqueId", e);
  memcpy(uid_host, id.internal, VDQN_COMM_UID_BYTES);
  return VDQN_OK;
}

int vdqn_comm_init(int32_t rank, int32_t nranks, const void* uid_host, vdqn_comm** out) {
  VDQN_CHECK(out != nullptr && uid_host != nullptr, "vdqn_comm_init: NULL argument");
  *out = nullptr;
  VDQN_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "vdqn_comm_init: rank %d of %d", rank, nranks);
  const Rccl* r = rccl();
  VDQN_CHECK(r != nullptr, "vdqn_comm_init: no RCCL library found (librccl.so; set VDQN_RCCL_LIB)");
  UniqueId id;
  memcpy(id.internal, uid_host, VDQN_COMM_UID_BYTES);
  Comm c = nullptr;
  const int e = r->comm_init_rank(&c, nranks, id, rank);  // on the calling thread's current HIP device
  if (e != 0) return fail(r, "ncclCommInitRank", e);
  vdqn_comm* h = new vdqn_comm{c, rank, nranks, 0};
  (void)hipGetDevice(&h->device);
  *out = h;
  return VDQN_OK;
}

int vdqn_allreduce_bucket(vdqn_comm* comm, void* ptr, int64_t count, int32_t dtype, void* stream) {
  VDQN_CHECK(comm != nullptr && comm->comm != nullptr, "vdqn_allreduce_bucket: no communicator");
  VDQN_CHECK(count >= 0 && (ptr != nullptr || count == 0), "vdqn_allreduce_bucket: bad buffer");
  VDQN_CHECK(dtype == VDQN_F32 || dtype == VDQN_BF16, "vdqn_allreduce_bucket: dtype %d", dtype);
  if (count == 0) return VDQN_OK;
  const Rccl* r = rccl();
  VDQN_CHECK(r != nullptr, "vdqn_allreduce_bucket: RCCL is gone");
  const int e = r->all_reduce(ptr, ptr, (size_t)count, dtype == VDQN_F32 ? kNcclFloat32 : kNcclBfloat16, kNcclSum, comm->comm,
                              (hipStream_t)stream);
  if (e != 0) return fail(r, "ncclAllReduce", e);
  return VDQN_OK;
}

int vdqn_comm_rank(const vdqn_comm* comm) { return comm ? comm->rank : -1; }
int vdqn_comm_size(const vdqn_comm* comm) { return comm ? comm->nranks : -1; }

int vdqn_comm_destroy(vdqn_comm* comm) {
  if (!comm) return VDQN_OK;
  const Rccl* r = rccl();
  int e = 0;
  if (r && comm->comm) e = r->comm_destroy(comm->comm);
  delete comm;
  if (e != 0) return fail(r, "ncclCommDestroy", e);
  return VDQN_OK;
}

}  // extern "C"
