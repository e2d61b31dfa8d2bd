// A12 / SURVEY 8(e): the one collective of the path - a float64 SUM all-reduce of the per-speaker [sums..., count] buffer
// over RCCL (xGMI) - behind the C ABI, for callers that are not Python (the Python mirror reaches the same RCCL through
// torch.distributed, ssr_eval_amd/dist.py).  RCCL is resolved with dlopen at the first call: the library carries no
// link-time dependency on it and loads on a box without RCCL.
#include <dlfcn.h>

#include <mutex>

#include "ssr_host.h"

namespace {
struct NcclUid { char internal[128]; };                      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES)
typedef int (*fn_get_uid)(NcclUid*);
typedef int (*fn_init_rank)(void**, int, NcclUid, int);      // (ncclComm_t*, nranks, id BY VALUE, rank)
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);
struct Rccl {
  void* h = nullptr;
  fn_get_uid get_uid = nullptr; fn_init_rank init_rank = nullptr; fn_destroy destroy = nullptr;
  fn_allreduce allreduce = nullptr; fn_errstr errstr = nullptr;
};
int rccl(Rccl** out) {
  static Rccl r;
  static bool ok = false;
  static std::once_flag once;                                 // entry points may be called from several host threads
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!r.h) r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.h) {
      r.get_uid = (fn_get_uid)dlsym(r.h, "ncclGetUniqueId");
      r.init_rank = (fn_init_rank)dlsym(r.h, "ncclCommInitRank");
      r.destroy = (fn_destroy)dlsym(r.h, "ncclCommDestroy");
      r.allreduce = (fn_allreduce)dlsym(r.h, "ncclAllReduce");
      r.errstr = (fn_errstr)dlsym(r.h, "ncclGetErrorString");
    }
    ok = r.h && r.get_uid && r.init_rank && r.destroy && r.allreduce;
  });
  if (!ok) return ssr_fail(SSR_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded");
  *out = &r;
  return SSR_OK;
}
int nccl_fail(Rccl* r, const char* what, int code) {
  return ssr_fail(SSR_ERR_HIP, std::string(what) + ": " + (r->errstr ? r->errstr(code) : "RCCL error") );
}
}  // namespace

extern "C" int ssr_comm_unique_id(void* uid128) {
  if (!uid128) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  Rccl* r;
  if (int rc = rccl(&r)) return rc;
  if (int e = r->get_uid((NcclUid*)uid128)) return nccl_fail(r, "ncclGetUniqueId", e);
  return SSR_OK;
}

extern "C" int ssr_comm_init_rank(const void* uid128, int n_ranks, int rank, void** comm) {
  if (!uid128 || !comm) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  *comm = nullptr;
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return ssr_fail(SSR_ERR_INVALID_ARG, "rank outside [0, n_ranks)");
  Rccl* r;
  if (int rc = rccl(&r)) return rc;
  NcclUid id;
  memcpy(&id, uid128, sizeof(id));
  if (int e = r->init_rank(comm, n_ranks, id, rank)) return nccl_fail(r, "ncclCommInitRank", e);
  return SSR_OK;
}

extern "C" int ssr_comm_destroy(void* comm) {
  if (!comm) return SSR_OK;
  Rccl* r;
  if (int rc = rccl(&r)) return rc;
  if (int e = r->destroy(comm)) return nccl_fail(r, "ncclCommDestroy", e);
  return SSR_OK;
}

extern "C" int ssr_allreduce_sums(double* buf, int n, void* comm, void* stream) {
  if (!buf || !comm) return ssr_fail(SSR_ERR_INVALID_ARG, "null argument");
  if (n < 0) return ssr_fail(SSR_ERR_INVALID_ARG, "negative count");
  if (n == 0) return SSR_OK;
  Rccl* r;
  if (int rc = rccl(&r)) return rc;
  if (int e = r->allreduce(buf, buf, (size_t)n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm, (hipStream_t)stream))
    return nccl_fail(r, "ncclAllReduce", e);
  return SSR_OK;
}
