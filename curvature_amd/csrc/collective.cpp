// The one data-path collective of the layer-sharded estimators: after sample_and_replace() every rank
// holds the sampled parameters of ITS layers; one all-gather over RCCL (xGMI) reassembles the flat parameter vector on
// every rank.  The shards have different lengths (the layer partition balances time, not parameter counts: by a factor
// of several), so this is a variable-count all-gather IN PLACE on the flat vector: one ncclBroadcast per rank inside a
// group call (RCCL fuses them into one launch), every rank's segment travelling exactly once - nothing is padded to
// the largest shard.
//
// RCCL is bound at run time (dlopen / dlsym): the library keeps libamdhip64 as its only link-time dependency and uses
// whichever librccl the process has loaded already (torch's), so the communicator handed in by the caller and the
// functions called on it come from the same copy.
#include "common.h"

#include <dlfcn.h>
#include <mutex>

namespace curv {
namespace {
struct Rccl {
  void* handle = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, const void*, int) = nullptr;     // (ncclUniqueId passed by value: see comm_init)
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  char why[256] = "";                       // why `ok` is false (dlerror() is read ONCE: reading it clears it)
};
constexpr int kNcclFloat32 = 7;           // rccl.h: ncclFloat32 = 7
constexpr int kUniqueIdBytes = 128;       // rccl.h: NCCL_UNIQUE_ID_BYTES

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) { r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (r.handle) break; }   // already in the process?
    for (const char* n : names) { if (r.handle) break; r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
    if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) {
      const char* e = dlerror();
      snprintf(r.why, sizeof(r.why), "dlopen librccl.so: %s", e ? e : "unknown error");
      return;
    }
    auto sym = [&](const char* name) { return dlsym(r.handle, name); };
    r.GroupStart = (int (*)())sym("ncclGroupStart");
    r.GroupEnd = (int (*)())sym("ncclGroupEnd");
    r.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclBroadcast");
    r.CommCount = (int (*)(void*, int*))sym("ncclCommCount");
    r.CommUserRank = (int (*)(void*, int*))sym("ncclCommUserRank");
    r.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, const void*, int))sym("ncclCommInitRank");
    r.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    r.ok = r.GroupStart && r.GroupEnd && r.Broadcast && r.CommCount && r.CommUserRank && r.GetUniqueId && r.CommInitRank &&
           r.CommDestroy && r.GetErrorString;
    if (!r.ok) snprintf(r.why, sizeof(r.why), "librccl.so is loaded but lacks a symbol of the nccl* API this library binds");
  });
  return r;
}

int need_rccl(Rccl** out) {
  Rccl& r = rccl();
  if (!r.ok) {
    set_error("RCCL is not available in this process (%s)", r.why);
    return CURV_ERR_HIP;
  }
  *out = &r;
  return CURV_OK;
}

#define CURV_RCCL_CHECK(r, expr)                                                                    \
  do {                                                                                              \
    const int _s = (expr);                                                                          \
    if (_s != 0) {                                                                                  \
      set_error("%s failed: %s", #expr, (r)->GetErrorString(_s));                                   \
      return CURV_ERR_HIP;                                                                          \
    }                                                                                               \
  } while (0)
}  // namespace
}  // namespace curv

using namespace curv;

// 1 when RCCL can be reached through this library (dlopen + every symbol bound), 0 otherwise.  Local and free of side
// effects: no communicator, no bootstrap socket - the probe a caller runs before the collective steps.
extern "C" int curv_rccl_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int curv_allgather_weights(void* comm, void* stream, float* flat, const long long* counts,
                                      const long long* displs) {
  Rccl* r = nullptr;
  int rc = need_rccl(&r);
  if (rc != CURV_OK) return rc;
  CURV_REQUIRE(comm != nullptr && flat != nullptr && counts != nullptr && displs != nullptr,
               "curv_allgather_weights: null argument");
  int world = 0;
  CURV_RCCL_CHECK(r, r->CommCount(comm, &world));
  for (int k = 0; k < world; ++k)
    CURV_REQUIRE(counts[k] >= 0 && displs[k] >= 0, "curv_allgather_weights: negative count or displacement of rank %d", k);
  CURV_RCCL_CHECK(r, r->GroupStart());
  for (int k = 0; k < world; ++k) {
    if (counts[k] == 0) continue;
    float* seg = flat + displs[k];
    const int s = r->Broadcast(seg, seg, (size_t)counts[k], kNcclFloat32, k, comm, (hipStream_t)stream);
    if (s != 0) {
      (void)r->GroupEnd();
      set_error("ncclBroadcast (segment of rank %d) failed: %s", k, r->GetErrorString(s));
      return CURV_ERR_HIP;
    }
  }
  CURV_RCCL_CHECK(r, r->GroupEnd());
  return CURV_OK;
}

// Communicator plumbing for callers without RCCL bindings of their own (the Python estimators: torch.distributed does
// not hand out its ncclComm_t).  Rank 0 draws the id, the caller ships its 128 bytes to the other ranks by whatever
// means it has (a torch.distributed broadcast), every rank calls curv_comm_init with its device current.
extern "C" int curv_comm_unique_id(void* id_out) {
  Rccl* r = nullptr;
  int rc = need_rccl(&r);
  if (rc != CURV_OK) return rc;
  CURV_REQUIRE(id_out != nullptr, "curv_comm_unique_id: null argument");
  CURV_RCCL_CHECK(r, r->GetUniqueId(id_out));
  return CURV_OK;
}

extern "C" int curv_comm_init(void** comm_out, int n_ranks, const void* id, int rank) {
  Rccl* r = nullptr;
  int rc = need_rccl(&r);
  if (rc != CURV_OK) return rc;
  CURV_REQUIRE(comm_out != nullptr && id != nullptr && n_ranks > 0 && rank >= 0 && rank < n_ranks, "curv_comm_init: bad argument");
  // ncclCommInitRank takes the 128-byte id BY VALUE; calling through a struct-typed pointer keeps the C ABI for that
  struct Id { char b[kUniqueIdBytes]; };
  Id copy;
  memcpy(copy.b, id, kUniqueIdBytes);
  auto init = (int (*)(void**, int, Id, int))r->CommInitRank;
  CURV_RCCL_CHECK(r, init(comm_out, n_ranks, copy, rank));
  return CURV_OK;
}

extern "C" int curv_comm_destroy(void* comm) {
  Rccl* r = nullptr;
  int rc = need_rccl(&r);
  if (rc != CURV_OK) return rc;
  if (comm != nullptr) CURV_RCCL_CHECK(r, r->CommDestroy(comm));
  return CURV_OK;
}
