// Data-parallel exchange step of the fitting loop for MI355X nodes: ONE in-place RCCL all-reduce (sum) of the flat fp32 gradient
// bucket, enqueued on the caller's HIP stream (SURVEY.md §8(b) `allreduce_flat`, §8(e)).  The reference is single-device (SURVEY.md
// §2.2), so there is no call site to replace: the collective sits between `sum_loss.backward()` and `opt_*.step()`
// (optimize_sequence.py:567-573) of a frame-sharded job.
//
// RCCL is bound at run time (dlopen) and only when a communicator is first asked for: libharp_hip.so itself stays loadable on a
// host without RCCL, and inside a PyTorch process the copy torch has already mapped is reused (two RCCL runtimes in one process
// would each open their own xGMI / IPC state).  Being a plain stream-ordered enqueue, the call can be captured into the step's
// hipGraph like every other node (no host-side work object, no extra stream, no event round trip).
#include <dlfcn.h>
#include <functional>
#include <mutex>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "harp_common.h"
#include "harp_hip.h"

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

void rccl_bind(Rccl& r) {
  // 1. whatever the process already mapped (PyTorch ships its own librccl.so); 2. the ROCm install
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names) {
    r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (r.h) break;
  }
  if (!r.h) {
    const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* p : paths) {
      r.h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
  }
  if (!r.h) return;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
}

// bound once, thread-safe: two threads creating communicators at the same time must not race on the function table
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, rccl_bind, std::ref(r));
  return r;
}

inline int rc_of(ncclResult_t e) { return e == ncclSuccess ? HARP_OK : HARP_ERR_COMM + (int)e; }

}  // namespace

extern "C" {

int harp_comm_unique_id(void* id_out) {
  if (!id_out) return HARP_ERR_ARG;
  Rccl& r = rccl();
  if (!r.ok) return HARP_ERR_NO_RCCL;
  static_assert(sizeof(ncclUniqueId) == HARP_COMM_ID_BYTES, "HARP_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
  ncclUniqueId id;
  const ncclResult_t e = r.GetUniqueId(&id);
  if (e != ncclSuccess) return rc_of(e);
  memcpy(id_out, &id, sizeof(id));
  return HARP_OK;
}

int harp_comm_create(const void* id, int rank, int world, void** comm_out) {
  if (!id || !comm_out || world <= 0 || rank < 0 || rank >= world) return HARP_ERR_ARG;
  Rccl& r = rccl();
  if (!r.ok) return HARP_ERR_NO_RCCL;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  const ncclResult_t e = r.CommInitRank(&c, world, uid, rank);      // collective over all ranks; binds the current HIP device
  if (e != ncclSuccess) return rc_of(e);
  *comm_out = (void*)c;
  return HARP_OK;
}

int harp_comm_destroy(void* comm) {
  if (!comm) return HARP_ERR_ARG;
  Rccl& r = rccl();
  if (!r.ok) return HARP_ERR_NO_RCCL;
  return rc_of(r.CommDestroy((ncclComm_t)comm));
}

int harp_allreduce_flat(void* comm, float* buf, size_t n, hipStream_t stream) {
  if (!comm || !buf) return HARP_ERR_ARG;
  if (n == 0) return HARP_OK;
  Rccl& r = rccl();
  if (!r.ok) return HARP_ERR_NO_RCCL;
  return rc_of(r.AllReduce(buf, buf, n, ncclFloat32, ncclSum, (ncclComm_t)comm, stream));
}

}  // extern "C"
