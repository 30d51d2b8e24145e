// RCCL (xGMI) communicator of libpysteps_hip.so: one rank per GPU.
//
// The reference has no communication layer at all (no NCCL/MPI
// call sites); the advection path shards embarrassingly (independent members /
// fields per GPU), so the only collective on the data path is ONE broadcast of the
// input fields from the rank that produced them, plus small all-gathers of sparse
// vectors for tiled domains.  librccl.so is opened lazily with dlopen so that the
// single-GPU product never depends on it.
#include <dlfcn.h>

#include <cstring>

#include "common.h"

namespace psh {
namespace {

// minimal mirror of the RCCL C API (rccl.h: ncclUniqueId is 128 opaque bytes,
// ncclComm_t an opaque pointer, ncclUint8 == 1, ncclSuccess == 0)
struct UniqueId {
  char internal[128];
};
using Comm = void *;
using GetUniqueIdFn = int (*)(UniqueId *);
using CommInitRankFn = int (*)(Comm *, int, UniqueId, int);
using CommDestroyFn = int (*)(Comm);
using BroadcastFn = int (*)(const void *, void *, size_t, int, int, Comm, hipStream_t);
using AllGatherFn = int (*)(const void *, void *, size_t, int, Comm, hipStream_t);
using AllReduceFn = int (*)(const void *, void *, size_t, int, int, Comm, hipStream_t);
using ErrStrFn = const char *(*)(int);
constexpr int kNcclUint8 = 1, kNcclFloat32 = 7;       // ncclDataType_t
constexpr int kNcclSum = 0, kNcclMax = 2, kNcclMin = 3;  // ncclRedOp_t

struct Rccl {
  void *handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  BroadcastFn broadcast = nullptr;
  AllGatherFn all_gather = nullptr;
  AllReduceFn all_reduce = nullptr;
  ErrStrFn err_str = nullptr;
  Comm comm = nullptr;
  int nranks = 0, rank = -1;
};

Rccl &rccl() {
  static Rccl r;
  return r;
}

int load_rccl() {
  Rccl &r = rccl();
  if (r.handle) return PSH_OK;
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char *nm : names) {
    r.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (r.handle) break;
  }
  if (!r.handle) return fail(PSH_ECOMM, "cannot dlopen librccl.so: %s", dlerror());
  r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(r.handle, "ncclGetUniqueId"));
  r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(r.handle, "ncclCommInitRank"));
  r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(r.handle, "ncclCommDestroy"));
  r.broadcast = reinterpret_cast<BroadcastFn>(dlsym(r.handle, "ncclBroadcast"));
  r.all_gather = reinterpret_cast<AllGatherFn>(dlsym(r.handle, "ncclAllGather"));
  r.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(r.handle, "ncclAllReduce"));
  r.err_str = reinterpret_cast<ErrStrFn>(dlsym(r.handle, "ncclGetErrorString"));
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.broadcast || !r.all_gather || !r.all_reduce)
    return fail(PSH_ECOMM, "librccl.so lacks an expected symbol");
  return PSH_OK;
}

int rccl_fail(const char *what, int rc) {
  Rccl &r = rccl();
  return fail(PSH_ECOMM, "%s failed: %s", what, r.err_str ? r.err_str(rc) : "unknown RCCL error");
}

}  // namespace
}  // namespace psh

using psh::fail;
using psh::rccl;

extern "C" {

int psh_comm_unique_id_bytes(void) { return static_cast<int>(sizeof(psh::UniqueId)); }

int psh_comm_unique_id(void *id_out) {
  if (!id_out) return fail(PSH_EINVAL, "psh_comm_unique_id: NULL pointer");
  if (int rc = psh::load_rccl()) return rc;
  psh::UniqueId id;
  if (int rc = rccl().get_unique_id(&id)) return psh::rccl_fail("ncclGetUniqueId", rc);
  std::memcpy(id_out, &id, sizeof(id));
  return PSH_OK;
}

int psh_comm_init(const void *id, int nranks, int rank) {
  PSH_REQUIRE_INIT();
  if (!id) return fail(PSH_EINVAL, "psh_comm_init: NULL id");
  if (nranks < 1 || rank < 0 || rank >= nranks)
    return fail(PSH_EINVAL, "psh_comm_init: bad rank %d of %d", rank, nranks);
  if (int rc = psh::load_rccl()) return rc;
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  if (rccl().comm) return fail(PSH_EINVAL, "psh_comm_init: communicator already initialised");
  psh::UniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  psh::Comm comm = nullptr;
  if (int rc = rccl().comm_init_rank(&comm, nranks, uid, rank)) return psh::rccl_fail("ncclCommInitRank", rc);
  rccl().comm = comm;
  rccl().nranks = nranks;
  rccl().rank = rank;
  return PSH_OK;
}

int psh_comm_broadcast(void *buf_dev, size_t nbytes, int root) {
  PSH_REQUIRE_INIT();
  if (!rccl().comm) return fail(PSH_ECOMM, "psh_comm_broadcast: communicator not initialised");
  if (!buf_dev && nbytes) return fail(PSH_EINVAL, "psh_comm_broadcast: NULL buffer");
  if (root < 0 || root >= rccl().nranks) return fail(PSH_EINVAL, "psh_comm_broadcast: bad root %d", root);
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  if (nbytes == 0) return PSH_OK;
  // one large in-place message: xGMI is point-to-point (per-link bound), so the
  // fields travel as a single pipelined broadcast instead of many small ones
  if (int rc = rccl().broadcast(buf_dev, buf_dev, nbytes, psh::kNcclUint8, root, rccl().comm, c.stream))
    return psh::rccl_fail("ncclBroadcast", rc);
  return PSH_OK;
}

int psh_comm_allgather(const void *send_dev, void *recv_dev, size_t nbytes_per_rank) {
  PSH_REQUIRE_INIT();
  if (!rccl().comm) return fail(PSH_ECOMM, "psh_comm_allgather: communicator not initialised");
  if ((!send_dev || !recv_dev) && nbytes_per_rank) return fail(PSH_EINVAL, "psh_comm_allgather: NULL buffer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  if (nbytes_per_rank == 0) return PSH_OK;
  if (int rc = rccl().all_gather(send_dev, recv_dev, nbytes_per_rank, psh::kNcclUint8, rccl().comm, c.stream))
    return psh::rccl_fail("ncclAllGather", rc);
  return PSH_OK;
}

int psh_comm_allreduce_f32(float *buf_dev, size_t count, int op) {
  PSH_REQUIRE_INIT();
  if (!rccl().comm) return fail(PSH_ECOMM, "psh_comm_allreduce: communicator not initialised");
  if (!buf_dev && count) return fail(PSH_EINVAL, "psh_comm_allreduce: NULL buffer");
  const int red = op == PSH_COMM_MIN ? psh::kNcclMin : op == PSH_COMM_MAX ? psh::kNcclMax : op == PSH_COMM_SUM ? psh::kNcclSum : -1;
  if (red < 0) return fail(PSH_EINVAL, "psh_comm_allreduce: unknown reduction %d", op);
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  if (count == 0) return PSH_OK;
  if (int rc = rccl().all_reduce(buf_dev, buf_dev, count, psh::kNcclFloat32, red, rccl().comm, c.stream))
    return psh::rccl_fail("ncclAllReduce", rc);
  return PSH_OK;
}

int psh_comm_destroy(void) {
  if (!rccl().comm) return PSH_OK;
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  if (c.ready) {
    (void)hipSetDevice(c.device);
    (void)hipStreamSynchronize(c.stream);
  }
  const int rc = rccl().comm_destroy(rccl().comm);
  rccl().comm = nullptr;
  rccl().nranks = 0;
  rccl().rank = -1;
  if (rc) return psh::rccl_fail("ncclCommDestroy", rc);
  return PSH_OK;
}

}  // extern "C"
