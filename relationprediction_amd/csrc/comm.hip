// RCCL binding for the relation-sharded path (new: the reference is single-device).
// Per layer and direction the partial [V,d] sums are REDUCE-SCATTERED over the row shards (every rank finishes its own
// rows: relu / relu' / dropout on V/N rows instead of V), then the finished rows are ALL-GATHERED on a side stream
// while the next self-loop GEMM -- which needs only the rank's own rows -- already runs; the replicated weight
// gradients of all layers travel in ONE all-reduce at the end of the backward pass.  librccl.so.1 is dlopen'ed on
// first use so that a single-GPU context never depends on it; if another RCCL client in the
// process (e.g. torch) already mapped a librccl.so.1, the loader hands back that same image.
#include <dlfcn.h>

#include <cstdio>

#include "rgcn_internal.h"

namespace rgcn {

namespace {

typedef struct { char internal[128]; } nccl_unique_id;   // NCCL_UNIQUE_ID_BYTES = 128 (rccl.h:40-43)
typedef void* nccl_comm_t;
constexpr int kNcclSum = 0;        // rccl.h:448
constexpr int kNcclFloat32 = 7;    // rccl.h:466

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(nccl_unique_id*) = nullptr;
  int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(nccl_comm_t, int*) = nullptr;        // optional (rgcn_comm_info)
  int (*CommUserRank)(nccl_comm_t, int*) = nullptr;
  int (*CommCuDevice)(nccl_comm_t, int*) = nullptr;
  std::string error;
};

Rccl& rccl() {
  static Rccl r;
  return r;
}

bool load_rccl(std::string* err) {
  Rccl& r = rccl();
  if (r.handle) return true;
#ifdef RGCN_DEVTOOLS
  // librgcn_devtools.so only (a test seam, not in the product library): RGCN_RCCL_LIBRARY names another library with the
  // same entry points -- the multi-process tests use it to put several ranks on one GPU, which RCCL refuses
  if (const char* over = getenv("RGCN_RCCL_LIBRARY")) {
    r.handle = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    if (!r.handle) {
      *err = std::string("cannot dlopen RGCN_RCCL_LIBRARY=") + over + ": " + dlerror();
      return false;
    }
  }
#endif
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (r.handle) break;
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!r.handle) {
    *err = std::string("cannot dlopen librccl.so.1: ") + dlerror();
    return false;
  }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
  r.ReduceScatter = (decltype(r.ReduceScatter))dlsym(r.handle, "ncclReduceScatter");
  r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
  r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");
  r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.handle, "ncclCommUserRank");
  r.CommCuDevice = (decltype(r.CommCuDevice))dlsym(r.handle, "ncclCommCuDevice");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.ReduceScatter || !r.AllGather ||
      !r.GetErrorString) {
    *err = "librccl.so.1 lacks an expected nccl* symbol";
    dlclose(r.handle);
    r.handle = nullptr;
    return false;
  }
  return true;
}

std::string nccl_err(const char* what, int code) {
  return std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(code) : "?");
}

}  // namespace

rgcn_status comm_unique_id(uint8_t id[128]) {
  std::string err;
  if (!load_rccl(&err)) { set_global_error(err); return RGCN_ERR_RCCL; }
  nccl_unique_id u;
  int rc = rccl().GetUniqueId(&u);
  if (rc != 0) { set_global_error(nccl_err("ncclGetUniqueId", rc)); return RGCN_ERR_RCCL; }
  memcpy(id, u.internal, 128);
  return RGCN_OK;
}

rgcn_status comm_init(rgcn_ctx* c, const uint8_t id[128]) {
  if (c->comm) RGCN_FAIL(c, RGCN_ERR_STATE, "communicator already initialised");
  std::string err;
  if (!load_rccl(&err)) RGCN_FAIL(c, RGCN_ERR_RCCL, err);
  RGCN_HIP(c, hipSetDevice(c->cfg.device));
  nccl_unique_id u;
  memcpy(u.internal, id, 128);
  nccl_comm_t comm = nullptr;
  int rc = rccl().CommInitRank(&comm, c->world, u, c->rank);
  // with NCCL_DEBUG=VERSION (set on some hosts) RCCL leaves its banner in the C stdout buffer; push it out now,
  // so that it cannot surface after whatever the caller prints last (bench.py's one JSON line)
  fflush(stdout);
  if (rc != 0) RGCN_FAIL(c, RGCN_ERR_RCCL, nccl_err("ncclCommInitRank", rc));
  c->comm = comm;
  return RGCN_OK;
}

rgcn_status comm_allreduce(rgcn_ctx* c, float* buf, int64_t count) {
  if (!c->comm) RGCN_FAIL(c, RGCN_ERR_STATE, "no communicator: call rgcn_comm_init first");
  if (count <= 0) return RGCN_OK;
  ProfScope ps(c, "rccl_allreduce", 8.0 * count, 0);
  int rc = rccl().AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, (nccl_comm_t)c->comm, c->stream);
  if (rc != 0) RGCN_FAIL(c, RGCN_ERR_RCCL, nccl_err("ncclAllReduce", rc));
  return RGCN_OK;
}

// In place over a buffer of world * count floats: this rank's chunk [rank*count, +count) receives the sum of every
// rank's chunk.
rgcn_status comm_reduce_scatter(rgcn_ctx* c, float* buf, int64_t count) {
  if (!c->comm) RGCN_FAIL(c, RGCN_ERR_STATE, "no communicator: call rgcn_comm_init first");
  if (count <= 0) return RGCN_OK;
  ProfScope ps(c, "rccl_reduce_scatter", 4.0 * count * (c->world + 1), 0);
  int rc = rccl().ReduceScatter(buf, buf + (size_t)c->rank * count, (size_t)count, kNcclFloat32, kNcclSum,
                                (nccl_comm_t)c->comm, c->stream);
  if (rc != 0) RGCN_FAIL(c, RGCN_ERR_RCCL, nccl_err("ncclReduceScatter", rc));
  return RGCN_OK;
}

// In place over a buffer of world * count floats: every rank contributes its chunk [rank*count, +count).
rgcn_status comm_all_gather(rgcn_ctx* c, float* buf, int64_t count) {
  if (!c->comm) RGCN_FAIL(c, RGCN_ERR_STATE, "no communicator: call rgcn_comm_init first");
  if (count <= 0) return RGCN_OK;
  ProfScope ps(c, "rccl_all_gather", 4.0 * count * (c->world + 1), 0);
  int rc = rccl().AllGather(buf + (size_t)c->rank * count, buf, (size_t)count, kNcclFloat32, (nccl_comm_t)c->comm,
                            c->stream);
  if (rc != 0) RGCN_FAIL(c, RGCN_ERR_RCCL, nccl_err("ncclAllGather", rc));
  return RGCN_OK;
}

rgcn_status comm_info(rgcn_ctx* c, int32_t* ranks, int32_t* rank, int32_t* device) {
  int n = -1, r = -1, dev = -1;
  if (c->comm) {
    if (rccl().CommCount && rccl().CommCount((nccl_comm_t)c->comm, &n) != 0) n = -1;
    if (rccl().CommUserRank && rccl().CommUserRank((nccl_comm_t)c->comm, &r) != 0) r = -1;
    if (rccl().CommCuDevice && rccl().CommCuDevice((nccl_comm_t)c->comm, &dev) != 0) dev = -1;
  }
  if (ranks) *ranks = n;
  if (rank) *rank = r;
  if (device) *device = dev;
  return RGCN_OK;
}

void comm_destroy(rgcn_ctx* c) {
  if (c->comm && rccl().CommDestroy) rccl().CommDestroy((nccl_comm_t)c->comm);
  c->comm = nullptr;
}

}  // namespace rgcn
