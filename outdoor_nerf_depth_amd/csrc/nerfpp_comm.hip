// Data-parallel gradient averaging behind the C ABI (nerfpp_allreduce_mean; reference:
// nerf-methods/nerfplusplus/ddp_train_nerf.py:323 -- DistributedDataParallel averages the gradients of every parameter over
// the ranks, backend nccl).  RCCL is bound at the first call with dlopen("librccl.so.1"): the library loads and every other
// entry point works on a box without RCCL; nothing here falls back to a host-side reduction.
//
// One process per GPU; rank 0 creates the 128-byte unique id (nerfpp_rccl_unique_id), hands it to the other ranks over any
// channel the host has (torch.distributed's store, MPI, a file), every rank calls nerfpp_rccl_comm_init.  The all-reduce
// runs on the caller's stream: NerfppTrainer puts it on its update stream, under the next level's forward.
#include <dlfcn.h>
#include <mutex>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/nerfpp_hip.h"

namespace {

typedef int (*get_uid_fn)(void*);                                          // ncclGetUniqueId(ncclUniqueId*)
struct UniqueId { char internal[128]; };                                   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*comm_init_fn)(void**, int, UniqueId, int);                   // ncclCommInitRank(comm*, nranks, id BY VALUE, rank)
typedef int (*comm_destroy_fn)(void*);
typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*errstr_fn)(int);

struct Rccl {
  void* h = nullptr;
  get_uid_fn get_uid = nullptr;
  comm_init_fn comm_init = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  allreduce_fn allreduce = nullptr;
  errstr_fn errstr = nullptr;
  char err[256] = {0};
};
// bound once per process, thread-safely (two threads may make the first call): r.err keeps the reason when it fails
Rccl g_rccl;
std::once_flag g_rccl_once;
void bind_rccl() {
  Rccl& r = g_rccl;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.h) break;
  }
  if (!r.h) { snprintf(r.err, sizeof r.err, "dlopen(librccl.so.1) failed: %s", dlerror()); return; }
  r.get_uid = (get_uid_fn)dlsym(r.h, "ncclGetUniqueId");
  r.comm_init = (comm_init_fn)dlsym(r.h, "ncclCommInitRank");
  r.comm_destroy = (comm_destroy_fn)dlsym(r.h, "ncclCommDestroy");
  r.allreduce = (allreduce_fn)dlsym(r.h, "ncclAllReduce");
  r.errstr = (errstr_fn)dlsym(r.h, "ncclGetErrorString");
  if (!r.get_uid || !r.comm_init || !r.comm_destroy || !r.allreduce) {
    snprintf(r.err, sizeof r.err, "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
    r.allreduce = nullptr;
  }
}
Rccl* rccl() {
  std::call_once(g_rccl_once, bind_rccl);
  return g_rccl.h && g_rccl.allreduce ? &g_rccl : nullptr;
}
thread_local char g_comm_err[320];
int comm_fail(const char* what, int rc) {
  Rccl* r = rccl();
  snprintf(g_comm_err, sizeof g_comm_err, "%s: RCCL error %d (%s)", what, rc, r && r->errstr ? r->errstr(rc) : "?");
  return NERFPP_ERR_COMM;
}
int no_rccl(const char* what) {
  snprintf(g_comm_err, sizeof g_comm_err, "%s: RCCL is not available in this process: %s", what,
           g_rccl.err[0] ? g_rccl.err : "librccl.so.1 could not be loaded");
  return NERFPP_ERR_COMM;
}

__global__ void scale_kernel(float* __restrict__ x, int64_t n, float s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= s;
}

}  // namespace

extern "C" {

const char* nerfpp_comm_last_error(void) { return g_comm_err; }

int nerfpp_rccl_unique_id(char out_id[128]) {
  if (!out_id) { snprintf(g_comm_err, sizeof g_comm_err, "nerfpp_rccl_unique_id: null buffer"); return NERFPP_ERR_ARG; }
  Rccl* r = rccl();
  if (!r) return no_rccl("nerfpp_rccl_unique_id");
  UniqueId id;
  const int rc = r->get_uid(&id);
  if (rc != 0) return comm_fail("ncclGetUniqueId", rc);
  memcpy(out_id, id.internal, 128);
  return NERFPP_OK;
}

int nerfpp_rccl_comm_init(void** comm, int world_size, const char id[128], int rank) {
  if (!comm || !id || world_size < 1 || rank < 0 || rank >= world_size) {
    snprintf(g_comm_err, sizeof g_comm_err, "nerfpp_rccl_comm_init: need comm, id, 0 <= rank < world_size");
    return NERFPP_ERR_ARG;
  }
  Rccl* r = rccl();
  if (!r) return no_rccl("nerfpp_rccl_comm_init");
  UniqueId u;
  memcpy(u.internal, id, 128);
  const int rc = r->comm_init(comm, world_size, u, rank);
  return rc == 0 ? NERFPP_OK : comm_fail("ncclCommInitRank", rc);
}

int nerfpp_rccl_comm_destroy(void* comm) {
  Rccl* r = rccl();
  if (!r) return no_rccl("nerfpp_rccl_comm_destroy");
  if (!comm) return NERFPP_OK;
  const int rc = r->comm_destroy(comm);
  return rc == 0 ? NERFPP_OK : comm_fail("ncclCommDestroy", rc);
}

int nerfpp_allreduce_mean(void* stream, void* rccl_comm, float* grads, int64_t count, int world_size, int prescaled) {
  if (!rccl_comm || !grads || count <= 0 || world_size < 1) {
    snprintf(g_comm_err, sizeof g_comm_err, "nerfpp_allreduce_mean: need a communicator, a buffer, count > 0, world_size >= 1");
    return NERFPP_ERR_ARG;
  }
  Rccl* r = rccl();
  if (!r) return no_rccl("nerfpp_allreduce_mean");
  hipStream_t st = (hipStream_t)stream;
  // mean = sum of the per-rank gradients / world_size.  prescaled != 0: the caller's gradients already carry the 1 / world_size
  // (nerfpp_backward_args.grad_scale does that inside the slab reduction), a SUM all-reduce finishes the mean.
  if (!prescaled && world_size > 1)
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, grads, count, 1.0f / (float)world_size);
  const int rc = r->allreduce(grads, grads, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, rccl_comm, st);
  if (rc != 0) return comm_fail("ncclAllReduce", rc);
  return hipGetLastError() == hipSuccess ? NERFPP_OK : NERFPP_ERR_LAUNCH;
}

}  // extern "C"
