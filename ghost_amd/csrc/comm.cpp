// comm.cpp -- multi-GPU control plane over RCCL (xGMI), one process per GPU.
// The data path has no collective (channels are sharded); this file provides the
// one broadcast of the filter bank plus barrier / max-reduce for bench timing.
// RCCL is resolved with dlopen so that single-GPU users need no librccl.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../include/ghostcwt.h"

// minimal RCCL surface (matches rccl.h / nccl.h ABI)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { kNcclUint8 = 1, kNcclFloat64 = 8 };
enum { kNcclMax = 2 };

float2* gcwt_internal_bank_ptr(gcwt_plan* p, size_t* bytes);
hipStream_t gcwt_internal_stream(gcwt_plan* p);
int gcwt_internal_refresh_bank(gcwt_plan* p);

namespace {

thread_local std::string g_comm_err;

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                         "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.handle) break;
  }
  if (!r.handle) return r;
#define LOAD(field, sym) r.field = (decltype(r.field))dlsym(r.handle, sym)
  LOAD(GetUniqueId, "ncclGetUniqueId");
  LOAD(CommInitRank, "ncclCommInitRank");
  LOAD(CommDestroy, "ncclCommDestroy");
  LOAD(CommAbort, "ncclCommAbort");
  LOAD(Broadcast, "ncclBroadcast");
  LOAD(AllReduce, "ncclAllReduce");
  LOAD(GetErrorString, "ncclGetErrorString");
#undef LOAD
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Broadcast && r.AllReduce;
  return r;
}

// RCCL may print a banner on stdout when it initialises.  This library never touches the
// process's file descriptors to hide it (a redirect around a call that can block would
// leave fd 1 pointing elsewhere): a caller whose stdout is a protocol keeps a private
// copy of it and points fd 1 at stderr for the whole run, as bench.py does.

}  // namespace

struct gcwt_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1;
  hipStream_t stream = nullptr;
  double* d_val = nullptr;
  int device = -1;              // the device that was current when the communicator was made: every later call
                                // on it selects that device itself (a caller may have moved on to another)
};

// errors from this file are reported through the same gcwt_last_error() string
extern "C" const char* gcwt_last_error(void);
int gcwt_internal_set_error(int code, const char* msg);

namespace {
int cerr_(int code, const std::string& m) { return gcwt_internal_set_error(code, m.c_str()); }
int nccl_fail(const char* what, ncclResult_t r) {
  std::string m = std::string(what) + ": ";
  m += rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error";
  return cerr_(GCWT_ERR_COMM, m);
}
}  // namespace

extern "C" {

int gcwt_comm_unique_id(void* id128) {
  if (!id128) return cerr_(GCWT_ERR_INVALID, "NULL id buffer");
  Rccl& r = rccl();
  if (!r.ok) return cerr_(GCWT_ERR_COMM, "librccl not found or incomplete");
  ncclUniqueId id;
  ncclResult_t rc;
  rc = r.GetUniqueId(&id);
  if (rc != 0) return nccl_fail("ncclGetUniqueId", rc);
  static_assert(sizeof(ncclUniqueId) == GCWT_COMM_ID_BYTES, "unique id size");
  memcpy(id128, &id, sizeof(id));
  return GCWT_OK;
}

int gcwt_comm_create(gcwt_comm** out, int rank, int n_ranks, const void* id128) {
  if (!out || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks)
    return cerr_(GCWT_ERR_INVALID, "bad communicator arguments");
  *out = nullptr;
  Rccl& r = rccl();
  if (!r.ok) return cerr_(GCWT_ERR_COMM, "librccl not found or incomplete");
  gcwt_comm* c = new (std::nothrow) gcwt_comm();
  if (!c) return cerr_(GCWT_ERR_NOMEM, "out of host memory");
  // ncclCommInitRank binds the communicator to the calling thread's current device: take note of it (and make
  // sure there is one: without a device this is the caller's error, not RCCL's)
  if (hipGetDevice(&c->device) != hipSuccess || hipSetDevice(c->device) != hipSuccess) {
    (void)hipGetLastError();
    delete c;
    return cerr_(GCWT_ERR_NO_DEVICE, "no current HIP device for the communicator (gcwt_set_device first)");
  }
  c->rank = rank;
  c->n_ranks = n_ranks;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t rc;
  rc = r.CommInitRank(&c->comm, n_ranks, id, rank);
  if (rc != 0) { delete c; return nccl_fail("ncclCommInitRank", rc); }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void**)&c->d_val, sizeof(double)) != hipSuccess) {
    r.CommDestroy(c->comm);
    delete c;
    return cerr_(GCWT_ERR_HIP, "communicator scratch allocation failed");
  }
  *out = c;
  return GCWT_OK;
}

void gcwt_comm_destroy(gcwt_comm* c) {
  if (!c) return;
  if (c->device >= 0) (void)hipSetDevice(c->device);
  if (c->d_val) (void)hipFree(c->d_val);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->comm) rccl().CommDestroy(c->comm);
  delete c;
}

// After a failed collective: a peer may never arrive, so nothing here may wait on the
// communicator's stream -- ncclCommAbort tears the communicator down without completing what
// is in flight; the stream and the 8-byte scratch are left to process exit (hipFree and
// hipStreamDestroy synchronise).
void gcwt_comm_abort(gcwt_comm* c) {
  if (!c) return;
  if (c->comm) {
    if (rccl().CommAbort) rccl().CommAbort(c->comm);
    // (no ncclCommAbort in this librccl: leak the communicator rather than block in its destructor)
  }
  delete c;
}

int gcwt_comm_allreduce_max(gcwt_comm* c, double* value) {
  if (!c || !value) return cerr_(GCWT_ERR_INVALID, "NULL argument");
  if (hipSetDevice(c->device) != hipSuccess) return cerr_(GCWT_ERR_HIP, "the communicator's device cannot be selected");
  if (hipMemcpyAsync(c->d_val, value, sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess)
    return cerr_(GCWT_ERR_HIP, "copy to device failed");
  ncclResult_t rc;
  rc = rccl().AllReduce(c->d_val, c->d_val, 1, kNcclFloat64, kNcclMax, c->comm, c->stream);
  if (rc != 0) return nccl_fail("ncclAllReduce", rc);
  if (hipMemcpyAsync(value, c->d_val, sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess)
    return cerr_(GCWT_ERR_COMM_INCOMPLETE, "all-reduce was enqueued and did not complete");
  return GCWT_OK;
}

int gcwt_comm_barrier(gcwt_comm* c) {
  double v = 0.0;
  return gcwt_comm_allreduce_max(c, &v);
}

int gcwt_comm_broadcast_bank(gcwt_comm* c, gcwt_plan* plan, int root) {
  if (!c || !plan) return cerr_(GCWT_ERR_INVALID, "NULL argument");
  int rc = gcwt_plan_upload(plan);
  if (rc) return rc;
  // the plan's bank lives on the plan's device; the communicator must have been made on the same one
  hipPointerAttribute_t attr;
  size_t bytes = 0;
  float2* bank = gcwt_internal_bank_ptr(plan, &bytes);
  if (hipPointerGetAttributes(&attr, bank) == hipSuccess && attr.device != c->device)
    return cerr_(GCWT_ERR_INVALID, "the plan and the communicator are on different devices");
  if (hipSetDevice(c->device) != hipSuccess) return cerr_(GCWT_ERR_HIP, "the communicator's device cannot be selected");
  hipStream_t st = gcwt_internal_stream(plan);
  const ncclResult_t nr = rccl().Broadcast(bank, bank, bytes, kNcclUint8, root, c->comm, st);
  if (nr != 0) return nccl_fail("ncclBroadcast", nr);
  if ((rc = gcwt_internal_refresh_bank(plan))) return rc;   // signed gain table follows the bank
  if (hipStreamSynchronize(st) != hipSuccess)
    return cerr_(GCWT_ERR_COMM_INCOMPLETE, "broadcast was enqueued and did not complete");
  return GCWT_OK;
}

}  // extern "C"
