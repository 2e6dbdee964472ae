// Gradient exchange over RCCL behind the C ABI: thin wrappers taking a flat fp32 bucket pointer
// ("DDP hook").  Replaces the reduce-add of nn.DataParallel (reference
// train.py:509-510).  librccl is opened lazily (dlopen), so the library loads on hosts without it and
// single-GPU use never touches it.  The default Python path drives the same collective through
// torch.distributed (backend "nccl" = RCCL); `parallel.GradSync(..., backend="rccl")` uses these.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {
// the few RCCL symbols used, with the ABI of rccl.h (ncclResult_t = int, ncclComm_t = opaque pointer,
// ncclUniqueId = 128 opaque bytes passed by value, ncclFloat32 = 7, ncclSum = 0)
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*GetErrorStringFn)(int);
struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle) return 0;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) {
    hrseg_set_error("hrseg_comm: cannot open librccl.so (%s)", dlerror());
    return HRSEG_ERR_UNSUPPORTED;
  }
  Rccl r;
  r.handle = h;
  r.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
  r.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
  r.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
  r.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
  r.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
    hrseg_set_error("hrseg_comm: librccl.so lacks an expected symbol");
    return HRSEG_ERR_UNSUPPORTED;
  }
  g_rccl = r;
  return 0;
}
int check(int rc, const char* what) {
  if (rc == 0) return 0;
  hrseg_set_error("hrseg_comm: %s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "rccl error");
  return HRSEG_ERR_LAUNCH;
}
}  // namespace

// 128 bytes identifying a new communicator; called by ONE rank, shipped to the others by the caller
extern "C" int hrseg_comm_unique_id(void* id128) {
  HRSEG_CHECK_ARG(id128 != nullptr, "hrseg_comm_unique_id: null buffer");
  if (int e = load_rccl()) return e;
  UniqueId id;
  if (int e = check(g_rccl.get_unique_id(&id), "ncclGetUniqueId")) return e;
  memcpy(id128, &id, sizeof(id));
  return 0;
}

// collective over all `world` ranks; the calling thread's current HIP device is the rank's GPU
extern "C" int hrseg_comm_init(void** comm, int rank, int world, const void* id128) {
  HRSEG_CHECK_ARG(comm && id128 && world >= 1 && rank >= 0 && rank < world, "hrseg_comm_init: bad arguments");
  if (int e = load_rccl()) return e;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  return check(g_rccl.comm_init_rank(comm, world, id, rank), "ncclCommInitRank");
}

// in-place sum over the ranks of `count` floats, stream-ordered on `stream` (returns at once)
extern "C" int hrseg_comm_allreduce_async(void* comm, float* buf, long count, hrseg_stream_t stream) {
  HRSEG_CHECK_ARG(comm && buf && count > 0, "hrseg_comm_allreduce_async: bad arguments");
  if (int e = load_rccl()) return e;
  return check(g_rccl.all_reduce(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, (hipStream_t)stream),
               "ncclAllReduce");
}

// everything queued on `comm_stream` so far (the collectives) happens before what `consumer` runs next;
// no host synchronisation
extern "C" int hrseg_comm_wait(hrseg_stream_t comm_stream, hrseg_stream_t consumer) {
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    hrseg_set_error("hrseg_comm_wait: hipEventCreate failed");
    return HRSEG_ERR_LAUNCH;
  }
  hipError_t e1 = hipEventRecord(ev, (hipStream_t)comm_stream);
  hipError_t e2 = hipStreamWaitEvent((hipStream_t)consumer, ev, 0);
  (void)hipEventDestroy(ev);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    hrseg_set_error("hrseg_comm_wait: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return HRSEG_ERR_LAUNCH;
  }
  return 0;
}

extern "C" int hrseg_comm_destroy(void* comm) {
  if (!comm) return 0;
  if (int e = load_rccl()) return e;
  return check(g_rccl.comm_destroy(comm), "ncclCommDestroy");
}
