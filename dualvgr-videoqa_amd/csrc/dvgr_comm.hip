// dvgr_comm.hip -- native gradient all-reduce of the unit over RCCL (include/dvgr_comm.h).  Host code only.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>

#include "../../include/dvgr.h"
#include "../../include/dvgr_comm.h"
#include "dvgr_side.h"

namespace {

// the slice of the NCCL / RCCL API this file uses (stable since NCCL 2.10: ncclAvg)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[DVGR_COMM_ID_BYTES]; } ncclUniqueId;
enum { ncclFloat32 = 7, ncclSum = 0, ncclAvg = 4 };
struct Api {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
} api;
std::mutex api_mu;

inline int nccl_rc(int r) { return r == 0 ? 0 : 1000 + r; }
inline bool api_loaded() { std::lock_guard<std::mutex> lk(api_mu); return api.handle != nullptr; }

}  // namespace

struct dvgr_comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev[64] = {};          // ordering events (stream -> stream on one device), recycled
  int used = 0;
  int world = 1;
  int device = 0;
  hipEvent_t next() {
    if (used >= 64) used = 0;
    if (!ev[used] && hipEventCreateWithFlags(&ev[used], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return nullptr;
    return ev[used++];
  }
};

extern "C" {

int dvgr_comm_load(const char* path) {
  std::lock_guard<std::mutex> lk(api_mu);
  if (api.handle) return 0;
  void* h = dlopen(path ? path : "librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return DVGR_E_NULL;
  Api a;
  a.GetUniqueId = (int (*)(ncclUniqueId*))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (int (*)(ncclComm_t))dlsym(h, "ncclCommDestroy");
  a.AllReduce = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))dlsym(h, "ncclAllReduce");
  a.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce) { dlclose(h); return DVGR_E_KIND; }
  a.handle = h;
  api = a;
  return 0;
}

int dvgr_comm_unique_id(void* id_out) {
  if (!id_out) return DVGR_E_NULL;
  if (!api_loaded()) { const int e = dvgr_comm_load(nullptr); if (e) return e; }
  ncclUniqueId id;
  const int r = api.GetUniqueId(&id);
  if (r) return nccl_rc(r);
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

int dvgr_comm_create(const void* id, int rank, int world, dvgr_comm** out) {
  if (!id || !out) return DVGR_E_NULL;
  if (world < 1 || rank < 0 || rank >= world) return DVGR_E_DIMS;
  if (!api_loaded()) { const int e = dvgr_comm_load(nullptr); if (e) return e; }
  dvgr_comm* c = new dvgr_comm();
  c->world = world;
  hipError_t e = hipGetDevice(&c->device);
  if (e != hipSuccess || c->device < 0 || c->device >= 16) { delete c; return e != hipSuccess ? (int)e : DVGR_E_DIMS; }
  // the device's one communication stream.  It must be created BEFORE the unit's side stream (dvgr_side.h: the runtime's
  // stream -> hardware-queue mapping follows creation order; the other order costs 3 x the step time)
  if (!dvgr::shared_side_stream(c->device, 2, false) && dvgr::shared_side_stream(c->device, 0, false)) { delete c; return DVGR_E_ORDER; }
  c->stream = dvgr::shared_side_stream(c->device, 2);
  if (!c->stream) { delete c; return (int)hipErrorOutOfMemory; }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  const int r = api.CommInitRank(&c->comm, world, uid, rank);
  if (r) { delete c; return nccl_rc(r); }
  *out = c;
  return 0;
}

int dvgr_comm_destroy(dvgr_comm* c) {
  if (!c) return 0;
  (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)api.CommDestroy(c->comm);
  for (hipEvent_t e : c->ev)
    if (e) (void)hipEventDestroy(e);
  delete c;                                  // (the stream belongs to the device, not to the communicator)
  return 0;
}

int dvgr_comm_world(const dvgr_comm* c) { return c ? c->world : 0; }

int dvgr_comm_allreduce_avg(dvgr_comm* c, float* buf, int64_t count, void* wait_event, void* after_stream) {
  if (!c || !buf) return DVGR_E_NULL;
  if (count <= 0) return 0;
  hipError_t e = hipSuccess;
  if (wait_event) {
    e = hipStreamWaitEvent(c->stream, (hipEvent_t)wait_event, 0);
  } else {
    hipEvent_t ev = c->next();
    if (!ev) return (int)hipErrorOutOfMemory;
    e = hipEventRecord(ev, (hipStream_t)after_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ev, 0);
  }
  if (e != hipSuccess) return (int)e;
  return nccl_rc(api.AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclAvg, c->comm, c->stream));
}

int dvgr_comm_device_stream(void** stream_out) {
  if (!stream_out) return DVGR_E_NULL;
  int dev = 0;
  const hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= 16) return DVGR_E_DIMS;
  if (!dvgr::shared_side_stream(dev, 2, false) && dvgr::shared_side_stream(dev, 0, false)) return DVGR_E_ORDER;
  *stream_out = (void*)dvgr::shared_side_stream(dev, 2);
  return *stream_out ? 0 : (int)hipErrorOutOfMemory;
}

int dvgr_comm_join(dvgr_comm* c, void* stream) {
  if (!c) return DVGR_E_NULL;
  hipEvent_t ev = c->next();
  if (!ev) return (int)hipErrorOutOfMemory;
  hipError_t e = hipEventRecord(ev, c->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, ev, 0);
  return (int)e;
}

}  // extern "C"
