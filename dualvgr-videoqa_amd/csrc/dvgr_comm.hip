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
  int inorder = 0;                 // dvgr_comm_set_inorder: collectives go onto the stream that produced the buffer
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

int dvgr_comm_place(dvgr_comm* c, void* const* busy_streams, int n_busy, float* report, int report_len) {
  if (!c) return DVGR_E_NULL;
  if (n_busy < 0 || n_busy > 5 || (n_busy && !busy_streams)) return DVGR_E_DIMS;
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return (int)e;
  static bool placed[16] = {};               // once per device: other communicators of the process share the chosen stream
  static std::mutex placed_mu;
  std::lock_guard<std::mutex> lk(placed_mu);
  if (placed[c->device]) {
    c->stream = dvgr::shared_side_stream(c->device, 2);
    if (report && report_len >= 4) { report[0] = 0.f; report[1] = 0.f; report[2] = 0.f; report[3] = 0.f; }
    return 0;
  }
  hipStream_t caller = n_busy ? (hipStream_t)busy_streams[0] : nullptr;
  for (int i = 1; i < n_busy; ++i) dvgr::note_placed(c->device, (hipStream_t)busy_streams[i], dvgr::kRankHost + i - 1);
  // every queue of the package exists before the candidates are created: the side and the query stream
  if (!dvgr::shared_side_stream(c->device, 1, true, caller)) return (int)hipErrorOutOfMemory;
  dvgr::PlaceReport R;
  hipStream_t s = dvgr::place_stream(c->device, /*priority: as created*/ [&]() { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); return hi; }(),
                                     caller, dvgr::kRankComm, c->stream, &R);
  if (!s) return (int)hipErrorOutOfMemory;
  c->stream = s;
  dvgr::set_shared_stream(c->device, 2, s);
  placed[c->device] = true;
  dvgr::write_place_report(R, report, report_len);
  return 0;         // a failed MEASUREMENT (R.mask < 0, reported as such) is not an error: candidate 0 is installed and usable
}

int dvgr_comm_create(const void* id, int rank, int world, dvgr_comm** out) {
  if (!id || !out) return DVGR_E_NULL;
  if (world < 1 || rank < 0 || rank >= world) return DVGR_E_DIMS;
  if (!api_loaded()) { const int e = dvgr_comm_load(nullptr); if (e) return e; }
  dvgr_comm* c = new dvgr_comm();
  c->world = world;
  hipError_t e = hipGetDevice(&c->device);
  if (e != hipSuccess || c->device < 0 || c->device >= 16) { delete c; return e != hipSuccess ? (int)e : DVGR_E_DIMS; }
  // the device's one communication stream (which hardware queue it should sit on: dvgr_comm_place)
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

int dvgr_comm_set_inorder(dvgr_comm* c, int on) {
  if (!c) return DVGR_E_NULL;
  c->inorder = on ? 1 : 0;
  return 0;
}

int dvgr_comm_allreduce_avg(dvgr_comm* c, float* buf, int64_t count, void* wait_event, void* after_stream) {
  if (!c || !buf) return DVGR_E_NULL;
  if (count <= 0) return 0;
  if (c->inorder) {
    // after_stream = the producing stream.  A wait_event is honoured as well (free when it was recorded on that same
    // stream): a caller of the event convention -- an event plus a NULL after_stream -- must not get an unordered collective.
    if (wait_event) {
      const hipError_t ew = hipStreamWaitEvent((hipStream_t)after_stream, (hipEvent_t)wait_event, 0);
      if (ew != hipSuccess) return (int)ew;
    }
    return nccl_rc(api.AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclAvg, c->comm, (hipStream_t)after_stream));
  }
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

int dvgr_comm_allreduce_avg_on(dvgr_comm* c, float* buf, int64_t count, void* stream) {
  if (!c || !buf) return DVGR_E_NULL;
  if (count <= 0) return 0;
  return nccl_rc(api.AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclAvg, c->comm, (hipStream_t)stream));
}

int dvgr_comm_device_stream(void** stream_out) {
  if (!stream_out) return DVGR_E_NULL;
  int dev = 0;
  const hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= 16) return DVGR_E_DIMS;
  *stream_out = (void*)dvgr::shared_side_stream(dev, 2);
  return *stream_out ? 0 : (int)hipErrorOutOfMemory;
}

int dvgr_comm_join(dvgr_comm* c, void* stream) {
  if (!c) return DVGR_E_NULL;
  if (c->inorder) return 0;                  // every collective sits in its producer's stream order already
  hipEvent_t ev = c->next();
  if (!ev) return (int)hipErrorOutOfMemory;
  hipError_t e = hipEventRecord(ev, c->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, ev, 0);
  return (int)e;
}

}  // extern "C"
