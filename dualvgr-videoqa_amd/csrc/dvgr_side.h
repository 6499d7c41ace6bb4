// dvgr_side.h -- the internal side stream shared by the schedules of dvgr_api.hip and dvgr_head.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace dvgr {

// Side stream of the backward pass: weight / bias gradients and the query branch are off the critical path
// (the dX chain), so they run on a second, lower-priority stream and fill the CUs the small critical-path
// kernels leave idle.  One per device, created on first use; events are recycled call after call (every wait
// on an event is enqueued right after its record, so a later re-record cannot be observed by it).
struct SideStream {
  hipStream_t st = nullptr;
  hipEvent_t ev[256] = {};
  int used = 0;
  int init() {
    if (st) return 0;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    return (int)hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo);
  }
  hipEvent_t next() {
    if (used >= 256) used = 0;
    if (!ev[used] && hipEventCreateWithFlags(&ev[used], hipEventDisableTiming) != hipSuccess) return nullptr;
    return ev[used++];
  }
  // everything enqueued on `from` so far happens before whatever is enqueued on `to` from now on
  int order(hipStream_t from, hipStream_t to) {
    hipEvent_t e = next();
    if (!e) return (int)hipErrorOutOfMemory;
    hipError_t r = hipEventRecord(e, from);
    if (r == hipSuccess) r = hipStreamWaitEvent(to, e, 0);
    return (int)r;
  }
};
inline SideStream* side_stream() {
  static thread_local SideStream per_dev[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  SideStream* s = &per_dev[dev];
  s->used = 0;
  return s->init() == 0 ? s : nullptr;
}

}  // namespace dvgr
