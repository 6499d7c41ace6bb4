// dvgr_side.h -- the internal side stream shared by the schedules of dvgr_api.hip and dvgr_head.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

namespace dvgr {

// Side stream of the backward pass: weight / bias gradients and the query branch are off the critical path
// (the dX chain), so they run on a second, lower-priority stream and fill the CUs the small critical-path
// kernels leave idle.  One per device, created on first use; events are recycled call after call (every wait
// on an event is enqueued right after its record, so a later re-record cannot be observed by it).
// The stream itself is one per device and process (forward in the caller's thread and backward in autograd's thread
// share it); the event pools are per thread.  The runtime maps streams onto its few hardware queues in creation order,
// and that mapping decides what a cross-stream dependency costs: dvgr_streams_init() lets the host side create this
// stream at a fixed place in the order of all the package's streams (dualvgr-videoqa_amd/_lib.py: streams()).
inline hipStream_t shared_side_stream(int dev) {
  static hipStream_t st[16] = {};
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (!st[dev]) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&st[dev], hipStreamNonBlocking, lo) != hipSuccess) st[dev] = nullptr;
  }
  return st[dev];
}
// The events only order streams of ONE device: no system-scope fence (host / peer visibility) at the record.
constexpr unsigned kEventFlags = hipEventDisableTiming | hipEventDisableSystemFence;
struct SideStream {
  hipStream_t st = nullptr;
  hipEvent_t ev[256] = {};
  int used = 0;
  int init(int dev) {
    if (st) return 0;
    st = shared_side_stream(dev);
    return st ? 0 : (int)hipErrorOutOfMemory;
  }
  hipEvent_t next() {
    if (used >= 256) used = 0;
    if (!ev[used] && hipEventCreateWithFlags(&ev[used], kEventFlags) != hipSuccess) return nullptr;
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
  return s->init(dev) == 0 ? s : nullptr;
}

}  // namespace dvgr
