// dvgr_side.h -- the internal side stream shared by the schedules of dvgr_api.hip and dvgr_head.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <stdlib.h>

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
// `which` = 0: the side stream proper; 1: the query stream of the unit's backward (the query branch is a chain of
// latency-bound kernels that must not queue behind the side stream's weight-gradient contractions).
inline hipStream_t shared_side_stream(int dev, int which = 0) {
  static hipStream_t st[16][2] = {};
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  for (int w = 0; w <= which; ++w)             // always created in the same order
    if (!st[dev][w]) {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (hipStreamCreateWithPriority(&st[dev][w], hipStreamNonBlocking, lo) != hipSuccess) st[dev][w] = nullptr;
    }
  return st[dev][which];
}
// The events only order streams of ONE device: no system-scope fence (host / peer visibility) at the record.
constexpr unsigned kEventFlags = hipEventDisableTiming | hipEventDisableSystemFence;
struct SideStream {
  hipStream_t st = nullptr;
  hipStream_t st2 = nullptr;      // query stream (see shared_side_stream)
  hipEvent_t ev[256] = {};
  int used = 0;
  int init(int dev) {
    if (st) return 0;
    st = shared_side_stream(dev, 0);
    return st ? 0 : (int)hipErrorOutOfMemory;
  }
  // the query stream, created on first use (DVGR_OPT_NO_QUERY_STREAM, or DVGR_NO_QUERY_STREAM=1 in a -DDVGR_TIMING_AIDS build: the side stream instead)
  hipStream_t query(bool allowed) {
#ifdef DVGR_TIMING_AIDS
    static const bool env_off = []() { const char* e = getenv("DVGR_NO_QUERY_STREAM"); return e && e[0] == '1'; }();
#else
    constexpr bool env_off = false;
#endif
    if (!allowed || env_off) return st;
    if (!st2) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return st;
      st2 = shared_side_stream(dev, 1);
    }
    return st2 ? st2 : st;
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
