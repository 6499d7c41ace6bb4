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
// `which` = 2: the communication stream of the native gradient all-reduce (dvgr_comm.hip), highest priority, created by the
// first dvgr_comm_create() on the device.
// ORDER MATTERS and is measured, not understood: the runtime's mapping of streams onto hardware queues depends on creation
// order in a way GPU_MAX_HW_QUEUES (4, 8, 16: no change, tools/hwq_matrix.sh) does not control.  Known-good orders, all with
// the caller's stream first: [side, query on first use] (no communicator) and [communication, side, query] (communicator
// created BEFORE the unit's first call).  Known-bad: a communicator created after the unit has run (its stream last: a
// data-parallel step took 3.3 ms instead of 1.1 ms) and all three created together at dvgr_streams_init (the whole-model
// step took 10.9 ms instead of 5.1 ms).  dvgr_comm_create() therefore refuses (DVGR_E_ORDER) once the side stream exists.
inline hipStream_t shared_side_stream(int dev, int which = 0, bool create = true) {
  static hipStream_t st[16][3] = {};
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (!create) return st[dev][which];
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (which == 2) {
    if (!st[dev][2] && hipStreamCreateWithPriority(&st[dev][2], hipStreamNonBlocking, hi) != hipSuccess) st[dev][2] = nullptr;
    return st[dev][2];
  }
  for (int w = 0; w <= which; ++w)             // side before query, always
    if (!st[dev][w] && hipStreamCreateWithPriority(&st[dev][w], hipStreamNonBlocking, lo) != hipSuccess) st[dev][w] = nullptr;
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
