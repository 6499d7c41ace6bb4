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
// `which` = 2: the communication stream of the native gradient all-reduce (dvgr_comm.hip), highest priority.
//
// WHICH HARDWARE QUEUE A STREAM GETS MATTERS (round 4, tools/micro/pipe_probe.hip): the command processor serves its hardware
// queues on 4 pipes, a queue's pipe is its creation index modulo 4, and two queues of ONE pipe that are both non-idle -- a
// kernel in flight, or a barrier packet waiting for another queue's signal -- cost every dispatch of the other ~34 us.  With
// at most four queues in the process (caller, side, query, + one) nothing collides; a fifth one (a torch stream of the
// whole model, the process group's stream, a loader's copy stream) does, and which pair depends on creation order: that is
// what rounds 2-3 recorded as "order matters and is not understood" (a step 2-3 x slower on the DEVICE).  The communication
// stream spends most of a step waiting, so dvgr_comm_place() (dvgr_comm.hip) picks its queue by MEASUREMENT against the
// streams that are busy during a step; set_shared_stream() installs the choice.
// ---- placement (dvgr_place.hip) ----
// importance of a stream when a new one must share a pipe with somebody: lower = more important (never disturbed first)
constexpr int kRankCaller = 0, kRankSide = 1, kRankHost = 2 /* streams of the host side: 2 .. 9 */, kRankQuery = 10, kRankComm = 11;
struct PlaceReport { int nc = 0, nb = 0, chosen = 0; long mask = 0; float base[8] = {}, us[4][8] = {}; };
bool placement_enabled();                 // DVGR_NO_STREAM_PLACEMENT=1: plain hipStreamCreateWithPriority, creation order decides
// A stream of `priority` on the hardware queue that disturbs the streams placed before it least (the caller's first, then by
// rank); `keep`: an existing stream that is candidate 0 (left idle if another one wins).  Registers the result under `rank`.
hipStream_t place_stream(int dev, int priority, hipStream_t caller, int rank, hipStream_t keep, PlaceReport* rep);
void note_placed(int dev, hipStream_t s, int rank);
inline void write_place_report(const PlaceReport& R, float* report, int report_len) {
  if (!report || report_len < 4) return;     // [candidates, streams measured against, chosen, collision mask, base[], us[0][], us[1][] ...]
  int w = 0;
  report[w++] = (float)R.nc; report[w++] = (float)R.nb; report[w++] = (float)R.chosen; report[w++] = (float)R.mask;
  for (int i = 0; i < R.nb && w < report_len; ++i) report[w++] = R.base[i];
  for (int k = 0; k < R.nc; ++k)
    for (int i = 0; i < R.nb && w < report_len; ++i) report[w++] = R.us[k][i];
}

inline hipStream_t (&shared_stream_table())[16][3] { static hipStream_t st[16][3] = {}; return st; }
inline std::mutex& shared_stream_mutex() { static std::mutex mu; return mu; }
// `caller`: the stream of the API call that needs the side / query stream first (they are placed against it)
inline hipStream_t shared_side_stream(int dev, int which = 0, bool create = true, hipStream_t caller = nullptr) {
  auto& st = shared_stream_table();
  std::lock_guard<std::mutex> lk(shared_stream_mutex());
  if (!create) return st[dev][which];
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (which == 2) {                            // (its queue is chosen later, by dvgr_comm_place, when every busy stream exists)
    if (!st[dev][2] && hipStreamCreateWithPriority(&st[dev][2], hipStreamNonBlocking, hi) != hipSuccess) st[dev][2] = nullptr;
    return st[dev][2];
  }
  // (both at the LOWEST priority.  Round 6 re-measured it -- the forward's query chain S0 on the side stream is on layer 0's
  // critical path -- with either or both at the highest: C2 0.9218-0.9251 against 0.9250 ms, C4 1.1197-1.1257 against 1.1216:
  // no difference, experiments_r06.tsv side_stream_priority_*)
  for (int w = 0; w <= which; ++w)             // side before query, always
    if (!st[dev][w]) st[dev][w] = place_stream(dev, lo, caller, w == 0 ? kRankSide : kRankQuery, nullptr, nullptr);
  return st[dev][which];
}
inline void set_shared_stream(int dev, int which, hipStream_t s) {
  std::lock_guard<std::mutex> lk(shared_stream_mutex());
  shared_stream_table()[dev][which] = s;
}
// The events only order streams of ONE device: no system-scope fence (host / peer visibility) at the record.
constexpr unsigned kEventFlags = hipEventDisableTiming | hipEventDisableSystemFence;
struct SideStream {
  hipStream_t st = nullptr;
  hipStream_t st2 = nullptr;      // query stream (see shared_side_stream)
  hipEvent_t ev[256] = {};
  int used = 0;
  hipStream_t caller = nullptr;   // the stream of the API call that asked (placement measures against it)
  int init(int dev, hipStream_t caller_) {
    caller = caller_;
    if (st) return 0;
    st = shared_side_stream(dev, 0, true, caller);
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
      st2 = shared_side_stream(dev, 1, true, caller);
    }
    return st2 ? st2 : st;
  }
  hipEvent_t next() {
    if (used >= 256) used = 0;
    if (!ev[used] && hipEventCreateWithFlags(&ev[used], kEventFlags) != hipSuccess) return nullptr;
    return ev[used++];
  }
  // order() for a fork right behind ONE launch: `e` was armed (arm_stop) before that launch; if it rode on the kernel's dispatch only
  // the wait is enqueued, else the ordinary record + wait
  int order_armed(hipStream_t from, hipStream_t to, hipEvent_t e, bool rode) {
    hipError_t r = rode ? hipSuccess : hipEventRecord(e, from);
    if (r == hipSuccess) r = hipStreamWaitEvent(to, e, 0);
    return (int)r;
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
inline SideStream* side_stream(hipStream_t caller) {
  static thread_local SideStream per_dev[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  SideStream* s = &per_dev[dev];
  s->used = 0;
  return s->init(dev, caller) == 0 ? s : nullptr;
}

}  // namespace dvgr
