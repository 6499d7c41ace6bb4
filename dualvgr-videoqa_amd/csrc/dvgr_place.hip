// dvgr_place.hip -- which hardware queue a stream of this package sits on, chosen by measurement (dvgr_side.h).
//
// The command processor serves a process's hardware queues on 4 pipes (a queue's pipe = its creation index mod 4).  While the
// HEAD PACKET of one queue is blocked -- a dispatch behind a kernel in flight, a barrier waiting for another queue's signal --
// every dispatch of the pipe's other queue pays ~25-30 us (tools/micro/pipe_probe.hip: 10-us kernels 1.7 -> 6.3 ms per 150,
// whatever the priorities).  The streams that are busy during a step are in that state nearly all the time (the host runs
// ahead), so two of them on one pipe cost every launch of both: a whole-model step went from 5.3 to 8.7 ms just because a
// torch.distributed process group had been initialised first and shifted the creation indices of the later streams, and to
// 14 ms with a fifth stream.  Rounds 2-3 knew this as "stream creation order matters, not understood".
//
// place_stream(): up to four candidate streams of the wanted priority (new hardware queues on consecutive pipes, or -- when
// the runtime's per-priority pool of GPU_MAX_HW_QUEUES queues is full -- its existing ones in turn); each candidate is put in
// the blocked-head state for 0.8 ms while a chain of 24 empty dispatches is timed on every stream placed before (the caller's
// stream first); the candidate that disturbs none of them -- or only the least important ones -- is used, the others stay idle.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "../../include/dvgr.h"
#include "dvgr_side.h"

namespace dvgr {
namespace {

__global__ void k_place_hold(unsigned long long ticks, int* sink) {      // ONE wave in flight for `ticks` of the 100 MHz clock
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  if (ticks == ~0ull && sink) *sink = 1;
}
__global__ void k_place_tiny(int* sink) { if (sink && threadIdx.x == 4096) *sink = 1; }
__global__ void k_place_link(int* sink) {                                // one link of the timed chain: ~3 us in flight (an EMPTY kernel
  const unsigned long long t0 = wall_clock64();                          // pays ~2.5 us per dispatch on a shared pipe, a real one ~25)
  while (wall_clock64() - t0 < 300) __builtin_amdgcn_s_sleep(2);
  if (sink && threadIdx.x == 4096) *sink = 1;
}
constexpr int kChain = 24;             // dispatches per timed chain
constexpr float kHoldMs = 1.0f;        // the candidate's head packet stays blocked this long (covers a slowed chain: 24 x <= 35 us)
// Measured: an undisturbed chain of 24 links takes 91-92 us (+-2), one beside a blocked queue of the SAME PIPE 223-237 us,
// one queued on the SAME HARDWARE QUEUE as the candidate 400-640 us (under rocprofv3: 170-250, 297, 116).

struct Placed { hipStream_t s; int rank; };
std::vector<Placed>& registry(int dev) { static std::vector<Placed> r[16]; return r[dev]; }
std::mutex& place_mutex() { static std::mutex m; return m; }
struct Idle { hipStream_t s; int priority; };
std::vector<Idle>& idle_pool(int dev) { static std::vector<Idle> p[16]; return p[dev]; }   // under place_mutex()

hipError_t chain_us(hipStream_t s, hipEvent_t t0, hipEvent_t t1, float* out_us) {
  hipError_t r = hipEventRecord(t0, s);
  for (int k = 0; k < kChain && r == hipSuccess; ++k) { k_place_link<<<1, 64, 0, s>>>(nullptr); r = hipGetLastError(); }
  if (r == hipSuccess) r = hipEventRecord(t1, s);
  if (r == hipSuccess) r = hipEventSynchronize(t1);
  float ms = 0.f;
  if (r == hipSuccess) r = hipEventElapsedTime(&ms, t0, t1);
  *out_us = ms * 1e3f;
  return r;
}

}  // namespace

bool placement_enabled() {
  static const bool on = []() { const char* e = getenv("DVGR_NO_STREAM_PLACEMENT"); return !(e && e[0] == '1'); }();
  return on;
}

void note_placed(int dev, hipStream_t s, int rank) {
  std::lock_guard<std::mutex> lk(place_mutex());
  auto& reg = registry(dev);
  for (auto& p : reg) if (p.s == s) { p.rank = std::min(p.rank, rank); return; }
  reg.push_back({s, rank});
}

hipStream_t place_stream(int dev, int priority, hipStream_t caller, int rank, hipStream_t keep, PlaceReport* rep) {
  std::lock_guard<std::mutex> lk(place_mutex());
  PlaceReport local;
  PlaceReport& R = rep ? *rep : local;
  R = PlaceReport();
  auto plain = [&]() -> hipStream_t {
    hipStream_t s = keep;
    if (!s && hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority) != hipSuccess) return nullptr;
    registry(dev).push_back({s, rank});
    return s;
  };
  if (!placement_enabled()) return plain();
  // No probe inside a capture -- on the caller's stream or, as far as this thread can tell, on any other: the probe
  // synchronises the device, which invalidates a relaxed-mode capture elsewhere.  (hipThreadExchangeStreamCaptureMode swaps
  // this thread's mode; in any mode but relaxed a device synchronise during a capture is an error the probe must not cause,
  // so the check is: the caller's stream is not capturing AND a harmless legacy-stream query succeeds.)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(caller, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return plain();
  if (hipStreamQuery(nullptr) == hipErrorStreamCaptureImplicit) { (void)hipGetLastError(); return plain(); }   // a global-mode capture is active
  // at most kMaxCandidates candidate streams are ever created per device (the losers of a placement are kept alive, see
  // below): later placements measure the idle losers again instead of creating more hardware queues
  static int created[16] = {};
  constexpr int kMaxCandidates = 12;
  // the streams placed before, most important first: the caller's, then by rank
  std::vector<Placed> busy;
  busy.push_back({caller, kRankCaller});
  for (const Placed& p : registry(dev))
    if (p.s != caller && p.s != keep) busy.push_back(p);
  std::stable_sort(busy.begin(), busy.end(), [](const Placed& a, const Placed& b) { return a.rank < b.rank; });
  if (busy.size() > 8) busy.resize(8);
  const int nb = (int)busy.size();
  hipStream_t cand[4] = {keep, nullptr, nullptr, nullptr};
  int nc = keep ? 1 : 0;
  {
    auto& pool = idle_pool(dev);                 // losers of earlier placements, same priority: candidates again before new queues
    for (size_t i = 0; i < pool.size() && nc < 4;) {
      if (pool[i].priority == priority) { cand[nc++] = pool[i].s; pool.erase(pool.begin() + i); } else ++i;
    }
  }
  while (nc < 4 && created[dev] < kMaxCandidates && hipStreamCreateWithPriority(&cand[nc], hipStreamNonBlocking, priority) == hipSuccess) {
    ++nc;
    ++created[dev];
  }
  if (nc == 0 && hipStreamCreateWithPriority(&cand[0], hipStreamNonBlocking, priority) == hipSuccess) nc = 1;   // (cap reached, pool empty)
  if (nc == 0) return nullptr;
  for (int k = 0; k < nc; ++k) k_place_tiny<<<1, 64, 0, cand[k]>>>(nullptr);      // (a stream's hardware queue exists from its first use on)
  for (int i = 0; i < nb; ++i) k_place_tiny<<<1, 64, 0, busy[i].s>>>(nullptr);
  hipError_t e = hipDeviceSynchronize();
  hipEvent_t t0 = nullptr, t1 = nullptr;
  if (e == hipSuccess) e = hipEventCreate(&t0);
  if (e == hipSuccess) e = hipEventCreate(&t1);
  for (int i = 0; i < nb && e == hipSuccess; ++i) {          // (the first chains after a pause run slow: one untimed, best of three)
    float a = 0.f;
    e = chain_us(busy[i].s, t0, t1, &a);
    R.base[i] = 1e30f;
    for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) { e = chain_us(busy[i].s, t0, t1, &a); R.base[i] = std::min(R.base[i], a); }
  }
  // PAIRED measurements: the chain beside the blocked candidate and, right behind it, the same chain with the candidate idle;
  // five pairs, the median of their ratios decides.  (Under a profiler the undisturbed chain drifts between 170 and 250 us from one second to the next
  // while a pipe collision sits at a steady 297: judged against one reference taken at the start, one run in three kept a
  // colliding queue.)
  float ratio[4][8];
  for (int k = 0; k < nc && e == hipSuccess; ++k)
    for (int i = 0; i < nb && e == hipSuccess; ++i) {
      constexpr int kReps = 5;
      float r[kReps], blocked[kReps];
      for (int rep = 0; rep < kReps && e == hipSuccess; ++rep) {
        // the candidate in the state a busy stream is in for most of a step: a dispatch queued behind a kernel in flight
        k_place_hold<<<1, 64, 0, cand[k]>>>((unsigned long long)(kHoldMs * 1e5f), nullptr);
        k_place_tiny<<<1, 64, 0, cand[k]>>>(nullptr);
        float a = 0.f, f = 1.f;
        e = chain_us(busy[i].s, t0, t1, &a);
        if (e == hipSuccess) e = hipStreamSynchronize(cand[k]);
        if (e == hipSuccess) e = chain_us(busy[i].s, t0, t1, &f);
        blocked[rep] = a;
        r[rep] = a / (f > 1.f ? f : 1.f);
      }
      std::sort(r, r + kReps);
      std::sort(blocked, blocked + kReps);
      ratio[k][i] = r[kReps / 2];
      R.us[k][i] = blocked[kReps / 2];
    }
  if (t0) (void)hipEventDestroy(t0);
  if (t1) (void)hipEventDestroy(t1);
  int best = 0;
  long best_key = -1;
  if (e == hipSuccess) {
    // disturbed = the MEDIAN over five pairs of (chain beside the blocked candidate) / (the same chain right afterwards, candidate
    // idle) is above 1.12 -- one pipe: unprofiled 225 / 92 us = 2.4, under rocprofv3 297 / 170-250 = 1.2-1.7 -- or below 0.8:
    // the candidate sits on the stream's very hardware queue, the chain queues behind the hold kernel and then runs back to
    // back (116 / 176 us under rocprofv3; unprofiled that case shows as 400-640 us)
    for (int k = 0; k < nc; ++k) {
      long key = 0;
      for (int i = 0; i < nb; ++i) key = key * 2 + ((ratio[k][i] > 1.12f || ratio[k][i] < 0.8f) ? 1 : 0);
      if (best_key < 0 || key < best_key) { best_key = key; best = k; }
    }
  }
  // The losing candidates are NOT destroyed: the runtime destroys a hardware queue with its last stream, the driver then
  // rebuilds the process's run list, and the pipe of every later queue -- the one just measured included -- moves (measured:
  // with the losers destroyed the plain C2 step went from 1.0 to 2.4 ms).  An idle stream costs nothing.
#ifdef DVGR_TIMING_AIDS
  if (const char* f = getenv("DVGR_PLACE_FORCE")) {          // probe: "<side><query><comm><host>" digits, e.g. 0000 = always candidate 0
    const int slot = rank == kRankSide ? 0 : rank == kRankQuery ? 1 : rank == kRankComm ? 2 : 3;
    if ((int)strlen(f) > slot && f[slot] >= '0' && f[slot] < '0' + nc) best = f[slot] - '0';
  }
#endif
  R.nc = nc; R.nb = nb; R.chosen = best; R.mask = e == hipSuccess ? best_key : -1;
  if (getenv("DVGR_PLACE_LOG")) {
    fprintf(stderr, "[dvgr place] rank %d: %d candidates against %d streams, chosen %d, mask %ld; base", rank, nc, nb, best, best_key);
    for (int i = 0; i < nb; ++i) fprintf(stderr, " %.0f", R.base[i]);
    for (int k = 0; k < nc; ++k) { fprintf(stderr, " | c%d:", k); for (int i = 0; i < nb; ++i) fprintf(stderr, " %.0f", R.us[k][i]); }
    fprintf(stderr, "\n");
  }
  for (int k = 0; k < nc; ++k)
    if (k != best && cand[k] != keep) idle_pool(dev).push_back({cand[k], priority});
  registry(dev).push_back({cand[best], rank});
  return cand[best];
}

}  // namespace dvgr

extern "C" {

int dvgr_stream_create_placed(int priority_class, int rank, void* caller_stream, void** stream_out, float* report, int report_len) {
  if (!stream_out) return DVGR_E_NULL;
  int dev = 0;
  const hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= 16 || priority_class < -1 || priority_class > 1) return DVGR_E_DIMS;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  dvgr::PlaceReport R;
  hipStream_t s = dvgr::place_stream(dev, priority_class < 0 ? hi : priority_class > 0 ? lo : 0, (hipStream_t)caller_stream,
                                     rank < dvgr::kRankHost ? dvgr::kRankHost : rank, nullptr, &R);
  if (!s) return (int)hipErrorOutOfMemory;
  *stream_out = (void*)s;
  dvgr::write_place_report(R, report, report_len);
  return 0;
}

}  // extern "C"
