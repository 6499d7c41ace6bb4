// Does a stream that WAITS (a barrier packet polling another queue's signal) slow down a BUSY stream, and does it depend on
// which pair of streams it is?  (Round-4 finding: a whole-model step with one more stream -- the communication stream, idle but
// for its waits -- took 2-3 x longer on the DEVICE, and throwaway streams created first changed that by 2 x.)
//
//   busy stream i : a chain of K dependent short kernels, timed by events on that stream
//   waiter j      : hipStreamWaitEvent on an event recorded behind the chain + one tiny kernel, enqueued while the chain runs
//
// Prints chain time [us] for every (i, j) and for no waiter.  hipcc --offload-arch=gfx950 -O2 pipe_probe.hip -o pipe_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_spin(float* p, int iters) {
  float a = p[threadIdx.x & 63];
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.678f) p[0] = a;
}
__global__ void k_tiny(float* p) { if (p[0] == -1.f) p[1] = 0.f; }

int main(int argc, char** argv) {
  const int NS = argc > 1 ? atoi(argv[1]) : 8;          // streams besides the null stream
  const int K = argc > 2 ? atoi(argv[2]) : 150;         // kernels in the chain
  const int iters = argc > 3 ? atoi(argv[3]) : 6000;    // ~ 15-20 us per kernel
  const char* prios = argc > 4 ? argv[4] : "nnnnnnnn";  // per stream: n(ormal) h(igh) l(ow)
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  float* buf;
  CK(hipMalloc(&buf, 1 << 20));
  CK(hipMemset(buf, 0, 1 << 20));
  std::vector<hipStream_t> st(NS + 1, nullptr);         // st[0] = the null stream
  for (int s = 1; s <= NS; ++s) {
    const char c = prios[(s - 1) % (int)strlen(prios)];
    CK(hipStreamCreateWithPriority(&st[s], hipStreamNonBlocking, c == 'h' ? hi : c == 'l' ? lo : 0));
    k_tiny<<<1, 64, 0, st[s]>>>(buf);                   // the stream's hardware queue exists from here on
  }
  CK(hipDeviceSynchronize());
  hipEvent_t t0, t1, done;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  // the waiter waits for ONE long single-workgroup kernel on the last stream (st[NS]) that outlasts the chain: it is blocked for the
  // whole measurement, whatever the busy stream does
  const int long_iters = argc > 5 ? atoi(argv[5]) : 3000000;
  const int grid = argc > 6 ? atoi(argv[6]) : 256;     // workgroups per chain kernel
  auto chain = [&](int i, int j) -> float {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipDeviceSynchronize();
      if (j >= 0) {
        k_spin<<<1, 64, 0, st[NS]>>>(buf + 4096, long_iters);
        (void)hipEventRecord(done, st[NS]);
        (void)hipStreamWaitEvent(st[j], done, 0);
        k_tiny<<<1, 64, 0, st[j]>>>(buf);
      }
      (void)hipEventRecord(t0, st[i]);
      for (int k = 0; k < K; ++k) k_spin<<<grid, 256, 0, st[i]>>>(buf + 1024, iters);
      (void)hipEventRecord(t1, st[i]);
      (void)hipDeviceSynchronize();
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, t0, t1);
      best = ms < best ? ms : best;
    }
    return best * 1e3f;
  };
  {
    (void)hipEventRecord(t0, st[NS]);
    k_spin<<<1, 64, 0, st[NS]>>>(buf + 4096, long_iters);
    (void)hipEventRecord(t1, st[NS]);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t0, t1);
    printf("the long kernel the waiter waits for: %.0f us\n", ms * 1e3f);
  }
  printf("streams: 0 = null"); for (int s = 1; s <= NS; ++s) printf(", %d = %c", s, prios[(s - 1) % (int)strlen(prios)]); printf("\n");
  printf("chain of %d kernels (%d workgroups, %d iterations each), time in us; rows: busy stream, columns: waiting stream ('-' = none)\n        -", K, grid, iters);
  for (int j = 0; j < NS; ++j) printf("%8d", j);
  printf("\n");
  for (int i = 0; i < NS; ++i) {
    printf("%2d %8.0f", i, chain(i, -1));
    for (int j = 0; j < NS; ++j) {
      if (j == i) { printf("       ."); continue; }
      printf("%8.0f", chain(i, j));
    }
    printf("\n");
  }
  return 0;
}
