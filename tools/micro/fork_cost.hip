// What does a fork cost the stream that forks?  A chain of K kernels on stream A, and after every kernel one "this much is done"
// marker for another stream -- as a hipEvent record with different flags, as a hipStreamWriteValue32 -- with and without a second
// stream that actually waits for it and runs a tiny kernel.  Prints the added time per marker.
//   hipcc --offload-arch=gfx950 -O2 fork_cost.hip -o fork_cost.bin ;  ./fork_cost.bin [K=200] [iters=600]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_spin(float* p, int iters) {
  float a = p[threadIdx.x & 63];
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.678f) p[0] = a;
}
__global__ void k_tiny(float* p) { if (p[0] == -1.f) p[1] = 0.f; }

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 200;
  const int iters = argc > 2 ? atoi(argv[2]) : 600;
  float* buf;
  CK(hipMalloc(&buf, 1 << 20));
  CK(hipMemset(buf, 0, 1 << 20));
  uint32_t* flag;
  CK(hipMalloc(&flag, 4096));
  CK(hipMemset(flag, 0, 4096));
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t A, B;
  CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, 0));
  CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, lo));
  k_tiny<<<1, 64, 0, A>>>(buf);
  k_tiny<<<1, 64, 0, B>>>(buf);
  CK(hipDeviceSynchronize());
  const unsigned flagsets[3] = {hipEventDisableTiming | hipEventDisableSystemFence, hipEventDisableTiming, hipEventDefault};
  const char* flagnames[3] = {"DisableTiming|DisableSystemFence", "DisableTiming", "default (timing, system fence)"};
  std::vector<hipEvent_t> ev[3];
  for (int f = 0; f < 3; ++f) { ev[f].resize(K); for (int k = 0; k < K; ++k) CK(hipEventCreateWithFlags(&ev[f][k], flagsets[f])); }
  hipEvent_t t0, t1, done_ev;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  CK(hipEventCreateWithFlags(&done_ev, hipEventDisableTiming | hipEventDisableSystemFence));
  uint32_t epoch = 0;
  auto run = [&](int mode, int fset, bool waiter) -> float {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(t0, A);
      for (int k = 0; k < K; ++k) {
        if (mode == 4) {           // a JOIN: the chain's stream waits for an event of the other stream that completed long ago
          k_spin<<<256, 256, 0, A>>>(buf + 1024, iters);
          (void)hipStreamWaitEvent(A, done_ev, 0);
          continue;
        }
        if (mode == 3) {
          hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, nullptr, ev[fset][k], 0, buf + 1024, iters);
          if (waiter) { (void)hipStreamWaitEvent(B, ev[fset][k], 0); k_tiny<<<1, 64, 0, B>>>(buf); }
          continue;
        }
        k_spin<<<256, 256, 0, A>>>(buf + 1024, iters);
        if (mode == 1) {
          (void)hipEventRecord(ev[fset][k], A);
          if (waiter) { (void)hipStreamWaitEvent(B, ev[fset][k], 0); k_tiny<<<1, 64, 0, B>>>(buf); }
        } else if (mode == 3) {
          // the marker rides on the kernel's own dispatch packet: hipExtLaunchKernelGGL(..., stopEvent) -- no barrier packet between
          // this kernel and the next one of the chain
        } else if (mode == 2) {
          ++epoch;
          (void)hipStreamWriteValue32(A, flag, epoch, 0);
          if (waiter) { (void)hipStreamWaitValue32(B, flag, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu); k_tiny<<<1, 64, 0, B>>>(buf); }
        }
      }
      (void)hipEventRecord(t1, A);
      (void)hipDeviceSynchronize();
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, t0, t1);
      best = ms < best ? ms : best;
    }
    return best * 1e3f;
  };
  k_tiny<<<1, 64, 0, B>>>(buf);
  CK(hipEventRecord(done_ev, B));
  CK(hipDeviceSynchronize());
  const float base = run(0, 0, false);
  printf("chain of %d kernels (256 workgroups, %d iterations): %.0f us = %.2f us per kernel\n", K, iters, base, base / K);
  for (int f = 0; f < 3; ++f) {
    const float a = run(1, f, false), b = run(1, f, true);
    printf("  hipEventRecord after every kernel, flags %-34s: +%.2f us per record;  with a waiting stream: +%.2f us\n", flagnames[f],
           (a - base) / K, (b - base) / K);
  }
  for (int f = 0; f < 3; ++f) {
    const float a = run(3, f, false), b = run(3, f, true);
    printf("  stop event ON the kernel's launch (hipExtLaunchKernelGGL), flags %-34s: +%.2f us per kernel;  with a waiting stream: +%.2f us\n", flagnames[f],
           (a - base) / K, (b - base) / K);
  }
  {
    const float a = run(4, 0, false);
    printf("  hipStreamWaitEvent on the chain's stream for an event that completed long ago (a join): +%.2f us per wait\n", (a - base) / K);
  }
  {
    const float a = run(2, 0, false), b = run(2, 0, true);
    printf("  hipStreamWriteValue32 after every kernel                                 : +%.2f us per write;   with hipStreamWaitValue32 on the other stream: +%.2f us\n",
           (a - base) / K, (b - base) / K);
  }
  return 0;
}
