// Where does the ~8.6 us floor of an (empty) contraction launch come from?  Back-to-back launches of empty kernels that
// differ in workgroup size, static LDS and kernel-argument size.   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { char b[1900]; };
struct Small { int x; };
template <int T, int LDS, typename A>
__global__ __launch_bounds__(T) void k(const A a, float* out) {
  __shared__ char sm[LDS > 0 ? LDS : 4];
  if (LDS > 0 && threadIdx.x == 1023) sm[0] = 1;
  if (reinterpret_cast<const int*>(&a)[0] == 12345) out[0] = sm[0];
}
template <int T, int LDS, typename A>
void run(const char* name, int grid, float* out) {
  A a{};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<T, LDS, A>), dim3(grid), dim3(T), 0, 0, a, out);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int n = 400;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k<T, LDS, A>), dim3(grid), dim3(T), 0, 0, a, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-60s %6.2f us per launch\n", name, ms * 1e3 / n);
}
int main() {
  float* out;
  hipMalloc(&out, 64);
  run<64, 0, Small>("64 threads x 240, no LDS, 4 B args", 240, out);
  run<256, 0, Small>("256 threads x 240, no LDS, 4 B args", 240, out);
  run<768, 0, Small>("768 threads x 240, no LDS, 4 B args", 240, out);
  run<768, 65536, Small>("768 threads x 240, 64 KB LDS, 4 B args", 240, out);
  run<768, 131072, Small>("768 threads x 240, 128 KB LDS, 4 B args", 240, out);
  run<768, 131072, Big>("768 threads x 240, 128 KB LDS, 1900 B args", 240, out);
  run<768, 0, Big>("768 threads x 240, no LDS, 1900 B args", 240, out);
  run<256, 0, Big>("256 threads x 240, no LDS, 1900 B args", 240, out);
  run<768, 131072, Big>("768 threads x 1024, 128 KB LDS, 1900 B args", 1024, out);
  run<256, 0, Small>("256 threads x 4096, no LDS, 4 B args", 4096, out);
  return 0;
}
