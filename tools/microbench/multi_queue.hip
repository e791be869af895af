// Do dependent launches on several HIP streams proceed independently?  N streams, each replaying a hipGraph of
// 64 dependent kernels (trivial, or a 128-workgroup kernel that spins ~3 us), 20 replays per stream.
// Prints the aggregate launch rate and the time per launch per stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void trivial(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void busy(int* p, int spin) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
int main() {
  int* d; (void)hipMalloc(&d, 4096);
  for (int mode = 0; mode < 2; ++mode)
    for (int n : {1, 2, 3, 4}) {
      hipStream_t st[4]; hipGraph_t g[4]; hipGraphExec_t ge[4];
      for (int i = 0; i < n; ++i) {
        (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        (void)hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < 64; ++k) {
          if (mode == 0) hipLaunchKernelGGL(trivial, dim3(1), dim3(64), 0, st[i], d + 64 * i);
          else hipLaunchKernelGGL(busy, dim3(128), dim3(128), 0, st[i], d + 64 * i, 300);  // 300 ticks of 100 MHz = 3 us
        }
        (void)hipStreamEndCapture(st[i], &g[i]);
        (void)hipGraphInstantiate(&ge[i], g[i], nullptr, nullptr, 0);
      }
      for (int i = 0; i < n; ++i) (void)hipGraphLaunch(ge[i], st[i]);
      (void)hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < 20; ++r) for (int i = 0; i < n; ++i) (void)hipGraphLaunch(ge[i], st[i]);
      (void)hipDeviceSynchronize();
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("%s kernels, %d stream(s): %.2f us per launch per stream, %.0f k launches/s in total\n", mode ? "3 us" : "trivial", n,
             us / (20 * 64), n * 20 * 64 / us * 1e3);
      for (int i = 0; i < n; ++i) { (void)hipGraphExecDestroy(ge[i]); (void)hipGraphDestroy(g[i]); (void)hipStreamDestroy(st[i]); }
    }
  return 0;
}
