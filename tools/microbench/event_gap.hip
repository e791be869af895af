// What does a cross-stream dependency cost between two graph launches?  Stream A replays a graph of 10 dependent
// 3-us kernels back to back 200 times; variants add, per replay, (b) an event record, (c) a wait on an event that
// another stream recorded long ago, (d) a hand-over A -> B -> A each replay (B runs one short kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void busy(int* p, int spin) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
int main() {
  int* d; (void)hipMalloc(&d, 4096);
  hipStream_t a, b; (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  hipGraph_t g; hipGraphExec_t ge, gb;
  (void)hipStreamBeginCapture(a, hipStreamCaptureModeThreadLocal);
  for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(busy, dim3(128), dim3(128), 0, a, d, 300);
  (void)hipStreamEndCapture(a, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraph_t g2;
  (void)hipStreamBeginCapture(b, hipStreamCaptureModeThreadLocal);
  hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, b, d + 64, 100);
  (void)hipStreamEndCapture(b, &g2); (void)hipGraphInstantiate(&gb, g2, nullptr, nullptr, 0);
  hipEvent_t e[4]; for (auto& x : e) (void)hipEventCreateWithFlags(&x, hipEventDisableTiming);
  (void)hipEventRecord(e[0], b); (void)hipDeviceSynchronize();
  const char* names[4] = {"graphs back to back", "+ event record after each", "+ wait on an old event of another stream", "+ hand-over A -> B -> A"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int r = 0; r < 5; ++r) (void)hipGraphLaunch(ge, a);
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 200; ++r) {
      if (mode == 2) (void)hipStreamWaitEvent(a, e[0], 0);
      if (mode == 3 && r > 0) (void)hipStreamWaitEvent(a, e[2], 0);
      (void)hipGraphLaunch(ge, a);
      if (mode >= 1) (void)hipEventRecord(e[1], a);
      if (mode == 3) { (void)hipStreamWaitEvent(b, e[1], 0); (void)hipGraphLaunch(gb, b); (void)hipEventRecord(e[2], b); }
    }
    (void)hipDeviceSynchronize();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%-44s %.1f us per replay (10 kernels of ~4 us)\n", names[mode], us / 200);
  }
  return 0;
}
