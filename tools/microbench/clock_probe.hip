// What shader clock do short dependent-MFMA kernels actually see?  One wave runs N dependent
// v_mfma_f32_16x16x4_f32 (40 cycles each when back-to-back dependent) and reports s_memtime ticks
// (shader cycles) and s_memrealtime ticks (100 MHz) -> effective MHz, both for one long kernel and
// for a graph of many short kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int n, unsigned long long* out, float* sink) {
  f32x4 acc = {0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  sink[blockIdx.x * 64 + threadIdx.x] = acc[0];
}
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned long long* d; float* sink; hipMalloc(&d, 16); hipMalloc(&sink, 1024 * 64 * 4);
  unsigned long long h[2];
  for (int blocks : {1, 256, 1024}) for (int n : {64, 1024, 65536}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    const int launches = n >= 65536 ? 1 : 64;
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, s, n, d, sink);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 5; ++w) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("blocks %4d  n=%6d: %.2f us/launch | cyc/mfma %.1f | realtime ticks/mfma %.3f -> %.0f MHz (if 100 MHz ref)\n", blocks, n,
           ms * 1000.f / (10 * launches), (double)h[0] / n, (double)h[1] / n, (double)h[0] / (double)h[1] * 100.0);
  }
  return 0;
}
