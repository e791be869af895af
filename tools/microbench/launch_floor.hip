// Measures the per-launch floor of a dependent kernel chain on this box: eager vs hipGraph,
// trivial kernel vs "read a scalar + 64 KB of L2-resident data" kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_scalar(const int* p, float* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)p[0]; }
__global__ void k_touch(const int* p, const float* in, float* out, int n) {
  const int hop = *p;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + (float)hop;
}

template <class F>
float time_chain(hipStream_t s, int n_launch, int reps, bool graph, F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
  if (graph) {
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n_launch; ++i) launch(i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  }
  for (int w = 0; w < 3; ++w) { if (graph) hipGraphLaunch(ge, s); else for (int i = 0; i < n_launch; ++i) launch(i); }
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int r = 0; r < reps; ++r) { if (graph) hipGraphLaunch(ge, s); else for (int i = 0; i < n_launch; ++i) launch(i); }
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / (reps * n_launch);
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int* d_hop; float *d_a, *d_b; const int n = 64 * 1024;
  CK(hipMalloc(&d_hop, 4)); CK(hipMalloc(&d_a, n * 4)); CK(hipMalloc(&d_b, n * 4));
  CK(hipMemset(d_hop, 0, 4)); CK(hipMemset(d_a, 0, n * 4));
  for (int graph = 0; graph < 2; ++graph) {
    float t0 = time_chain(s, 64, 50, graph, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); });
    float t1 = time_chain(s, 64, 50, graph, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s); });
    float t2 = time_chain(s, 64, 50, graph, [&](int) { hipLaunchKernelGGL(k_scalar, dim3(1), dim3(64), 0, s, d_hop, d_b); });
    float t3 = time_chain(s, 64, 50, graph, [&](int i) { hipLaunchKernelGGL(k_touch, dim3(n / 256), dim3(256), 0, s, d_hop, (i & 1) ? d_a : d_b, (i & 1) ? d_b : d_a, n); });
    printf("%s: empty 1wg %.2f us | empty 256wg %.2f us | scalar-load %.2f us | touch 256KB dependent %.2f us  (per launch)\n",
           graph ? "graph" : "eager", t0, t1, t2, t3);
  }
  return 0;
}
