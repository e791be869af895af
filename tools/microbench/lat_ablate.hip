// Ablation of lat_gemm_kernel on one representative layer (256 -> 256 linear with residual,
// B = 256 rows): which part of the kernel costs what.  Chain of 64 launches in a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lat_gemm.hip.h"
namespace bhip { LaunchHook*& launch_hook() { static thread_local LaunchHook* h = nullptr; return h; } }
using LQ = Layer<256, 256, 1, 1, 1, 1, PRE_NONE, ACT_NONE, EPI_BIAS, true>;
using L5 = Layer<256, 256, 5, 1, 1, 1, PRE_NONE, ACT_GELU, EPI_BIAS, true>;

template <class L, int ABL>
float run(const ConvArgs& a0, const ConvArgs& a1, hipStream_t s) {
  dim3 grid((a0.B + 15) / 16, L::NOUT / 32);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 64; ++i) hipLaunchKernelGGL((lat_gemm_kernel<L, 2, ABL>), grid, dim3(128 * L::P), 0, s, (i & 1) ? a1 : a0);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / (20 * 64);
}

int main() {
  const int B = 256;
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  float *x, *y, *w, *bias; int* hop;
  hipMalloc(&x, B * 5 * 256 * 4); hipMalloc(&y, B * 5 * 256 * 4); hipMalloc(&w, 5 * 256 * 256 * 4); hipMalloc(&bias, 1024); hipMalloc(&hop, 4);
  hipMemset(x, 0, B * 5 * 256 * 4); hipMemset(y, 0, B * 5 * 256 * 4); hipMemset(w, 0, 5 * 256 * 256 * 4); hipMemset(bias, 0, 1024); hipMemset(hop, 0, 4);
  Ring rx{x, 256, 1, 5}, ry{y, 256, 1, 5};
  ConvArgs a0 = conv_args(rx, ry, w, bias, hop, B), a1 = conv_args(ry, rx, w, bias, hop, B);
  printf("linear 256->256 (P=1):  full %.2f | noW %.2f | noA %.2f | noMFMA %.2f | noEpiPrefetch %.2f | noHop %.2f | none(31) %.2f us\n",
         run<LQ, 0>(a0, a1, s), run<LQ, 1>(a0, a1, s), run<LQ, 2>(a0, a1, s), run<LQ, 4>(a0, a1, s), run<LQ, 8>(a0, a1, s),
         run<LQ, 16>(a0, a1, s), run<LQ, 31>(a0, a1, s));
  printf("conv k5 256->256 (P=5): full %.2f | noW %.2f | noA %.2f | noMFMA %.2f | noEpiPrefetch %.2f | noHop %.2f | none(31) %.2f us\n",
         run<L5, 0>(a0, a1, s), run<L5, 1>(a0, a1, s), run<L5, 2>(a0, a1, s), run<L5, 4>(a0, a1, s), run<L5, 8>(a0, a1, s),
         run<L5, 16>(a0, a1, s), run<L5, 31>(a0, a1, s));
  printf("conv k5 extra ablations: none(31) %.2f | none+noReduce(63) %.2f | none+noAstore(95) %.2f | none+both(127) %.2f us\n",
         run<L5, 31>(a0, a1, s), run<L5, 63>(a0, a1, s), run<L5, 95>(a0, a1, s), run<L5, 127>(a0, a1, s));
  printf("linear extra: none(31) %.2f | +noAstore(95) %.2f\n", run<LQ, 31>(a0, a1, s), run<LQ, 95>(a0, a1, s));
  return 0;
}
