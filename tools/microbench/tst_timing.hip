// Per-phase timing of the tail-stage bodies (tail_stages.hip.h) as launches of their own: B streams, zero data.
//   ./tst_timing [B] [copies]     copies > 1: that many launches' worth of workgroups in one grid (co-residency with itself)
#define TST_TIMING
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "conv_gemm.hip.h"
#include "tail_stages.hip.h"
template <class Op>
__global__ __launch_bounds__(512, 4) void probe(const tst::StageArgs a, int wgs) {
  __shared__ __attribute__((aligned(16))) float lds[Op::LDS_FLOATS];
  Op::run(a, blockIdx.x % wgs, 0, lds);
}
template <class Op>
void run(const char* name, tst::StageArgs a, int copies, unsigned long long* st, int n_phase) {
  const int wgs = Op::grid(a).x, total = wgs * copies;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(probe<Op>, dim3(total), dim3(512), 0, 0, a, wgs);
  hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(probe<Op>, dim3(total), dim3(512), 0, 0, a, wgs);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)total * 16);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  printf("%s: %d workgroups, %.2f us per launch; wavefront 0, shader cycles per phase:", name, total, ms * 100.0);
  for (int i = 0; i + 1 < n_phase; ++i) {
    double s = 0; for (int w = 0; w < total; ++w) s += (double)(h[(size_t)w * 16 + i + 1] - h[(size_t)w * 16 + i]);
    printf(" %.0f", s / total);
  }
  double s = 0; for (int w = 0; w < total; ++w) s += (double)(h[(size_t)w * 16 + n_phase - 1] - h[(size_t)w * 16]);
  printf("  total %.0f\n", s / total);
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, copies = argc > 2 ? atoi(argv[2]) : 1;
  float *ring_in, *ring_mid, *ring_out, *state, *w, *bias, *out; unsigned long long* st;
  hipMalloc(&ring_in, (size_t)B * 40 * 64 * 4); hipMalloc(&ring_mid, (size_t)B * 160 * 32 * 4); hipMalloc(&ring_out, (size_t)B * 480 * 16 * 4);
  hipMalloc(&state, (size_t)B * TAIL_STATE_FLOATS * 4); hipMalloc(&w, 1 << 20); hipMalloc(&bias, 4096); hipMalloc(&out, (size_t)B * 240 * 4);
  hipMalloc(&st, (size_t)B * copies * 16 * 8);
  hipMemset(ring_in, 0, (size_t)B * 40 * 64 * 4); hipMemset(ring_mid, 0, (size_t)B * 160 * 32 * 4); hipMemset(ring_out, 0, (size_t)B * 480 * 16 * 4);
  hipMemset(state, 0, (size_t)B * TAIL_STATE_FLOATS * 4); hipMemset(w, 0, 1 << 20); hipMemset(bias, 0, 4096);
  hipMemcpyToSymbol(HIP_SYMBOL(g_tst_stamps), &st, sizeof(st));
  int* hop; hipMalloc(&hop, 8); hipMemset(hop, 0, 8);
  tst::StageArgs a{};
  a.state = state; a.hop = hop; a.B = B; a.fin_w = w; a.fin_b = bias; a.d_out = out;
  for (int i = 0; i < 3; ++i) { a.w[i] = w + i * 30000; a.b[i] = bias; }
  printf("phases: prologue (loads + stores) | barrier | resA | barrier | resB | barrier | up | state out\n");
  a.in = Ring{ring_in, 64, 20, 2}; a.out = Ring{ring_mid, 32, 80, 2};
  run<tst::T1Op>("T1", a, copies, st, 9);
  a.in = Ring{ring_mid, 32, 80, 2}; a.out = Ring{ring_out, 16, 240, 2};
  run<tst::T2Op>("T2", a, copies, st, 9);
  return 0;
}
