// Per-phase timing of wave_tail_kernel (wall_clock64 = 100 MHz) for B = 256 streams, zero data.
#define TAIL_TIMING
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "wave_tail.hip.h"
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256;   // 256 = one workgroup per CU; 512 = two (as inside a tick launch)
  float *ring, *state, *w, *bias, *out; int* hop; unsigned long long* st;
  hipMalloc(&ring, B * 40 * 64 * 4); hipMalloc(&state, B * TAIL_STATE_FLOATS * 4); hipMalloc(&w, 1 << 20); hipMalloc(&bias, 4096);
  hipMalloc(&out, B * 240 * 4); hipMalloc(&hop, 4); hipMalloc(&st, B * 16 * 8);
  hipMemset(ring, 0, B * 40 * 64 * 4); hipMemset(state, 0, B * TAIL_STATE_FLOATS * 4); hipMemset(w, 0, 1 << 20); hipMemset(bias, 0, 4096); hipMemset(hop, 0, 4);
  TailArgs a{};
  a.in = Ring{ring, 64, 20, 2}; a.state = state; a.fin_w = w; a.fin_b = bias; a.d_out = out; a.hop = hop; a.stamps = st;
  for (int i = 0; i < 8; ++i) { a.w[i] = w + i * 20000; a.b[i] = bias; }
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(wave_tail_kernel<1>, dim3(B), dim3(tail::NTHR), 0, 0, a);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(B * 16);
  hipMemcpy(h.data(), st, B * 16 * 8, hipMemcpyDeviceToHost);
  const char* names[10] = {"prologue", "res2a", "res2b", "up3", "res3a", "res3b", "up4", "res4a", "res4b", "final+store"};
  for (int i = 0; i < 10; ++i) {
    double s = 0; for (int b = 0; b < B; ++b) s += (double)(h[b * 16 + i + 1] - h[b * 16 + i]);
    printf("%-12s %.2f us\n", names[i], s / B * 0.01);
  }
  double tot = 0; for (int b = 0; b < B; ++b) tot += (double)(h[b * 16 + 10] - h[b * 16]);
  printf("total        %.2f us\n", tot / B * 0.01);
  const char* sub[4] = {"MFMA loops", "epilogues", "history in + barrier wait", "history out"};
  for (int k = 0; k < 4; ++k) {
    double s = 0; for (int b = 0; b < B; ++b) s += (double)h[b * 16 + 11 + k];
    printf("  wavefront 0, layers res2a..res4b: %-26s %.0f shader cycles\n", sub[k], s / B);
  }
  return 0;
}
