// Per-phase timing of the attention half of a conditioned block (rc::block_b_body) as a launch of its own.
//   ./blockb_timing [rows] [speakers] [quads 0|1]
// rows streams, round-robin over `speakers` K/V slots; quads = 1: one workgroup per two quads of <= 4 rows of a slot (block_bq_body, 4x4x1
// multi-block MFMAs), 0: one per 16-row tile of a slot (block_b_body).  Zero data; prints launch time and wavefront 0's cycles per phase.
#define RC_TIMING
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rowchain.hip.h"
__global__ __launch_bounds__(512, 4) void probe(const rc::BlockBArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[rc::kBlockBLds];
  if (threadIdx.x == 0) { stepc::pair[0] = 0; stepc::pair[1] = 0; }
  __syncthreads();
  rc::block_b_body(a, blockIdx.x, lds);
}
__global__ __launch_bounds__(512, 4) void probe_q(const rc::BlockBqArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[rc::kBlockBqLds];
  if (threadIdx.x == 0) { stepc::pair[0] = 0; stepc::pair[1] = 0; }
  __syncthreads();
  rc::block_bq_body(a, blockIdx.x, lds);
}
int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 256, speakers = argc > 2 ? atoi(argv[2]) : 64, quads = argc > 3 ? atoi(argv[3]) : 1;
  const size_t kvf = (size_t)speakers * 256 * 384;
  float *xa, *out, *w, *bias, *kt, *v, *ktp, *vp; unsigned long long* st;
  hipMalloc(&xa, (size_t)rows * 2 * 256 * 4); hipMalloc(&out, (size_t)rows * 2 * 256 * 4); hipMalloc(&w, 4 << 20); hipMalloc(&bias, 4096);
  hipMalloc(&kt, kvf * 4); hipMalloc(&v, kvf * 4); hipMalloc(&ktp, kvf * 4); hipMalloc(&vp, kvf * 4);
  hipMemset(xa, 0, (size_t)rows * 2 * 256 * 4); hipMemset(w, 0, 4 << 20); hipMemset(bias, 0, 4096);
  hipMemset(kt, 0, kvf * 4); hipMemset(v, 0, kvf * 4); hipMemset(ktp, 0, kvf * 4); hipMemset(vp, 0, kvf * 4);
  // lists: every slot's rows as quads (quads = 1) or as tiles of 16 (quads = 0)
  std::vector<int> order(rows), slot_of(rows);
  for (int i = 0; i < rows; ++i) { order[i] = i; slot_of[i] = i % speakers; }
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return slot_of[x] < slot_of[y]; });
  const int nt = (rows + 15) / 16 + speakers;
  std::vector<int> perm((size_t)nt * 16, -1), tslot(nt, -1), qslot((size_t)nt * 4, -1);
  int used = 0;
  if (quads) {
    int quad = -1, fill = 4, cur = -1;
    for (int r : order) { if (slot_of[r] != cur || fill == 4) { ++quad; fill = 0; cur = slot_of[r]; qslot[quad] = cur; } perm[quad * 4 + fill++] = r; }
    used = quad / 2 + 1;   // two quads per workgroup
  } else {
    int tile = -1, fill = 16, cur = -1;
    for (int r : order) { if (slot_of[r] != cur || fill == 16) { ++tile; fill = 0; cur = slot_of[r]; tslot[tile] = cur; } perm[tile * 16 + fill++] = r; }
    used = tile + 1;
  }
  int *d_perm, *d_tslot, *d_qslot;
  hipMalloc(&d_perm, perm.size() * 4); hipMalloc(&d_tslot, tslot.size() * 4); hipMalloc(&d_qslot, qslot.size() * 4);
  hipMemcpy(d_perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_tslot, tslot.data(), tslot.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_qslot, qslot.data(), qslot.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&st, (size_t)nt * 4 * 16 * 8); hipMemset(st, 0, (size_t)nt * 4 * 16 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_rc_stamps), &st, sizeof(st));
  const rc::BlockBArgs a{Ring{xa, 256, 1, 2}, Ring{out, 256, 1, 2}, w, bias, w + 70000, bias, kt, v, d_perm, d_tslot, nullptr};
  const rc::BlockBqArgs aq{Ring{xa, 256, 1, 2}, Ring{out, 256, 1, 2}, w, bias, w + 70000, bias, ktp, vp, d_perm, d_qslot, nullptr};
  auto launch = [&] { if (quads) hipLaunchKernelGGL(probe_q, dim3(used), dim3(512), 0, 0, aq); else hipLaunchKernelGGL(probe, dim3(used), dim3(512), 0, 0, a); };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) launch();
  hipEventRecord(e0);
  for (int it = 0; it < 10; ++it) launch();
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)used * 16);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  printf("%d rows, %d speakers, %s: %d workgroups, %.1f us per launch\n  wavefront 0 cycles: lists+load | q | scores | softmax | P.V | out:", rows, speakers,
         quads ? "quads (block_bq_body)" : "tiles (block_b_body)", used, ms * 100.0);
  for (int i = 0; i < 6; ++i) { double s = 0; for (int g = 0; g < used; ++g) s += (double)(h[(size_t)g * 16 + i + 1] - h[(size_t)g * 16 + i]); printf(" %.0f", s / used); }
  printf("\n");
  return 0;
}
