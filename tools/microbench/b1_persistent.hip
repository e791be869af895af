// b1_persistent.hip -- what a weight-stationary persistent kernel for ONE stream (BASELINE.json configs[1]) would pay for
// synchronisation, measured: (1) an all-to-all edge = every workgroup publishes its slab of a layer's output and every
// workgroup waits for all slabs (the per-layer dependency of a B = 1 chain whose layers are split by columns over the
// CUs), with and without an LDS-resident GEMV of the layer's size between edges; (2) the host <-> parked-kernel mailbox
// round trip through mapped host memory.  See profiles/r03_notes.md section 5 for the budget these numbers give.
//   ./b1_persistent [workgroups = 256]
// build: hipcc -O3 --offload-arch=gfx950 -o b1_persistent b1_persistent.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Granule { float v; unsigned tag; };   // one naturally aligned 8-byte store
// edges x { GEMV of K inputs against this workgroup's LDS-resident slab (K x 4 columns), publish 4 granules... here 1 },
// then sweep all G granules of the edge until every tag matches.
template <bool WORK>
__global__ __launch_bounds__(256) void chain_kernel(Granule* __restrict__ board /* [2][G] */, int G, int edges, int K, long long* cycles, float* sink) {
  __shared__ float w[1280 * 4];
  __shared__ float x[1280];
  const int tid = threadIdx.x, wg = blockIdx.x;
  for (int i = tid; i < 1280 * 4; i += 256) w[i] = 1e-4f * (float)((i * 7 + wg) & 255);
  for (int i = tid; i < 1280; i += 256) x[i] = 0.01f;
  __syncthreads();
  const long long t0 = clock64();
  float carry = 0.f;
  for (int e = 1; e <= edges; ++e) {
    float out = carry;
    if (WORK) {  // 4 columns x K: 64 lanes x 4 columns, k strided over the lanes of a wavefront, wave 0 only (the rest idle as in a GEMV)
      if (tid < 64) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int k = tid; k < K; k += 64) { const float xv = x[k]; a0 = fmaf(xv, w[k * 4], a0); a1 = fmaf(xv, w[k * 4 + 1], a1); a2 = fmaf(xv, w[k * 4 + 2], a2); a3 = fmaf(xv, w[k * 4 + 3], a3); }
        float s = a0 + a1 + a2 + a3;
        for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
        out += s;
      }
    }
    Granule* cur = board + (size_t)(e & 1) * G;
    if (tid == 0) {
      Granule g{out, (unsigned)e};
      __builtin_nontemporal_store(*reinterpret_cast<unsigned long long*>(&g), reinterpret_cast<unsigned long long*>(cur + wg));   // one 8-byte store
      __threadfence();
    }
    // sweep: wave 0 reads all granules until every tag == e, then the workgroup goes on
    if (tid < 64) {
      bool all;
      float acc;
      do {
        all = true; acc = 0.f;
        for (int i = tid; i < G; i += 64) {
          const unsigned long long raw = __atomic_load_n(reinterpret_cast<unsigned long long*>(cur + i), __ATOMIC_RELAXED);
          Granule g = *reinterpret_cast<const Granule*>(&raw);
          all = all && g.tag == (unsigned)e;
          acc += g.v;
        }
        all = __all(all);
      } while (!all);
      for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
      if (tid == 0) x[e % 1280] = acc * 1e-3f;
      carry = acc * 1e-6f;
    }
    __syncthreads();
  }
  const long long t1 = clock64();
  if (tid == 0) { cycles[wg] = t1 - t0; sink[wg] = carry; }
}

// parked kernel: one workgroup polls a mapped host word; on a new sequence number it copies 160 floats in, writes 240 out
// and publishes the sequence number back
__global__ void mailbox_kernel(volatile unsigned* req, volatile unsigned* ack, const float* in, float* out, int rounds) {
  for (int r = 1; r <= rounds; ++r) {
    if (threadIdx.x == 0) { while (__atomic_load_n(const_cast<const unsigned*>(req), __ATOMIC_ACQUIRE) != (unsigned)r) __builtin_amdgcn_s_sleep(1); }
    __syncthreads();
    if (threadIdx.x < 240) out[threadIdx.x] = in[threadIdx.x % 160] + 1.0f;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __atomic_store_n(const_cast<unsigned*>(ack), (unsigned)r, __ATOMIC_RELEASE);
  }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256;
  Granule* board; long long* cyc; float* sink;
  CK(hipMalloc(&board, sizeof(Granule) * 2 * G)); CK(hipMemset(board, 0, sizeof(Granule) * 2 * G));
  CK(hipMalloc(&cyc, 8 * G)); CK(hipMalloc(&sink, 4 * G));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](bool work, int edges, int K) {
    hipMemset(board, 0, sizeof(Granule) * 2 * G);
    hipEventRecord(e0);
    if (work) hipLaunchKernelGGL(chain_kernel<true>, dim3(G), dim3(256), 0, 0, board, G, edges, K, cyc, sink);
    else hipLaunchKernelGGL(chain_kernel<false>, dim3(G), dim3(256), 0, 0, board, G, edges, K, cyc, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
  };
  run(false, 10, 256);
  const float a200 = run(false, 200, 256), a1000 = run(false, 1000, 256);
  printf("%d workgroups, all-to-all edge (8-byte granule per workgroup, one wavefront sweeps %d granules): %.2f us per edge\n", G, G, (a1000 - a200) / 800.0f);
  for (int K : {256, 768, 1280}) {
    const float b200 = run(true, 200, K), b1000 = run(true, 1000, K);
    printf("  with an LDS-resident GEMV slab between edges (K = %d, 4 columns per workgroup): %.2f us per edge\n", K, (b1000 - b200) / 800.0f);
  }
  // mailbox
  unsigned *h_req, *h_ack; float *h_in, *h_out;
  CK(hipHostMalloc(&h_req, 64, hipHostMallocMapped)); CK(hipHostMalloc(&h_ack, 64, hipHostMallocMapped));
  CK(hipHostMalloc(&h_in, 640, hipHostMallocMapped)); CK(hipHostMalloc(&h_out, 960, hipHostMallocMapped));
  *h_req = 0; *h_ack = 0;
  for (int i = 0; i < 160; ++i) h_in[i] = (float)i;
  const int rounds = 2000;
  hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(256), 0, 0, h_req, h_ack, h_in, h_out, rounds);
  std::vector<double> rtt;
  for (int r = 1; r <= rounds; ++r) {
    h_in[0] = (float)r;
    const auto t0 = std::chrono::steady_clock::now();
    __atomic_store_n(h_req, (unsigned)r, __ATOMIC_RELEASE);
    while (__atomic_load_n(h_ack, __ATOMIC_ACQUIRE) != (unsigned)r) {}
    const auto t1 = std::chrono::steady_clock::now();
    if (h_out[0] != (float)r + 1.0f) { printf("mailbox payload mismatch at round %d\n", r); return 1; }
    rtt.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
  }
  CK(hipDeviceSynchronize());
  std::sort(rtt.begin() + 100, rtt.end());
  const size_t n = rtt.size() - 100;
  printf("mailbox round trip (host writes 640 B + sequence word in mapped memory, parked kernel answers 960 B + word): p50 %.1f us, p99 %.1f us\n",
         rtt[100 + n / 2], rtt[100 + (n * 99) / 100]);
  return 0;
}
