// How much does a device-wide (or XCD-team-wide) barrier inside a persistent kernel cost on MI355X,
// compared with a kernel boundary in a hipGraph (1.65 us, launch_floor.hip)?  One workgroup per CU
// (cooperative launch: all co-resident), N phases; in each phase a workgroup writes a 2 KB slice, the
// barrier follows, then it reads 16 KB written by the other workgroups of its team.
//   mode 0: all 256 workgroups form one team           (agent-scope release/acquire atomics)
//   mode 1: 8 teams of 32 workgroups, team = wg % 8    (what an XCD-local chain would use)
//   mode 2: 16 teams of 16 workgroups, team = (wg % 8) * 2 + (wg / 8) % 2
// Build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef VARIANT
#define VARIANT 1
#endif

struct Args { int* counters; float* buf; int phases; int mode; int* check; };

__device__ __forceinline__ void team_barrier(int* cnt, int team_size, int phase) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int target = (phase + 1) * team_size;
#if VARIANT == 0   // release add, acquire loads in the spin loop
    __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
#else              // one release fence, relaxed add and spin, one acquire fence
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void persistent(const Args a) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  int team, member, team_size;
  if (a.mode == 0) { team = 0; member = wg; team_size = gridDim.x; }
  else if (a.mode == 1) { team = wg % 8; member = wg / 8; team_size = gridDim.x / 8; }
  else { team = (wg % 8) * 2 + (wg / 8) % 2; member = wg / 16; team_size = gridDim.x / 16; }
  int* cnt = a.counters + team * 64;                // one cache line per team
  float* tbuf = a.buf + (size_t)team * 2 * 65536;   // double-buffered team scratch, 256 KB each side
  float acc = 0.f;
  for (int p = 0; p < a.phases; ++p) {
    float* w = tbuf + (p & 1) * 65536 + member * 512;
    w[tid] = (float)(p + member);
    w[tid + 256] = (float)(p - member);
    team_barrier(cnt, team_size, p);
    const float* r = tbuf + (p & 1) * 65536;
    // read 16 KB: the slices of 8 team members (wrapping)
    for (int i = 0; i < 16; ++i) {
      const int m = (member + 1 + (i >> 1)) % team_size;
      acc += __builtin_nontemporal_load(r + m * 512 + (i & 1) * 256 + tid);
    }
  }
  if (tid == 0) a.check[wg] = (int)acc;
}

int main() {
  (void)0; hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* counters; float* buf; int* check;
  hipMalloc(&counters, 64 * 64 * sizeof(int)); hipMalloc(&buf, 16 * 2 * 65536 * sizeof(float)); hipMalloc(&check, 256 * sizeof(int));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int phases : {1, 101}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemsetAsync(counters, 0, 64 * 64 * sizeof(int), s);
        Args a{counters, buf, phases, mode, check};
        void* params[] = {&a};
        hipEventRecord(e0, s);
        hipError_t err = hipLaunchCooperativeKernel((const void*)persistent, dim3(256), dim3(256), params, 0, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        if (err != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(err)); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("mode %d phases %3d: %.2f us total\n", mode, phases, best * 1000.f);
    }
  std::vector<int> h(256); hipMemcpy(h.data(), check, 256 * sizeof(int), hipMemcpyDeviceToHost);
  printf("check[0]=%d check[255]=%d  (per-phase cost = (t101 - t1) / 100)\n", h[0], h[255]);
  return 0;
}
