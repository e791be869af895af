// team_edge.hip -- the price of one all-to-all edge of a 1-stream layer chain inside ONE launch: a team of T workgroups, every
// phase each workgroup needs the whole 256-float activation vector the team produced in the phase before.
//   exchange A: tagged granules -- {value, phase tag} as ONE 8-byte agent-scope (sc1) store, consumers poll the granules
//               themselves with agent-scope loads (no flag, no fence)
//   exchange B: plain stores + agent release fence + relaxed counter add / spin + agent acquire fence (round 1's barrier)
//   placement 0: the T workgroups as dispatched (workgroup b -> XCD b % 8); 1: all on XCD 0 (8 T launched, b % 8 == 0 work)
// with W floats of "weights" (a slab of a [256 x 256] matrix) streamed per workgroup and phase (0 = none).
// build: hipcc -O3 --offload-arch=gfx950 -o team_edge team_edge.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct Args { unsigned long long* gran; float* plain; int* counter; const float* weights; float* out; int phases, team, same_xcd, exchange, wfloats; };

__global__ __launch_bounds__(256) void chain(const Args a) {
  __shared__ float x[256];
  int wg = blockIdx.x;
  if (a.same_xcd) { if (wg % 8 != 0) return; wg /= 8; }
  const int tid = threadIdx.x, T = a.team, per = 256 / T;
  float acc = 0.f;
  for (int p = 0; p < a.phases; ++p) {
    // ---- gather the vector of phase p (phase 0: nothing to wait for)
    if (p > 0) {
      if (a.exchange == 0) {
        unsigned long long* g = a.gran + (size_t)(p & 1) * 256 + tid;
        unsigned long long v;
        do { v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((int)(v >> 32) != p);
        x[tid] = __uint_as_float((unsigned)v);
      } else {
        if (tid == 0) {
          while (__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p * T) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        x[tid] = a.plain[(size_t)(p & 1) * 256 + tid];
      }
    } else x[tid] = 1.0f;
    __syncthreads();
    // ---- this workgroup's slice of the next vector: per columns, each a 256-long dot product (weights optional)
    float mine = 0.f;
    if (a.wfloats > 0) {
      const float* w = a.weights + ((size_t)(p % 8) * T + wg) * a.wfloats;
      for (int i = tid; i < a.wfloats; i += 256) mine += w[i] * x[i & 255];
    } else mine = x[tid] * 0.5f;
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
    acc += mine;
    __syncthreads();
    // ---- publish: thread t < per writes element wg * per + t
    if (tid < per) {
      const float val = x[wg * per + tid] * 0.999f + 0.001f;
      if (a.exchange == 0) {
        const unsigned long long v = ((unsigned long long)(unsigned)(p + 1) << 32) | __float_as_uint(val);
        __hip_atomic_store(a.gran + (size_t)((p + 1) & 1) * 256 + wg * per + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else a.plain[(size_t)((p + 1) & 1) * 256 + wg * per + tid] = val;
    }
    if (a.exchange == 1) {
      __syncthreads();
      if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
  }
  if (tid == 0) a.out[wg] = acc;
}

int main() {
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  Args a{};
  hipMalloc(&a.gran, 2 * 256 * 8); hipMalloc(&a.plain, 2 * 256 * 4); hipMalloc(&a.counter, 256); hipMalloc(&a.out, 4096);
  const size_t wmax = (size_t)8 * 64 * 65536;
  float* w; hipMalloc(&w, wmax * 4); hipMemset(w, 0, wmax * 4); a.weights = w;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int exchange : {0, 1})
    for (int same : {0, 1})
      for (int T : {8, 16, 32, 64})
        for (int wf : {0, 4096, 16384}) {
          float t[2];
          int k = 0;
          for (int phases : {1, 201}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
              hipMemsetAsync(a.gran, 0, 2 * 256 * 8, s); hipMemsetAsync(a.counter, 0, 4, s);
              a.phases = phases; a.team = T; a.same_xcd = same; a.exchange = exchange; a.wfloats = wf;
              hipEventRecord(e0, s);
              hipLaunchKernelGGL(chain, dim3(same ? 8 * T : T), dim3(256), 0, s, a);
              hipEventRecord(e1, s); hipEventSynchronize(e1);
              float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            t[k++] = best;
          }
          printf("exchange %s  %s  team %2d  weights %5d floats/wg/phase: %.2f us per phase (launch with one phase %.1f us)\n", exchange ? "fence+counter" : "tagged granules",
                 same ? "one XCD   " : "dispatched", T, wf, (t[1] - t[0]) * 1000.f / 200.f, t[0] * 1000.f);
        }
  return 0;
}
