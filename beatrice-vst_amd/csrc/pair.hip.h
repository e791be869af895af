// pair.hip.h -- two INDEPENDENT kernels of the per-hop chain in one launch.
//
// At a few hundred streams every launch of the chain is latency-bound (a dependent launch costs
// 4-9 us whatever it computes, profiles/r01_notes.md) and a hipGraph replays its nodes one after the
// other even when they are independent.  The content encoder and the pitch estimator only share the
// hop's audio, so the pitch estimator's seven launches ride along with seven launches of the content
// encoder: one grid, the first nA workgroups run body A, the rest run body B.  Each body is the
// unchanged kernel body (same arithmetic, same order: results are bit-identical to separate launches).
// The workgroup size is the larger of the two; the surplus wavefronts of the smaller body exit at
// once (s_barrier only waits for the wavefronts of a workgroup that are still alive).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "engine.h"

template <class OA, class OB>
__global__ __launch_bounds__((OA::NTHR > OB::NTHR ? OA::NTHR : OB::NTHR)) void pair_kernel(const typename OA::Args a,
                                                                                            const typename OB::Args b,
                                                                                            const int gxA, const int nA,
                                                                                            const int gxB) {
  int id = blockIdx.x;
  if (id < nA) {
    if (OA::NTHR >= OB::NTHR || (int)threadIdx.x < OA::NTHR) OA::run(a, id % gxA, id / gxA);
  } else {
    id -= nA;
    if (OB::NTHR >= OA::NTHR || (int)threadIdx.x < OB::NTHR) OB::run(b, id % gxB, id / gxB);
  }
}

template <class OA, class OB>
static inline void launch_pair(const bhip::LaunchInfo& ia, const typename OA::Args& a, dim3 ga, const bhip::LaunchInfo& ib,
                               const typename OB::Args& b, dim3 gb, hipStream_t stream) {
  constexpr int NTHR = OA::NTHR > OB::NTHR ? OA::NTHR : OB::NTHR;
  const int nA = (int)(ga.x * ga.y), nB = (int)(gb.x * gb.y);
  // the profiler keeps the name pointer: one string per (A, B) pair, alive for the process
  static const std::string name = std::string(ia.name) + "+" + ib.name;
  const bhip::LaunchInfo info{name.c_str(), ia.flops + ib.flops, ia.bytes + ib.bytes};
  bhip::launch_site(info, stream, [&] {
    hipLaunchKernelGGL((pair_kernel<OA, OB>), dim3(nA + nB), dim3(NTHR), 0, stream, a, b, (int)ga.x, nA, (int)gb.x);
  });
}
