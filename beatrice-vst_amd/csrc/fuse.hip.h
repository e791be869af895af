// fuse.hip.h -- several INDEPENDENT kernel bodies of the per-hop chain in one launch.
//
// At a few hundred streams every launch of the chain is latency-bound (a dependent launch costs 4-9 us whatever
// it computes, profiles/r01_notes.md) and a hipGraph replays its nodes one after the other even when they are
// independent.  Bodies that do not depend on each other therefore share a launch: one 1-D grid, the first n0
// workgroups run body 0, the next n1 body 1, ...  Each body is the unchanged kernel body (same arithmetic, same
// order: results are bit-identical to separate launches).  The workgroup size is the largest of the bodies'; the
// surplus wavefronts of a smaller body exit at once (s_barrier only waits for the wavefronts of a workgroup that
// are still alive); the LDS block is the largest of the bodies' (they get it as an argument), not the sum.
// An entry with n = 0 workgroups switches its body off for this launch.
//
// Who shares launches: the pitch estimator's launches ride in the content encoder's (front.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "engine.h"

namespace fuse {

template <class Op>
struct Entry {  // one body of a fused launch: its arguments, the x-extent of its (x, y) grid, its workgroup count
  typename Op::Args a;
  int gx, n;
};
template <class... Ops> struct Pack;
template <> struct Pack<> {};
template <class Op, class... Rest>
struct Pack<Op, Rest...> {
  Entry<Op> e;
  Pack<Rest...> rest;
};

template <class... Ops> struct Max;
template <class Op> struct Max<Op> { static constexpr int NTHR = Op::NTHR, LDS = Op::LDS_FLOATS; };
template <class Op, class... Rest>
struct Max<Op, Rest...> {
  static constexpr int NTHR = Op::NTHR > Max<Rest...>::NTHR ? Op::NTHR : Max<Rest...>::NTHR;
  static constexpr int LDS = Op::LDS_FLOATS > Max<Rest...>::LDS ? Op::LDS_FLOATS : Max<Rest...>::LDS;
};

template <int NTHR, class Op, class... Rest>
__device__ __forceinline__ void dispatch(const Pack<Op, Rest...>& p, int id, float* lds) {
  if (id < p.e.n) {
    if (Op::NTHR >= NTHR || (int)threadIdx.x < Op::NTHR) Op::run(p.e.a, id % p.e.gx, id / p.e.gx, lds);
  } else {
    if constexpr (sizeof...(Rest) > 0) dispatch<NTHR, Rest...>(p.rest, id - p.e.n, lds);
  }
}

template <class... Ops>
__global__ __launch_bounds__(Max<Ops...>::NTHR) void fused_kernel(const Pack<Ops...> p) {
  __shared__ __attribute__((aligned(16))) float lds[Max<Ops...>::LDS > 0 ? Max<Ops...>::LDS : 1];
  dispatch<Max<Ops...>::NTHR, Ops...>(p, blockIdx.x, lds);
}

// host side: what one body contributes to a fused launch
template <class Op>
struct Part {
  bhip::LaunchInfo info;
  typename Op::Args args;
  dim3 grid;
  bool on = true;
};
template <class Op>
static inline Part<Op> part(const bhip::LaunchInfo& info, const typename Op::Args& args, dim3 grid, bool on = true) {
  return Part<Op>{info, args, grid, on};
}

static inline void fill(Pack<>&, int&, double&, double&, std::string*) {}
template <class Op, class... Rest, class... PRest>
static inline void fill(Pack<Op, Rest...>& pk, int& total, double& flops, double& bytes, std::string* name, const Part<Op>& p,
                        const PRest&... rest) {
  pk.e.a = p.args;
  pk.e.gx = (int)p.grid.x > 0 ? (int)p.grid.x : 1;
  pk.e.n = p.on ? (int)(p.grid.x * p.grid.y) : 0;
  total += pk.e.n;
  if (p.on) { flops += p.info.flops; bytes += p.info.bytes; }
  if (name) { if (!name->empty()) *name += "+"; *name += p.info.name; }
  fill(pk.rest, total, flops, bytes, name, rest...);
}

template <class... Ops>
static inline void launch(hipStream_t stream, const Part<Ops>&... parts) {
  Pack<Ops...> pk;
  int total = 0;
  double flops = 0, bytes = 0;
  // the profiler keeps the name pointer: one string per combination of bodies, alive for the process
  static std::string name;
  const bool first = name.empty();
  fill(pk, total, flops, bytes, first ? &name : nullptr, parts...);
  if (total == 0) return;
  const bhip::LaunchInfo info{name.c_str(), flops, bytes};
  bhip::launch_site(info, stream, [&] {
    hipLaunchKernelGGL((fused_kernel<Ops...>), dim3(total), dim3(Max<Ops...>::NTHR), 0, stream, pk);
  });
}

}  // namespace fuse
