// ring.h -- per-stream activation rings ("data layout in HBM").
//
// Every tensor that a causal layer must remember across hops lives in a ring:
//   base[stream][frame][channel], frames per stream R = n * m,
//   n = frames produced per hop, m = 1 + ceil(history / n) hop slots.
// Hop h writes its n new frames contiguously at slot (h % m); a consumer reads the new frames
// plus up to `history` older frames (wrapping backwards).  Producers write straight into the
// consumer's ring, so state update costs no extra pass and the only per-hop state traffic is
// "write each activation once, read it once per tap".
#pragma once
#include <hip/hip_runtime.h>

struct Ring {
  float* base;
  int C;  // channels per frame
  int n;  // frames per hop
  int m;  // hop slots
};

// Pointers that reach a kernel through a table in device memory (the tick launch, fuse.hip.h) are GENERIC to the compiler:
// it cannot see that they point to global memory and emits flat_load / flat_store -- which count on BOTH vmcnt and lgkmcnt,
// so every wait for an LDS operand (s_waitcnt lgkmcnt(0): flat and LDS results return out of order) also waits for every
// weight prefetch in flight.  A round trip through address space 1 tells the compiler what it is (InferAddressSpaces then
// emits global_load / global_store with their own counter and the ISA shows the intended `s_waitcnt vmcnt(8) lgkmcnt(7)`).
// MEASURED (round 4, same box, A/B): the tick launch is 4 % SLOWER with it (69.2 vs 66.5 us at 256 streams, 241 vs 232 us at
// 1 024) -- VGPR spills 48 -> 112 in the table kernel, and the launch is bound by ALU issue, not by exposed latency.  The
// helpers stay for kernels that take pointers from memory and ARE latency-bound; fuse::run_type uses them only under
// -DFUSE_GLOBALIZE.
// (A plain generic -> global -> generic cast is folded away by the front end; the empty asm between the two casts keeps it.
//  "s": the pointer is wave-uniform and stays in scalar registers -- as_global_v for a pointer that may differ between lanes.)
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);   // (readfirstlane: a no-op for a value already known uniform)
  const unsigned long long u = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
                               ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
  __attribute__((address_space(1))) T* g = (__attribute__((address_space(1))) T*)reinterpret_cast<T*>(u);
  asm("" : "+s"(g));
  return (T*)g;
}
template <class T>
__device__ __forceinline__ T* as_global_v(T* p) {
  __attribute__((address_space(1))) T* g = (__attribute__((address_space(1))) T*)p;
  asm("" : "+v"(g));
  return (T*)g;
}
// globalize(args): every pointer of an argument block through as_global (one overload per block type, next to its struct)
__device__ __forceinline__ void globalize(Ring& r) { r.base = as_global(r.base); }

__host__ __device__ inline int ring_frames(const Ring& r) { return r.n * r.m; }
__host__ __device__ inline size_t ring_stream_floats(const Ring& r) { return (size_t)r.n * r.m * r.C; }

// first new frame of hop `hop`
__device__ __forceinline__ int ring_pos(const Ring& r, int hop) { return (hop % r.m) * r.n; }

// pointer to frame (pos + rel) of stream b; rel in [-history, n)
__device__ __forceinline__ float* ring_frame(const Ring& r, int b, int pos, int rel) {
  int f = pos + rel;
  const int R = r.n * r.m;
  if (f < 0) f += R;
  // 32-bit index arithmetic (a ring holds fewer than 2^32 floats: RingArena::build refuses larger ones): the 64-bit form cost
  // six to eight VALU instructions per call, and epilogues call this once per output element
  return r.base + (unsigned)(b * R + f) * (unsigned)r.C;
}

// The step counter a kernel works on, and the slot of resident I/O buffers that belongs to it.  Ordinary launches read
// the pair from device memory (args.hop); in the batch's tick launch (tick.hip.h) the table kernel takes every stage's
// pair from its kernel arguments and leaves it in LDS for the body it dispatches to -- no dependent global load at the
// start of a workgroup, no separate launch to publish the counters: there args.hop is null.
// Third form: the VALUE travels in the pointer bits (stepc::immediate(hop[, slot]): bits 0-23 = counter + 2, bits 24-35 = I/O slot;
// the pointers in question are the library's own device allocations, never below kImmediateTop = 64 GB), again without a dependent load -- the 1-stream calls whose host tracks
// the counter, and since round 5 the COMMON tick launch: the table kernel puts the stage's pair into the body's argument block, so a
// workgroup starts without the LDS write + barrier that published the pair (only the launch of RAGGED steps still leaves `hopv`,
// and the pair, in LDS).
namespace stepc {
__shared__ int pair[2];
// The tick launch with RAGGED steps (streams that sit a step out: the shell's silent-block rule per stream, batch.hip): this
// stage's per-stream step counters [B] -- stream b is at hopv[b], or -1 when it takes no part in the step -- left here by the table
// kernel beside `pair`; nullptr while every stream is at the step's common counter.  Only meaningful where args.hop is null.
__shared__ const int* hopv;
constexpr unsigned long long kImmediateTop = 1ull << 36;
constexpr int kImmediateMaxSlot = 4095;
constexpr unsigned kImmediateHopMask = 0xffffffu;   // (the counter wraps at B_HOP_WRAP = lcm(1..17) = 12 252 240 < 2^24 - 2)
__host__ __device__ inline const int* immediate(int hop, int slot = 0) {
  return reinterpret_cast<const int*>(static_cast<unsigned long long>((unsigned)(hop + 2)) | (static_cast<unsigned long long>((unsigned)slot) << 24));
}
__device__ __forceinline__ int step(const int* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  return v == 0 ? pair[0] : (v < kImmediateTop ? (int)((unsigned)v & kImmediateHopMask) - 2 : *p);
}
// the step counter of stream b in this step: the common one, or the stream's own (-1: absent) in a ragged tick step
__device__ __forceinline__ int of(const int* p, const int hop, const int b) {
  if (reinterpret_cast<unsigned long long>(p) != 0) return hop;
  const int* v = hopv;
  return v != nullptr ? v[b] : hop;
}
__device__ __forceinline__ bool ragged(const int* p) { return reinterpret_cast<unsigned long long>(p) == 0 && hopv != nullptr; }
// The same as compile-time variants: the tick launch exists twice (fuse::table_kernel_w<.., RAG>), and only the RAG = true one --
// launched once a stream has sat a step out -- pays for per-row counters; with RAG = false these fold to `hop` / `false` and
// the bodies are exactly what they were (the common path measured 3.5 % slower with run-time tests).
template <bool RAG> __device__ __forceinline__ int of_t(const int hop, const int b) {
  if constexpr (RAG) { const int* v = hopv; return v != nullptr ? v[b] : hop; }
  else return hop;
}
template <bool RAG> __device__ __forceinline__ bool rag_t() {
  if constexpr (RAG) return hopv != nullptr;
  else return false;
}
__device__ __forceinline__ int slot(const int* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  return v == 0 ? pair[1] : (v < kImmediateTop ? (int)(v >> 24) : p[1]);
}
// the `hop` member of a body's argument block (some blocks nest the kernel's own: F1Args2 { F1Args a; ... })
template <class A> __device__ __forceinline__ auto set_hop(A& a, const int* p, int) -> decltype((void)(a.hop = p)) { a.hop = p; }
template <class A> __device__ __forceinline__ auto set_hop(A& a, const int* p, long) -> decltype((void)(a.a.hop = p)) { a.a.hop = p; }
template <class A> __device__ __forceinline__ void set_hop(A&, const int*, ...) {}
}  // namespace stepc
