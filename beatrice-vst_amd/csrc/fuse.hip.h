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
#include "meas_env.h"
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "engine.h"
#include "ring.h"

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

// ---------------------------------------------------------------------------------------------------------------
// Table-driven variant: MANY bodies (a whole pipeline's worth, batch.hip tick mode) in one launch.  The arguments
// live in device memory (a launch with ~50 argument blocks exceeds what a kernel-argument segment should carry, and
// they are the same every launch); a workgroup finds its body with one vector load of the bodies' first-workgroup
// indices + a ballot, then branches on the body's TYPE, so that a kernel body used by several layers (the eight
// 256->256 linears of the conditioned blocks, say) is compiled into the launch once.
template <class Op, int CAP>
struct Many { using op = Op; static constexpr int cap = CAP; };
struct Span { int first, gx, type, arg; };  // workgroups [first, next first) run body `type` with its argument block `arg`
constexpr int kMaxSpans = 64;
// (bits 16-23 of Span::arg: the body's pipeline stage, see StepPairs)
// Step counter and I/O slot of every pipeline stage for ONE launch, passed to the table kernel by value (kernel arguments):
// a workgroup looks its body's stage up and leaves the pair in LDS for the body (stepc, ring.h); a stage whose counter is
// negative has no step in this launch and its workgroups leave at once.
constexpr int kMaxStepPairs = 32;
// hv / hopv / n_streams: ragged steps (ring.h stepc::hopv) -- stage s reads its per-stream counters at hopv + hv[s] * n_streams
// (hv[s] < 0 or hopv null: every stream is at hop[s])
struct StepPairs { int hop[kMaxStepPairs], io[kMaxStepPairs], hv[kMaxStepPairs]; const int* hopv; int n_streams; };
template <class... Ms> struct Banks;
template <> struct Banks<> {};
template <class M, class... Rest>
struct Banks<M, Rest...> {
  typename M::op::Args a[M::cap];
  Banks<Rest...> rest;
};
// One workgroup of a launch whose dispatch ORDER is free (Table::desc): body type, argument block | stage << 16 (as Span::arg),
// the workgroup's index inside its body, the body's grid x-extent.  With a descriptor per workgroup the order is free:
// TableBuilder::two_halves() puts every body with weights on one half of the chip.
struct WgDesc { int type, arg, local, gx; };
template <class... Ms>
struct Table {
  int n_spans, total, unused0, unused1;
  unsigned long long* trace;         // measurement aid (may be null): per workgroup {start, end} wall clock + body type
  Span span[kMaxSpans];  // 16 bytes each: a wavefront fetches all of them with one load, lane i = body i
  Banks<Ms...> banks;
};
template <class... Ms> struct MaxM;
template <class M> struct MaxM<M> { static constexpr int NTHR = M::op::NTHR, LDS = M::op::LDS_FLOATS; };
template <class M, class... Rest>
struct MaxM<M, Rest...> {
  static constexpr int NTHR = M::op::NTHR > MaxM<Rest...>::NTHR ? M::op::NTHR : MaxM<Rest...>::NTHR;
  static constexpr int LDS = M::op::LDS_FLOATS > MaxM<Rest...>::LDS ? M::op::LDS_FLOATS : MaxM<Rest...>::LDS;
};

template <int I, int NTHR, bool RAG, class M, class... Rest>
__device__ __forceinline__ void run_type(const Banks<M, Rest...>& b, const Span& sp, int id, float* lds, const int* hop_imm) {
  if (sp.type == I) {
    using Op = typename M::op;
    if (Op::NTHR >= NTHR || (int)threadIdx.x < Op::NTHR) {
      typename Op::Args a = b.a[sp.arg];   // (uniform: scalar loads)
      if constexpr (!RAG) stepc::set_hop(a, hop_imm, 0);   // the stage's {step counter, I/O slot} as an immediate (ring.h stepc)
#ifdef FUSE_GLOBALIZE   // A/B build switch, OFF: the table's pointers told to be global memory (ring.h as_global) turn the launch's 2 271
      globalize(a);         // flat loads into global loads with pipelined waits -- and the tick got 4 % SLOWER (profiles/r04_notes.md section 1)
#endif
      Op::template run_t<RAG>(a, id % sp.gx, id / sp.gx, lds);
    }
  } else {
    if constexpr (sizeof...(Rest) > 0) run_type<I + 1, NTHR, RAG, Rest...>(b.rest, sp, id, lds, hop_imm);
  }
}

template <int MINW, bool RAG, class... Ms>
__device__ __forceinline__ void table_body(const Table<Ms...>* __restrict__ t, const WgDesc* __restrict__ desc, const StepPairs& pairs);
// the same with a register budget: MINW = minimum wavefronts per SIMD the launch wants resident (HIP's second
// __launch_bounds__ parameter): 4 with 512-thread workgroups = two workgroups per CU = at most 128 VGPRs
// RAG: the launch of RAGGED steps (streams that sit steps out; ring.h stepc::hopv) -- a second instance of every body that works with
// per-row step counters; the common launch (RAG = false) contains none of that
// desc: the launch's workgroups in dispatch order, a KERNEL ARGUMENT of its own (`const ... __restrict__`): the compiler then reads a
// workgroup's descriptor with one scalar load that depends on nothing but the kernel arguments.  (Round 4 found the body by a vector
// load of the span table + ballot + readlane behind a dependent scalar load of the table's header; a descriptor pointer stored INSIDE
// the table is a plain pointer to the compiler -- the descriptor then comes through a flat load into VGPRs and the whole type dispatch
// turns into vector compares and exec masks: 226 spills, the tick 7-12 % slower, profiles/r05_notes.md.)
template <int MINW, bool RAG, class... Ms>
__global__ __launch_bounds__(MaxM<Ms...>::NTHR, MINW) void table_kernel_w(const Table<Ms...>* __restrict__ t, const WgDesc* __restrict__ desc, const StepPairs pairs) { table_body<MINW, RAG, Ms...>(t, desc, pairs); }
template <int MINW, bool RAG, class... Ms>
__device__ __forceinline__ void table_body(const Table<Ms...>* __restrict__ t, const WgDesc* __restrict__ desc, const StepPairs& pairs) {
  __shared__ __attribute__((aligned(16))) float lds[MaxM<Ms...>::LDS > 0 ? MaxM<Ms...>::LDS : 1];
  const WgDesc d = desc[blockIdx.x];
  Span sp;
  sp.first = 0; sp.gx = d.gx; sp.type = d.type; sp.arg = d.arg;
  if (sp.type < 0) return;  // filler index
  const int stage = (sp.arg >> 16) & 0xff;
  const int step = pairs.hop[stage];
  if (step < 0) return;    // fill / drain: this stage has no step in this launch
  const int io = pairs.io[stage];
  if constexpr (RAG) {   // (ragged steps: the bodies take the pair, and their streams' own counters, from LDS)
    if (threadIdx.x == 0) {
      stepc::pair[0] = step; stepc::pair[1] = io;
      stepc::hopv = pairs.hopv != nullptr && pairs.hv[stage] >= 0 ? pairs.hopv + (size_t)pairs.hv[stage] * pairs.n_streams : nullptr;
    }
    __syncthreads();
  }
  unsigned long long* const trace = t->trace;
  const unsigned long long t0 = trace ? wall_clock64() : 0;
  const unsigned long long c0 = trace ? __builtin_readcyclecounter() : 0;   // shader clock (the wall clock is 100 MHz): their ratio = the clock the launch ran at
  const int body_wg = d.local;
  sp.arg &= 0xff;
  run_type<0, MaxM<Ms...>::NTHR, RAG, Ms...>(t->banks, sp, body_wg, lds, stepc::immediate(step, io));
  if (trace && threadIdx.x == 0) {
    trace[3 * (size_t)blockIdx.x] = t0; trace[3 * (size_t)blockIdx.x + 1] = wall_clock64(); trace[3 * (size_t)blockIdx.x + 2] = (unsigned long long)sp.type | ((__builtin_readcyclecounter() - c0) << 8);
  }
}

// WITH_RAGGED = false: the launch has no ragged instance (its caller never passes ragged = true)
template <int MINW, bool WITH_RAGGED = true, class... Ms>
static inline void launch_table_w(const Table<Ms...>* d_table, const WgDesc* d_desc, int total, hipStream_t stream, const StepPairs& pairs, const bool ragged = false) {
  constexpr int pad = 0;   // (dynamic LDS on top of the static block: none.  Round 2 measured one workgroup per CU this way: slower)
  if constexpr (WITH_RAGGED) {
    if (ragged) { hipLaunchKernelGGL((table_kernel_w<MINW, true, Ms...>), dim3(total), dim3(MaxM<Ms...>::NTHR), pad, stream, d_table, d_desc, pairs); return; }
  }
  hipLaunchKernelGGL((table_kernel_w<MINW, false, Ms...>), dim3(total), dim3(MaxM<Ms...>::NTHR), pad, stream, d_table, d_desc, pairs);
}

// host side: fill a Table<Ms...>.  add<I>() appends one body of type I (its index in Ms...); bodies run in the order added
template <int I, class B> struct BankAt;
template <class M, class... Rest> struct BankAt<0, Banks<M, Rest...>> {
  static typename M::op::Args* get(Banks<M, Rest...>& b) { return b.a; }
  static constexpr int cap = M::cap;
};
template <int I, class M, class... Rest> struct BankAt<I, Banks<M, Rest...>> {
  static auto get(Banks<M, Rest...>& b) { return BankAt<I - 1, Banks<Rest...>>::get(b.rest); }
  static constexpr int cap = BankAt<I - 1, Banks<Rest...>>::cap;
};
template <class... Ms>
struct TableBuilder {
  Table<Ms...> t{};
  int used[sizeof...(Ms)] = {};
  double cost[kMaxSpans] = {};  // estimated time of the body's workgroups (us, from the launch's timelines): what two_halves balances
  int n_wg[kMaxSpans] = {};
  TableBuilder() { for (Span& sp : t.span) sp = Span{0x7fffffff, 1, -1, 0}; }
  bool ok = true;
  double flops = 0, bytes = 0;
  unsigned long long only = ~0ull;   // measurement builds (batch_tick.hip.h): bit I clear = bodies of type I are left out of the table
  // the workgroups in the order the bodies were added (every body a contiguous run)
  std::vector<WgDesc> in_span_order() const {
    std::vector<WgDesc> out;
    for (int i = 0; i < t.n_spans; ++i)
      for (int w = 0; w < n_wg[i]; ++w) out.push_back(WgDesc{t.span[i].type, t.span[i].arg, w, t.span[i].gx});
    return out;
  }
  // Two HALVES of the chip.  Workgroup i of a launch runs on XCD i % 8 (observed; used for speed only), so the indices with
  // i % 8 < 4 and those with i % 8 >= 4 are two queues, four XCDs each.  A body flagged `pinned` goes whole into ONE queue (the
  // one with less estimated time so far, cost[]; spans that name the same `group` >= 0 share a queue: the linked GRU cells of a
  // step's hops), so its weights pass through four of the eight L2s instead of all of them; the other bodies are dealt to both
  // queues, more to the one that is behind.  Order inside a queue = the order added.  The launch's memory-side traffic falls by
  // half the pinned bodies' weights x 8 (profiles/r05_notes.md); what it may cost is balance: the two halves no longer share one
  // queue of work.
  // stage_lo .. stage_hi: only the bodies of these stages (a partly filled tick's list, balanced over the halves for what it holds)
  std::vector<WgDesc> two_halves(const bool* pinned, const int* group, const int NQ = 2 /* queues: 2 halves or 4 quarters of the chip */,
                                 const int stage_lo = 0, const int stage_hi = 255) const {
    struct Item { int span, local; };
    std::vector<Item> q[4];
    double load[4] = {0, 0, 0, 0};
    int group_q[kMaxSpans];
    for (int& g : group_q) g = -1;
    auto lightest = [&]() { int h = 0; for (int x = 1; x < NQ; ++x) if (load[x] < load[h]) h = x; return h; };
    for (int i = 0; i < t.n_spans; ++i) {
      const int stage = (t.span[i].arg >> 16) & 0xff;
      if (stage < stage_lo || stage > stage_hi) continue;
      const double c = n_wg[i] > 0 ? cost[i] / n_wg[i] : 0.0;
      if (pinned[i]) {
        int h = lightest();
        if (group[i] >= 0 && group[i] < kMaxSpans) { if (group_q[group[i]] >= 0) h = group_q[group[i]]; else group_q[group[i]] = h; }
        for (int w = 0; w < n_wg[i]; ++w) q[h].push_back(Item{i, w});
        load[h] += cost[i];
      } else {
        for (int w = 0; w < n_wg[i]; ++w) { const int h = lightest(); q[h].push_back(Item{i, w}); load[h] += c; }
      }
    }
    std::vector<WgDesc> out;
    auto emit = [&](const Item& x) { const Span& sp = t.span[x.span]; out.push_back(WgDesc{sp.type, sp.arg, x.local, sp.gx}); };
    size_t at[4] = {0, 0, 0, 0};
    auto all_have = [&]() { for (int h = 0; h < NQ; ++h) if (at[h] >= q[h].size()) return false; return true; };
    while (all_have()) {   // 8 / NQ consecutive indices (= XCDs) for each queue in turn
      for (int h = 0; h < NQ; ++h)
        for (int j = 0; j < 8 / NQ; ++j) {
          if (at[h] < q[h].size()) emit(q[h][at[h]++]);
          else out.push_back(WgDesc{-1, 0, 0, 1});   // (a filler index: that slot's turn passes)
        }
    }
    for (bool more = true; more;) {   // (the longer queues' rests go to all eight XCDs, round-robin)
      more = false;
      for (int h = 0; h < NQ; ++h) if (at[h] < q[h].size()) { emit(q[h][at[h]++]); more = true; }
    }
    return out;
  }
  template <int I, class Args>
  void add(const bhip::LaunchInfo& info, const Args& a, dim3 grid, int stage, bool on = true, double wg_cost = 1.0, bool spread_over_xcds = false) {
    if (!on || ((only >> I) & 1) == 0) return;
    using BA = BankAt<I, Banks<Ms...>>;
    if (t.n_spans >= kMaxSpans || used[I] >= BA::cap) { ok = false; return; }
    BA::get(t.banks)[used[I]] = a;
    Span& sp = t.span[t.n_spans];
    sp.first = t.total; sp.gx = (int)grid.x > 0 ? (int)grid.x : 1; sp.type = I; sp.arg = used[I]++ | (stage << 16);
    n_wg[t.n_spans] = (int)(grid.x * grid.y);
    cost[t.n_spans] = wg_cost * n_wg[t.n_spans];
    ++t.n_spans;
    t.total += (int)(grid.x * grid.y);
    flops += info.flops; bytes += info.bytes;
  }
};

}  // namespace fuse
