// tick.hip.h -- "tick" pipelining of the batch: every LAYER of the per-hop chain is its own pipeline stage.
// (Included by batch.hip after struct BeatriceBatch; not a stand-alone header.)
//
// Why: at a few hundred streams a step is a string of ~40 dependent, latency-bound launches, each using a fraction of
// the chip for 4-9 us.  Streams never exchange data and every mutable thing is per-stream
// state, so consecutive steps can overlap as long as stage s of step t+1 runs after stage s of step t.  A tick is ONE
// launch (plus the fused upsampler tail, whose LDS footprint would cap the occupancy of everything else) in which
// stage s works on step (tick - s): ~5000 independent workgroups of 40 different steps fill the chip, and the only
// synchronisation is the kernel boundary between ticks.  Outputs are bit-identical to the in-order chain: the same
// kernel bodies run on the same data, only later.  A step's output appears n_stages - 1 ticks after its input was
// fed; BeatriceBatch_Synchronize drains the pipeline.  For callers that enqueue steps ahead of their completion
// (resident audio); a server that needs each hop back before the next runs the in-order chain.
//
// What makes it safe:
//   * every ring a LATER stage reads holds one more step slot than the in-order chain needs (State::create with
//     pipe_slack; the block scratch xa, read again four stages on, holds five), the small non-ring outputs of the pitch
//     head are double-buffered by step parity;
//   * each stage reads its own {step counter, I/O slot} pair (d_hops[stage]), written by a small prologue launch from
//     values the host computes (the host knows which step every stage is at); -1 = "no step this tick" (fill, drain);
//   * per-stream settings are versioned: a change is uploaded once into a slot of a snapshot ring, and the prologue
//     copies the snapshot into a consumer's private arrays at the tick that consumer reaches the step the change
//     belongs to (consumers: k-NN, pitch head, conditioning mix, the two attention kernels of each block).
#pragma once
#include "chain_layers.hip.h"
#include "fuse.hip.h"

namespace tick {

constexpr int kMaxStages = 48, kRing = 64, kMaxCopies = 16;

struct Copy { unsigned char* dst; const unsigned char* src; int bytes; };
struct Prolog {
  int n_stages;
  int hop[kMaxStages], io[kMaxStages];
  int n_copies;
  Copy copy[kMaxCopies];
};
// workgroup 0 writes the stages' counters, workgroup 1 + c performs settings copy c (16-byte granules)
static __global__ __launch_bounds__(256) void prologue_kernel(int* __restrict__ hops, const Prolog p) {
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) {
    for (int s = tid; s < p.n_stages; s += 256) { hops[2 * s] = p.hop[s]; hops[2 * s + 1] = p.io[s]; }
    return;
  }
  const Copy c = p.copy[blockIdx.x - 1];
  const uint4* src = reinterpret_cast<const uint4*>(c.src);
  uint4* dst = reinterpret_cast<uint4*>(c.dst);
  for (int i = tid; i < c.bytes / 16; i += 256) dst[i] = src[i];
}

using namespace bhip;
using namespace wave_layers;
// Tiling of the tick launches: 32 rows x 64 columns per workgroup, four wavefronts, each walking ALL reduction segments
// of its column tile (one accumulator per segment, added in order): 256 threads and ~10-20 KB of LDS whatever the layer,
// so several workgroups fit a CU at once -- a tick wants occupancy, not the shortest dependent chain.  Measured at 256
// streams (ms per tick): 16x64 tiles 0.106, 32x64 0.104, 16x32 0.131, 128-wide k-chunks (152 VGPRs) 0.119.
#ifndef TICK_KFEW
#define TICK_KFEW false
#endif
#ifndef TICK_WM
#define TICK_WM 2
#endif
#ifndef TICK_LN
#define TICK_LN 4
#endif
template <class L> using TT = TileCfg<TICK_WM, 1, 1, TICK_LN, 1, TICK_KFEW>;
template <class L> using CT = ConvOp<L, TT<L>>;
using PL = PhoneLayers<1>;
using QL1 = PitchLayers<1>;

// main launch (256-thread workgroups)
enum BodyType {
  T_F1, T_FFT, T_F2, T_F3, T_F4, T_F5, T_P1, T_RB, T_P23, T_POUT, T_HEAD, T_OUT, T_COND, T_INP,
  T_C1D1, T_C1D2, T_C1D4, T_C1D8, T_C2, T_Q, T_SCORE, T_PV, T_UP1, T_RES1A, T_RES1B, T_UP2, T_COUNT
};
#define TICK_MAIN_TYPES                                                                                                        \
    fuse::Many<F1Op, 1>, fuse::Many<FftOp, 1>, fuse::Many<CT<PL::F2>, 1>, fuse::Many<CT<PL::F3>, 1>, fuse::Many<CT<PL::F4>, 1>, \
    fuse::Many<CT<PL::F5>, 1>, fuse::Many<CT<QL1::P1>, 1>, fuse::Many<CT<PL::RBL>, 4>, fuse::Many<CT<QL1::P23>, 2>,            \
    fuse::Many<CT<QL1::POUT>, 1>, fuse::Many<HeadOp, 1>, fuse::Many<CT<PL::OUTL>, 1>, fuse::Many<CondOp, 1>,                    \
    fuse::Many<CT<INP<1>>, 1>, fuse::Many<CT<C1<1, 1>>, 1>, fuse::Many<CT<C1<2, 1>>, 1>, fuse::Many<CT<C1<4, 1>>, 1>,           \
    fuse::Many<CT<C1<8, 1>>, 1>, fuse::Many<CT<C2<1>>, 8>, fuse::Many<CT<QL<1>>, 4>, fuse::Many<CT<SCORE<1>>, 4>,               \
    fuse::Many<AttnPvOp, 4>, fuse::Many<CT<UP<256, 128, 5, 1>>, 1>, fuse::Many<CT<RES<128, 1, 5>>, 1>,                          \
    fuse::Many<CT<RES<128, 3, 5>>, 1>, fuse::Many<CT<UP<128, 64, 4, 5>>, 1>
using Tab = fuse::Table<TICK_MAIN_TYPES>;
using Builder = fuse::TableBuilder<TICK_MAIN_TYPES>;
// second launch: the bodies with larger workgroups (the two GRU cells: six wavefronts; k-NN: one thread per codebook row)
enum AuxType { A_QGRU, A_PGRU, A_VQ };
#define TICK_AUX_TYPES fuse::Many<GruOp<128, 128>, 1>, fuse::Many<GruOp<256, 256>, 1>, fuse::Many<VqOp, 1>
using AuxTab = fuse::Table<TICK_AUX_TYPES>;
using AuxBuilder = fuse::TableBuilder<TICK_AUX_TYPES>;

// stage of each layer: the in-order chain's launch order, the pitch estimator zipped into the content encoder's
// stages from the fourth launch on (its spectrum ring then has its reader one stage later, like every other ring)
enum Stage {
  S_F1 = 0, S_F2 = 1, S_F3 = 2, S_F4 = 3, S_FFT = 3, S_F5 = 4, S_P1 = 4, S_RB0 = 5, S_P2 = 5, S_RB1 = 6, S_P3 = 6, S_RB2 = 7, S_QGRU = 7,
  S_RB3 = 8, S_POUT = 8, S_PGRU = 9, S_HEAD = 9, S_OUT = 10, S_COND = 10, S_VQ = 11, S_INP = 12, S_BLK0 = 13 /* c1 c2 q qk pv o */,
  S_UP1 = 37, S_RES1A = 38, S_RES1B = 39, S_UP2 = 40, S_TAIL = 41, S_COUNT = 42
};
static_assert(S_COUNT <= kMaxStages && S_COUNT + 2 <= kRing, "stage bookkeeping");

struct Consumer {  // a kernel that reads per-stream settings: its private copy of a byte range of the settings block
  int stage;
  size_t off, bytes;    // range in the settings snapshot
  unsigned char* dst;   // its private copy on the device
  int held;             // snapshot the private copy currently equals (-1: none)
};

struct State {
  bool on = false;
  int* d_hops = nullptr;            // [kMaxStages][2]
  Tab* d_table = nullptr;
  AuxTab* d_aux = nullptr;
  int table_total = 0, aux_total = 0;
  bool table_dirty = true;
  unsigned char* d_snap = nullptr;  // [kRing][snap_bytes] settings snapshots
  size_t snap_bytes = 0;
  long long tick = 0, n_fed = 0, last_feed_tick = -1000;
  long long fed_step[kRing];        // step fed at tick (index tick % kRing), -1 none
  int snap_of_step[kRing], hop_of_step[kRing], io_of_step[kRing];
  int snap_cur = -1, snap_next = 0;
  std::vector<Consumer> consumers;
  TailArgs tail;
};

}  // namespace tick
