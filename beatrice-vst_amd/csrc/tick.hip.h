// tick.hip.h -- "tick" pipelining of the batch: every LAYER of the per-hop chain is its own pipeline stage.
// (Included by batch.hip after struct BeatriceBatch; not a stand-alone header.)
//
// Why: at a few hundred streams a step is a string of ~40 dependent, latency-bound launches, each using a fraction of
// the chip for 4-9 us.  Streams never exchange data and every mutable thing is per-stream
// state, so consecutive steps can overlap as long as stage s of step t+1 runs after stage s of step t.  A tick is ONE
// launch in which stage s works on step (tick - s): ~1900 independent workgroups of 26 different steps fill the chip, and
// the only synchronisation is the kernel boundary between ticks.  Outputs are bit-identical to the in-order chain: the same
// kernel bodies run on the same data, only later.  A step's output appears n_stages - 1 ticks after its input was
// fed; BeatriceBatch_Synchronize drains the pipeline.  For callers that enqueue steps ahead of their completion
// (resident audio); a server that needs each hop back before the next runs the in-order chain.
//
// What makes it safe:
//   * every ring a LATER stage reads holds one more step slot than the in-order chain needs (State::create with
//     pipe_slack; the block scratch xa, read again four stages on, holds five), the small non-ring outputs of the pitch
//     head are double-buffered by step parity;
//   * each stage has its own {step counter, I/O slot} pair, computed by the host (which knows the step every stage is
//     at) and passed to the launch BY VALUE (fuse::StepPairs in the kernel arguments; a workgroup leaves its body's pair
//     in LDS, ring.h stepc): no launch to publish counters, no dependent global load at the start of a workgroup;
//     -1 = "no step this tick" (fill, drain);
//   * per-stream settings are versioned: a change goes once into a slot of a snapshot ring on the device (staged in a
//     ring of pinned host copies and fetched by the prologue kernel itself), and the same prologue kernel copies the
//     byte range a consumer stage reads from its step's snapshot into that stage's private arrays at the tick the step
//     arrives there (consumers: k-NN, pitch head, conditioning mix, the attention half of each block).
#pragma once
#include <type_traits>

#include "chain_layers.hip.h"
#include "fuse.hip.h"
#include "rowchain.hip.h"
#include "tail_stages.hip.h"

namespace tick {

constexpr int kRangeHalvesFrom = 99;   // a partly filled tick with at least this many occupied stages runs in its own halves order (99: none does)
constexpr int kMaxStages = 32, kRing = 64, kMaxCopies = 16;  // (copies: 7 consumers + the snapshot upload)

struct Copy { unsigned char* dst; const unsigned char* src; int bytes; };
constexpr int kCopyChunk = 4096;  // bytes one workgroup moves (256 lanes x 16 bytes, one round trip)
struct Prolog {
  int n_stages;
  int hop[kMaxStages], io[kMaxStages];
  int n_copies;
  Copy copy[kMaxCopies];
  int first_chunk[kMaxCopies + 1];  // workgroups [first_chunk[c], first_chunk[c + 1]) perform copy c
};
// Settings traffic of a tick, one launch (only on ticks that have any): entry 0 may be the upload of a new snapshot --
// its source is PINNED HOST memory, read by the kernel itself over PCIe (a copy command in the stream costs a switch to the
// copy engine and back, ~40 us per tick when settings change every step) -- the others move a consumer stage's byte range
// from an OLDER snapshot slot into that stage's private copy.  One workgroup per 4 KB.
static __global__ __launch_bounds__(256) void prologue_kernel(const Prolog p) {
  int c = 0;
  while (c + 1 < p.n_copies && (int)blockIdx.x >= p.first_chunk[c + 1]) ++c;
  const Copy cp = p.copy[c];
  const int at = ((int)blockIdx.x - p.first_chunk[c]) * kCopyChunk + (int)threadIdx.x * 16;
  if (at + 16 <= cp.bytes) *reinterpret_cast<uint4*>(cp.dst + at) = *reinterpret_cast<const uint4*>(cp.src + at);  // (ranges are multiples of 16 bytes)
}

using namespace bhip;
using namespace wave_layers;
// Tilings of the tick launch.  Everything runs in 512-thread workgroups, two per CU (<= 128 VGPRs, <= 78 KB of LDS): the
// stream-stationary bodies (conditioned blocks as two row-local chains each, rowchain.hip.h; the fused upsampler tail)
// and the remaining layers as rc::conv_rows_body: 32 rows (two MFMA row tiles sharing every weight fragment) x the layer's
// width (the three 256-column layers with long reductions: 128-column slabs) per workgroup, the A operand streamed through
// LDS one 256-long reduction segment at a time, weights prefetched four to eight k-blocks ahead.  A tick wants MFMA density
// and occupancy, not the shortest dependent chain: per-layer launches with 16x32 tiles and 128-wide k-chunks (the in-order
// chain's choice) measured 0.224 ms per tick at 256 streams, 16x64 tiles in 256-thread workgroups 0.104, this layout 0.078.
// (A/B build switches; measured at 256 / 1024 streams: one row tile, full width 3.18 / 3.83 M frames/s; the three long
//  layers as 32 rows x 128 columns 3.27 / 3.85; every conv_rows body with two row tiles 3.29 / 3.98)
#ifndef TICK_RB_COLS
#define TICK_RB_COLS 128
#endif
#ifndef TICK_GRU_RT
#define TICK_GRU_RT 2
#endif
#ifndef TICK_MID_RT
#define TICK_MID_RT 2
#endif
// (phone.f3 with ONE row tile: it runs in the launch's second round, where short workgroups matter more than traffic --
//  32 workgroups of 28 us end the launch later than 64 of 18 us: 3.45 -> 3.55 M frames/s at 256 streams)
#ifndef TICK_F3_RT
#define TICK_F3_RT 1
#endif
#ifndef TICK_RB_RT
#define TICK_RB_RT 2
#endif
#ifndef TICK_P1_RT
#define TICK_P1_RT TICK_MID_RT
#endif

enum BodyType {
  T_F1, T_FFT, T_F2, T_F3, T_F4, T_F5, T_P1, T_RB, T_P23, T_POUT, T_HEAD, T_OUT, T_COND, T_INP, T_UP1, T_RES1A, T_RES1B, T_UP2,
  T_QGRU, T_PGRU, T_VQ, T_TAIL, T_TAIL1, T_TAIL2, T_TAIL3, T_BLKA1, T_BLKA2, T_BLKA4, T_BLKA8, T_BLKB, T_BLKBQ,
  T_F4S, T_F5S, T_RBS, T_P1S, T_UP1S, T_TAIL1S, T_TAIL2S, T_QGRU1, T_PGRU1, T_QGRUM, T_PGRUM, T_COUNT
};
// The bodies of a tick with H hops per stage (H = 1: a step is one 10 ms hop of every stream; H = 2: two -- every stage then
// works on twice the rows per weight fragment and per launch, and a launch's fixed costs are paid once per two hops; the two
// recurrent layers run their two hops one after the other inside a workgroup).  Same layer types as the in-order chain at H hops
// per step: rows are (stream, frame of the step), the recurrences (GRU, pitch head, the tail's histories) run in frame order.
template <int H>
struct Ops {
  using PL = PhoneLayers<H>;
  using QL1 = PitchLayers<H>;
  using OpF2 = rc::ConvRowsOp<typename PL::F2, 0, TICK_MID_RT>;
  using OpF3 = rc::ConvRowsOp<typename PL::F3, 0, TICK_F3_RT>;
  using OpF4 = rc::ConvRowsOp<typename PL::F4, TICK_RB_COLS, TICK_RB_RT>;
  using OpF5 = rc::ConvRowsOp<typename PL::F5, TICK_RB_COLS, TICK_RB_RT>;
  using OpRB = rc::ConvRowsOp<typename PL::RBL, TICK_RB_COLS, TICK_RB_RT>;
  using OpOUT = rc::ConvRowsOp<typename PL::OUTL>;
  using OpP1 = rc::ConvRowsOp<typename QL1::P1, 0, TICK_P1_RT>;
  using OpP23 = rc::ConvRowsOp<typename QL1::P23>;
  using OpPOUT = rc::ConvRowsOp<typename QL1::POUT>;
  using OpINP = rc::ConvRowsOp<INP<H>>;
  using OpUP1 = rc::ConvRowsOp<UP<256, 128, 5, H>, 128, TICK_MID_RT>;   // 640 columns: five slabs
  using OpRES1A = rc::ConvRowsOp<RES<128, 1, 5 * H>, 0, TICK_MID_RT>;
  using OpRES1B = rc::ConvRowsOp<RES<128, 3, 5 * H>, 0, TICK_MID_RT>;
  using OpUP2 = rc::ConvRowsOp<UP<128, 64, 4, 5 * H>, 0, TICK_MID_RT>;
  // The SPARSE table (the first ticks of a fill, while only front-end stages have a step): the bodies whose workgroups last longest
  // in half-size pieces -- one row tile per convolution workgroup (and half the streams per tail workgroup, unused: see tick_run).  A
  // partly filled launch lasts as long as its longest workgroup and has idle slots to spare, so shorter workgroups in larger
  // numbers are what it wants (in a full tick they cost ~3 %: twice the weight traffic per row, more prologues).  Same arithmetic,
  // same rings: a tick may use either table.
  using OpF4s = rc::ConvRowsOp<typename PL::F4, TICK_RB_COLS, 1>;
  using OpF5s = rc::ConvRowsOp<typename PL::F5, TICK_RB_COLS, 1>;
  using OpRBs = rc::ConvRowsOp<typename PL::RBL, TICK_RB_COLS, 1>;
  using OpP1s = rc::ConvRowsOp<typename QL1::P1, 0, 1>;
  using OpUP1s = rc::ConvRowsOp<UP<256, 128, 5, H>, 128, 1>;
  // the tail's stages: streams per workgroup so that a workgroup holds the same number of rows at every H (80 / 240 / 480; H = 2: T2 160).
  // H = 4: a stream's 320 / 960 frames of a step do not fit the LDS -- T2 and T3 run the step as two sub-steps of two hops, one after
  // the other inside the workgroup, the layers' histories carried in LDS (tail_stages.hip.h prologue / carry_histories)
  static constexpr int NSUB = H > 2 ? H / 2 : 1;
  using T1 = tst::T1OpS<(H == 1 ? tst::kT1Streams : (H == 2 ? 2 : 1)), H>;
  using T2 = tst::T2OpS<(H == 1 ? tst::kT2Streams : 1), H, NSUB>;
  using T3 = tst::T3OpS<(H == 1 ? tst::kT3Streams : 1), H, NSUB>;
  using T1s = tst::T1OpS<(H == 1 ? 2 : 1), H>;
  using T2s = tst::T2OpS<(H == 1 ? 2 : 1), H, NSUB>;
  // The GRUs: the column-split cell (fused_small.hip.h; 32 streams x 16 hidden units per workgroup, a weight fragment held in
  // registers).  Hop t of a step needs the whole state vector of hop t - 1: with several hops per step the cell of hop t PUBLISHES
  // its state as tagged granules and the cell of hop t + 1 -- another set of workgroups of the SAME launch and stage, later in
  // dispatch order (tick_build_table) -- polls them (GruArgs::link_*): the one place where workgroups of a tick wait for each other.
  // GruQ / GruP: hop 0 (publishes when H > 1); GruQm / GruPm: hops 1 .. H - 2 (poll and publish; H = 4: two of each, every link
  // its own granule array); GruQ1 / GruP1: the last hop (polls)
  static_assert(H <= 4, "link arrays: tick::State");
  using GruQ = GruOp<128, 128, TICK_GRU_RT, (H > 1 ? 1 : 0)>;
  using GruP = GruOp<256, 256, TICK_GRU_RT, (H > 1 ? 1 : 0)>;
  using GruQ1 = std::conditional_t<H == 1, rc::NopOp, GruOp<128, 128, TICK_GRU_RT, 2>>;
  using GruP1 = std::conditional_t<H == 1, rc::NopOp, GruOp<256, 256, TICK_GRU_RT, 2>>;
  using GruQm = std::conditional_t<H <= 2, rc::NopOp, GruOp<128, 128, TICK_GRU_RT, 3>>;
  using GruPm = std::conditional_t<H <= 2, rc::NopOp, GruOp<256, 256, TICK_GRU_RT, 3>>;
  using Vq = std::conditional_t<H == 1, VqOp, VqRowsOp<H>>;   // (several hops: a stream's rows in one workgroup, the codebook read once)
  template <class... Ms> struct List { using Tab = fuse::Table<Ms...>; using Builder = fuse::TableBuilder<Ms...>; };
  using L = List<
      fuse::Many<F1Op2, 1>, fuse::Many<FftOp2, 1>, fuse::Many<OpF2, 1>, fuse::Many<OpF3, 1>, fuse::Many<OpF4, 1>, fuse::Many<OpF5, 1>,
      fuse::Many<OpP1, 1>, fuse::Many<OpRB, 4>, fuse::Many<OpP23, 2>, fuse::Many<OpPOUT, 1>, fuse::Many<HeadOp8, 1>,
      fuse::Many<OpOUT, 1>, fuse::Many<CondOp2, 1>, fuse::Many<OpINP, 1>, fuse::Many<OpUP1, 1>, fuse::Many<OpRES1A, 1>,
      fuse::Many<OpRES1B, 1>, fuse::Many<OpUP2, 1>, fuse::Many<GruQ, 1>, fuse::Many<GruP, 1>,
      fuse::Many<Vq, 1>, fuse::Many<TailOp<H>, 1>, fuse::Many<T1, 1>, fuse::Many<T2, 1>, fuse::Many<T3, 1>, fuse::Many<rc::BlockAOp<1, H>, 1>, fuse::Many<rc::BlockAOp<2, H>, 1>,
      fuse::Many<rc::BlockAOp<4, H>, 1>, fuse::Many<rc::BlockAOp<8, H>, 1>, fuse::Many<rc::BlockBOpH<H>, 4>, fuse::Many<rc::BlockBqOpH<H>, 4>,
      fuse::Many<OpF4s, 1>, fuse::Many<OpF5s, 1>, fuse::Many<OpRBs, 4>, fuse::Many<OpP1s, 1>, fuse::Many<OpUP1s, 1>, fuse::Many<T1s, 1>, fuse::Many<T2s, 1>,
      fuse::Many<GruQ1, 1>, fuse::Many<GruP1, 1>, fuse::Many<GruQm, 2>, fuse::Many<GruPm, 2>>;
  using Tab = typename L::Tab;
  using Builder = typename L::Builder;
};

// Stage of each body.  Front end: the in-order chain's launch order, the pitch estimator zipped into the content
// encoder's stages from the fourth launch on (its spectrum ring then has its reader one stage later, like every other
// ring).  A conditioned block is two stages (rowchain.hip.h).  The same plan at every number of hops per step.
struct Plan {
  static constexpr int F1 = 0, F2 = 1, F3 = 2, F4 = 3, FFT = 3, F5 = 4, P1 = 4, RB0 = 5, P2 = 5, QGRU = 7, POUT = 8, PGRU = 9, HEAD = 9,
                       OUT = 10, COND = 10, VQ = 11, INP = 12, BLK0 = 13;
  static constexpr int per_block = 2;  // stages per conditioned block
  int blk(int b) const { return BLK0 + per_block * b; }
  int up1() const { return blk(B_NBLOCKS); }
  int tail() const { return up1() + 4; }   // first of the tail's stages (one with the fused body, three with tail_stages.hip.h)
  bool split_tail = true;
  int count() const { return tail() + (split_tail ? 3 : 1); }
};
constexpr int kMaxHops = 4;   // hops per stage per tick the launch is built for (Ops<1>, Ops<2>, Ops<4>)
static_assert(Plan::BLK0 + Plan::per_block * B_NBLOCKS + 7 <= kMaxStages && kMaxStages + 2 <= kRing && kMaxStages <= fuse::kMaxStepPairs, "stage bookkeeping");

struct Consumer {  // a kernel that reads per-stream settings: its private copy of a byte range of the settings block
  int stage;
  size_t off, bytes;    // range in the settings snapshot
  unsigned char* dst;   // its private copy on the device
  int held;             // snapshot the private copy currently equals (-1: none)
};

struct State {
  bool on = false;
  void* d_table = nullptr;           // Ops<H>::Tab on the device
  int table_total = 0;
  void* d_table_sparse = nullptr;    // the same stages with the long bodies in half-size pieces (fill and drain ticks)
  int table_sparse_total = 0;
  double table_flops = 0, table_bytes = 0;  // algorithmic work of one full tick (sum over the bodies)
  fuse::WgDesc* d_desc = nullptr;    // dispatch order of the full / the sparse table by workgroup (tick_build_table_h)
  fuse::WgDesc* d_desc_sparse = nullptr;
  size_t desc_cap = 0, desc_sparse_cap = 0;
  // the full table's workgroups in plain span order (every body on all eight XCDs): the order of PARTLY FILLED ticks -- with few stages
  // occupied a body confined to half the chip leaves the other half idle (a 20-step run: 3.78 -> 3.51 M frames/s with the halves everywhere)
  fuse::WgDesc* d_desc_plain = nullptr;
  size_t desc_plain_cap = 0;
  int table_total_plain = 0;
  // ... and the same order with the workgroups of EMPTY stages left out, for the two shapes a fill and a drain go through: stages
  // 0 .. k occupied (range_off[0][k]) and stages k .. last occupied (range_off[1][k]).  A partly filled launch otherwise dispatches
  // every stage's workgroups only for most of them to leave at once -- thousands of slot turns per launch in a run of few steps.
  // One buffer; n = 0: not built (very large batches) or the tick's occupied stages are no such range -> the plain list.
  fuse::WgDesc* d_desc_ranges = nullptr;
  size_t desc_ranges_cap = 0;
  size_t range_off[2][kMaxStages] = {};
  int range_n[2][kMaxStages] = {};
  unsigned long long* d_trace = nullptr;  // BEATRICE_HIP_TICK_TRACE=<file>: per-workgroup timeline of the last full tick
  bool table_dirty = true;
  unsigned char* d_snap = nullptr;  // [kRing][snap_bytes] settings snapshots
  static constexpr int kStaging = 16;
  unsigned char* h_stage = nullptr; // [kStaging][snap_bytes] pinned: a snapshot on its way to the device
  hipEvent_t stage_ev[kStaging] = {};
  bool stage_pending[kStaging] = {};
  size_t snap_bytes = 0;
  long long tick = 0, n_fed = 0, last_feed_tick = -1000;
  long long fed_step[kRing];        // step fed at tick (index tick % kRing), -1 none
  int snap_of_step[kRing], hop_of_step[kRing], io_of_step[kRing];
  int snap_cur = -1, snap_next = 0;
  std::vector<Consumer> consumers;
  Plan plan;
  // Ragged steps (the shell's silent-block rule per stream, BeatriceBatch_SetSilentStreams in tick mode): a stream that sits a
  // step out does not advance -- so every stream has its OWN step counter from then on, and a step carries the counters of its
  // streams (-1: absent) through the stages, like its settings: [kRing][row] on the device, staged through pinned copies.
  bool ragged = false;
  int row = 0;                        // ints per step (B rounded up to a multiple of 4: the prologue copies 16-byte pieces)
  std::vector<int> hop_s;             // [B] the streams' counters on the host
  int* d_hopv = nullptr;              // [kRing][row]
  int* h_hopv = nullptr;              // [kStaging][row] pinned
  hipEvent_t hv_ev[kStaging] = {};
  bool hv_pending[kStaging] = {};
  bool step_ragged[kRing] = {};       // step u (at [u % kRing]) carries per-stream counters
  // back to one counter at a drained point (batch_tick.hip.h tick_relevel): the table of the batch's rings, the streams' deficits
  void* d_ring_table = nullptr;       // FreezeRing [n_ring_table]
  int n_ring_table = 0;
  int* d_shift = nullptr;             // [B]
  // several hops per step: the granules that link the GRU cells of a step's hops inside a launch (fused_small.hip.h GruArgs::link_*)
  unsigned long long* d_link_q = nullptr;   // [H - 1][B][128]: link t = hop t's state for hop t + 1
  unsigned long long* d_link_p = nullptr;   // [H - 1][B][256]
  int* h_link_dead = nullptr;               // pinned: a cell gave a wait up
};

}  // namespace tick
