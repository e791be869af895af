// tail_stages.hip.h -- the upsampler tail of the waveform generator as THREE pipeline stages of the tick launch
// (MODEL_SPEC 4.4.3, from the first residual conv of stage 2 to the output samples):
//
//   T1: res2a, res2b (64 ch, 20 frames / stream-hop) -> up3        writes the 32-channel frames to a ring in HBM
//   T2: res3a, res3b (32 ch, 80 frames)              -> up4        writes the 16-channel frames to a ring in HBM
//   T3: res4a, res4b (16 ch, 240 frames) -> lrelu, Conv1d(16 -> 1, k7), tanh -> 240 samples
//
// Why not the one-workgroup-per-stream kernel of wave_tail.hip.h (which stays the in-order chain's tail): inside the tick
// launch the tail held a third of all workgroup time for a fifth of the FLOPs (profiles/r03_notes.md).  Measured with
// tools/microbench/tail_timing: its MFMA loops already saturate the CU's matrix pipes while they run -- the waste is
// (a) PADDING and IMBALANCE: 20 frames fill 1.25 row tiles of 16, 80 frames five tiles over four row groups, 48 output
// columns three tiles over eight wavefronts: the busiest wavefront issues 352 MFMAs where 235 would do; and (b) 40 % of
// a wavefront's time in epilogues, history copies and barriers, nine layers deep, with 78 KB of LDS (three rotating
// buffers of raw + activated copies) allowing no second stream to fill the gaps.
// Here a workgroup runs ONE THIRD of the chain for SEVERAL streams: rows = (stream, frame) fill the row tiles
// (4 x 20 = 80 = 5 tiles, 3 x 80 = 15 tiles, 2 x 240 = 30 tiles), every SIMD gets the same MFMA count, a barrier or an
// epilogue is paid once per 2-4 streams.  More stages cost a throughput pipeline nothing (a step's latency grows by two ticks).
//
// The cost model that shapes the bodies (tools/microbench/mfma_mix, profiles/r03_notes.md): a SIMD issues ONE instruction
// stream -- a v_mfma_f32_16x16x4_f32 holds it for 32 cycles, and every VALU instruction beside it costs ~5 cycles, every LDS
// read ~12, whichever wavefront issues them; nothing overlaps.  A body's time is 32 MFMA + 5 VALU + 12 LDS, so VALU work
// per MFMA is what to minimise:
//   * the LDS holds ACTIVATED values only (lrelu applied once per element where it is produced, not once per use as an A
//     operand: a 64-channel frame is used 12 times); the RAW value a residual connection adds back comes from global memory
//     (the stage's input ring; a scratch the first residual layer writes for the second), requested before the tile's MFMAs;
//   * A operands are read at compile-time offsets from one pointer per row tile (lowest tap first: no address arithmetic);
//   * an epilogue computes (stream, frame) once per four rows (T is a multiple of 4: the four rows a lane owns in a 16 x 16
//     result tile never straddle streams) and the bias once per layer.
//
// Numerics: operation for operation those of wave_tail.hip.h (every K <= 256: one k-ascending MFMA chain per output,
// bias, then residual), same packed weights, and the SAME per-stream state block (TS_* offsets) for the histories a layer
// keeps across hops -- each stage touches only its own part of it -- so the tick pipeline and the in-order chain can be
// switched on the same streams at any drained point.  The 32- and 16-channel frames between the stages go through two-slot
// rings (step parity); their two-frame input histories stay in the state block (TS_YA3, TS_YA4) as before.
#pragma once
#include <hip/hip_runtime.h>

#include "engine.h"
#include "kernels_misc.hip.h"
#include "ring.h"
#include "spec_math.hip.h"
#include "wave_tail.hip.h"

// tools/microbench/tst_timing.hip: shader-clock stamps of wavefront 0 at the phase boundaries of a stage body
#ifdef TST_TIMING
__device__ unsigned long long* g_tst_stamps;   // [workgroups][16]
#define TST_STAMP(i) do { if (threadIdx.x == 0) g_tst_stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define TST_STAMP(i) do { } while (0)
#endif

namespace tst {

constexpr int NTHR = 512, NWAVE = 8;
constexpr int HP = 6;  // history rows in front of each stream's frames in an LDS buffer (the deepest tap: k3, dilation 3)

struct StageArgs {
  Ring in;                 // T1: output of up2 (C 64, 20 frames, history 2); T2 / T3: the two-slot ring of the stage before
  Ring out;                // T1 / T2: ring the transposed conv writes (32 ch x 80, 16 ch x 240 frames per step)
  float* state;            // [B][TAIL_STATE_FLOATS]
  const float *w[3], *b[3];  // resA, resB, up (T3: only two)
  const float *fin_w, *fin_b;  // T3
  float* d_out;            // T3: [B][240] (x resident slots)
  size_t io_stride;
  const int* hop;
  int B;
};
__device__ __forceinline__ void globalize(StageArgs& a) {
  globalize(a.in); globalize(a.out); a.state = as_global(a.state);
  for (int i = 0; i < 3; ++i) { a.w[i] = as_global(a.w[i]); a.b[i] = as_global(a.b[i]); }
  a.fin_w = as_global(a.fin_w); a.fin_b = as_global(a.fin_b); a.d_out = as_global(a.d_out); a.hop = as_global(a.hop);
}

// lrelu(x) = max(x, 0.1 x), bit for bit MODEL_SPEC's `x > 0 ? x : 0.1 x` (x and 0.1 x have the same sign).  The instruction
// itself: fmaxf() makes the compiler canonicalise both operands first (two more VALU instructions per value).
__device__ __forceinline__ float lrelu_max(float x) {
  float r;
  const float y = 0.1f * x;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// LDS buffer of S streams: [S][HP + T][C + 2] ACTIVATED values; row (s, t), t in [-HP, T).  Stride C + 2: the A-operand read
// of a wavefront (16 rows x 4 k) hits 32 distinct banks per half (bank = 2 row + k, as in wave_tail.hip.h)
template <int C> __host__ __device__ constexpr int cs() { return C + 2; }
template <int C, int T, int S> __host__ __device__ constexpr int buf_floats() { return S * (HP + T) * cs<C>(); }
template <int C, int T> __device__ __forceinline__ int row_off(int s, int t) { return (s * (HP + T) + HP + t) * cs<C>(); }
// LDS of a stage (floats): two activation buffers, the raw-input and third-layer-history stashes, the raw y of the 64-channel stage, the streams' step counters
template <int C, int T, int S, int HC_ROWS> constexpr int stage_lds() { return 2 * buf_floats<C, T, S>() + S * (2 + HC_ROWS) * C + (C > 32 ? S * T * C : 0) + 8; }   // (+ 8: the streams' step counters)

// Work split of a layer over the 8 wavefronts: NWN column groups x NWM row groups.  A wavefront keeps the B fragments of its
// CT column tiles in registers (1, or all of them when the tile count is not a power of two) and walks the row tiles
// wm, wm + NWM, ... two at a time.  Wavefronts w and w + 4 share a SIMD: with NWM = 2 the two row groups of a column sit
// on the same SIMD, so every SIMD gets the same number of MFMAs.
template <int NOUT>
struct Split {
  static constexpr int NTL = NOUT / 16;
  static constexpr bool POW2 = (NTL & (NTL - 1)) == 0;
  static constexpr int NWN = POW2 ? (NTL < 8 ? NTL : 8) : 1;
  static constexpr int NWM = 8 / NWN;
  static constexpr int CT = POW2 ? 1 : NTL;
};

// B fragments of a layer for this wavefront, k-blocks [KB0, KB1).  A layer's fragments are fetched in two halves: the first
// before the PREVIOUS layer's MFMAs (latency hidden behind them), the second after them -- with 64 channels a layer's
// fragments are 48 registers, and two whole layers' worth beside the accumulators and the A-operand pipeline spill.
template <int K, int NOUT, int KB0 = 0, int KB1 = K / 16>
__device__ __forceinline__ void fetch_b(const float* __restrict__ wpacked, float4 (&bf)[Split<NOUT>::CT][K / 16], int wave, int lane) {
  using SP = Split<NOUT>;
  const int wn = wave % SP::NWN;
#pragma unroll
  for (int ct = 0; ct < SP::CT; ++ct) {
    const int nt = SP::POW2 ? wn : ct;
    const float4* p = reinterpret_cast<const float4*>(wpacked) + (size_t)nt * (K / 16) * 64 + lane;
#pragma unroll
    for (int kb = KB0; kb < KB1; ++kb) bf[ct][kb] = p[(size_t)kb * 64];
  }
}

#ifndef TST_PIN
#define TST_PIN 1
#endif
// One pass of a layer: NT (1 or 2, compile time) row tiles of 16 that share every B operand.  ap[u] = the lane's A pointer
// of tile u: (row lane & 15 of the tile, LOWEST tap, k offset lane >> 4), so that every operand sits at a compile-time,
// non-negative offset.  The reduction runs in groups of four MFMA steps (16 k = one B fragment record); the A operands of
// group g + PD are read from LDS before the MFMAs of group g issue, and the order is pinned.
template <int CIN, int NOUT, int KSZ, int DIL, int NT>
__device__ __forceinline__ void pass(const float* const (&ap)[2], const float4 (&bf)[Split<NOUT>::CT][KSZ * CIN / 16],
                                     tail_f32x4 (&acc)[2][Split<NOUT>::CT]) {
  using SP = Split<NOUT>;
  constexpr int CS = cs<CIN>(), NS = KSZ * CIN / 4, NG = NS / 4;
  static_assert(NS % 4 == 0, "reduction length in blocks of 16");
  auto a_off = [](int ks) { const int kk = ks * 4, j = kk / CIN, c = kk % CIN; return c + j * DIL * CS; };
  constexpr int PD = CIN >= 64 ? 1 : 2;   // groups of distance between an operand's read and its use (64 channels: registers are short)
  float xq[PD][4][NT];
#pragma unroll
  for (int q = 0; q < PD; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int u = 0; u < NT; ++u) xq[q][e][u] = q < NG ? ap[u][a_off(q * 4 + e)] : 0.0f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    float ac[4][NT];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int u = 0; u < NT; ++u) ac[e][u] = xq[g % PD][e][u];
    if (g + PD < NG) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int u = 0; u < NT; ++u) xq[g % PD][e][u] = ap[u][a_off((g + PD) * 4 + e)];
    }
#if TST_PIN
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int ct = 0; ct < SP::CT; ++ct) {
        const float4 f = bf[ct][g];
        const float bv = e == 0 ? f.x : (e == 1 ? f.y : (e == 2 ? f.z : f.w));
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[u][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[e][u], bv, acc[u][ct], 0, 0, 0);
      }
  }
}

// One layer over the R = S * T rows of the workgroup: rows = (stream, frame), K = KSZ * CIN from the LDS buffer `in`
// (activated values; taps are row offsets inside a stream's block, history rows included), N = NOUT.
// A lane owns, of each 16 x 16 result tile, four consecutive rows of one column; T is a multiple of 4, so they are frames
// t0 .. t0 + 3 of ONE stream: pre(s, t0, n, ct, res) may request what the epilogue will need (residual values) before the
// tile's MFMAs, epi(s, t0, n, ct, acc4, res) receives the four finished chains.  Rows >= n_rows are padding (recomputed from
// the last row, never handed out).
// Passes a wavefront makes over a layer's row tiles (two tiles per pass), upper bound
template <int NOUT, int T, int S> __host__ __device__ constexpr int n_pass() { return ((S * T + 15) / 16 + 2 * Split<NOUT>::NWM - 1) / (2 * Split<NOUT>::NWM); }
// `res` = four values per (pass, tile of the pass, column tile) that live in REGISTERS across the layer -- and, when the
// caller hands the same array to the next layer of the same shape (same rows, same columns: the tile-to-wavefront map is
// the same), across layers: the first residual layer leaves its raw output there and the second one finds its residual
// without a round trip through memory.  FILL = 1: pre() fills them for every pass before the layer's first MFMA (a pass of a
// 16- or 32-channel layer is far shorter than a global round trip inside the tick launch); 2: per pass, before that pass's MFMAs
// (64 channels: registers are short there and a pass is 96 MFMAs long); 0: the caller's values are used as they are.
template <int CIN, int NOUT, int KSZ, int DIL, int T, int S, int FILL, class Pre, class Epi>
__device__ __forceinline__ void layer(const float* __restrict__ in, const float4 (&bf)[Split<NOUT>::CT][KSZ * CIN / 16], const int n_rows,
                                      const int wave, const int lane, float (&res)[n_pass<NOUT, T, S>()][2][Split<NOUT>::CT][4], Pre pre, Epi epi) {
  using SP = Split<NOUT>;
  static_assert(T % 4 == 0, "four rows of a lane in one stream");
  constexpr int R = S * T, NRT = (R + 15) / 16;
  const int i = lane & 15, kq = lane >> 4;
  // (the wavefront index as a SCALAR: the row-tile loop and its "second tile?" test must be scalar branches -- derived from
  //  threadIdx.x they are vector conditions, and every MFMA of the second tile ends up under its own exec-mask branch)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wn = wave_u % SP::NWN, wm = wave_u / SP::NWN;
  // what the epilogues will need from global memory (residual values): requested for EVERY pass of this wavefront before its
  // first MFMA -- a pass of a 16- or 32-channel layer is 12-48 MFMAs, far shorter than a global round trip inside the tick
  // launch.  (64 channels: per pass, registers are short there and a pass is 96 MFMAs long.)
  constexpr int NPASS = n_pass<NOUT, T, S>();
  constexpr bool AHEAD = FILL == 1;
  auto tile_rows = [&](int tile, int* s, int* t, bool* live) {
    const int r4 = tile * 16 + kq * 4;
    *live = r4 < n_rows; *s = r4 / T; *t = r4 % T;
  };
  if constexpr (AHEAD) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int tile = wm + (2 * p + u) * SP::NWM;
        int s_, t_; bool live_;
        tile_rows(tile, &s_, &t_, &live_);
#pragma unroll
        for (int ct = 0; ct < SP::CT; ++ct)
          if (tile < NRT && live_) pre(s_, t_, (SP::POW2 ? wn : ct) * 16 + i, ct, res[p][u][ct]);
      }
  }
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int t0 = wm + 2 * p * SP::NWM;
    if (t0 >= NRT) break;
    const bool two = t0 + SP::NWM < NRT;  // a second row tile shares every B operand of this pass
    const float* ap[2];
    int es[2], et[2];   // (stream, first frame) of the four result rows this lane owns in each tile
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int r = (t0 + u * SP::NWM) * 16 + i;
      r = r > R - 1 ? R - 1 : r;
      ap[u] = in + row_off<CIN, T>(r / T, r % T - (KSZ - 1) * DIL) + kq;
      tile_rows(t0 + u * SP::NWM, &es[u], &et[u], &live[u]);
      live[u] = live[u] && (u == 0 || two);
    }
    if constexpr (FILL == 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ct = 0; ct < SP::CT; ++ct)
          if (live[u]) pre(es[u], et[u], (SP::POW2 ? wn : ct) * 16 + i, ct, res[p][u][ct]);
    }
    tail_f32x4 acc[2][SP::CT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ct = 0; ct < SP::CT; ++ct) acc[u][ct] = tail_f32x4{0.f, 0.f, 0.f, 0.f};
    if (two) pass<CIN, NOUT, KSZ, DIL, 2>(ap, bf, acc);
    else pass<CIN, NOUT, KSZ, DIL, 1>(ap, bf, acc);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ct = 0; ct < SP::CT; ++ct)
        if (live[u]) epi(es[u], et[u], (SP::POW2 ? wn : ct) * 16 + i, ct, acc[u][ct], res[p][u][ct]);
  }
}

// ---- global memory <-> LDS, latency-aware.  A workgroup of these stages is a serial chain of phases; inside the tick launch
// a global round trip takes 1-2 us (the memory system is shared with ~500 other workgroups), so a body may afford very few
// of them on its critical path: what a stage reads from global memory up front -- its input frames, its pieces of the state
// block -- is requested in one burst before the first barrier (every thread issues all of its loads, then stores them to
// LDS), residual values are requested a tile ahead of their use, and the state block is written behind everything else.

// S x N floats (N a multiple of 4, S * N / 4 <= 512: one float4 per thread) of the state block at `ts_off`
// (shop: LDS [S], the step counter of each of the workgroup's streams, -1 = past the batch or sitting this step out)
// is stream s of the workgroup there in this step?  (RAG = false, the common launch: "inside the batch", no memory access)
template <bool RAG>
__device__ __forceinline__ bool live_s(const int* shop, const int b0, const int s, const int B) {
  if constexpr (RAG) return shop[s] >= 0;
  else return b0 + s < B;
}
template <int S, int N, bool RAG>
__device__ __forceinline__ float4 state_load(const float* __restrict__ state, const int ts_off, const int b0, const int* shop, const int B, const int tid) {
  static_assert(N % 4 == 0 && S * N / 4 <= NTHR, "one float4 per thread");
  const int s = tid / (N / 4), q = tid % (N / 4);
  return tid < S * N / 4 && live_s<RAG>(shop, b0, s, B) ? *reinterpret_cast<const float4*>(state + (size_t)(b0 + s) * TAIL_STATE_FLOATS + ts_off + 4 * q)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
}
// the streams of a workgroup: their step counters (the common one; their own in a ragged tick step; -1 where there is no stream)
template <int S, bool RAG>
__device__ __forceinline__ void stream_hops(const StageArgs& a, const int hop, const int b0, int* shop) {
  if constexpr (RAG) {
    if (threadIdx.x < S) {
      const int b = b0 + (int)threadIdx.x;
      shop[threadIdx.x] = b < a.B ? stepc::of_t<RAG>(hop, b) : -1;
    }
    __syncthreads();
  }
}
template <bool RAG>
__device__ __forceinline__ int stream_pos(const Ring& ring, const int pos, const int* shop, const int s) {
  if constexpr (RAG) return stepc::hopv != nullptr ? ring_pos(ring, shop[s]) : pos;
  else return pos;
}
__device__ __forceinline__ void store_act4(float* __restrict__ d, const float4 v) {   // (row stride C + 2: 8-byte aligned, not 16)
  reinterpret_cast<float2*>(d)[0] = make_float2(lrelu_max(v.x), lrelu_max(v.y));
  reinterpret_cast<float2*>(d)[1] = make_float2(lrelu_max(v.z), lrelu_max(v.w));
}

// The prologue shared by the three stages: the step's input frames -> X rows [-2, T), activated (history rows from the ring
// itself, or from the state block at TS_IN -- then the raw frames that will be the NEXT step's history are stashed in `hin`
// and reach the state block at the end); the six history rows of the second layer -> Y rows [-6, 0); the history of the third
// layer (HC_ROWS rows) -> `hc`, activated, to be copied into X once the first layer is done with it.
// A step cut into SUB-STEPS (NSUB > 1: the tick launch at four hops per step, where a stream's 320 / 960 frames of a step do not fit
// the LDS): the body runs the sub-steps of T frames one after the other, `fb` = the sub-step's first frame inside the step.  Only
// the FIRST sub-step takes the layers' histories from the state block -- the later ones find them in LDS, where carry_histories()
// left them (the input's two history frames are simply the ring's frames fb - 2, fb - 1) -- and only the LAST one writes them back.
template <int C, int T, int S, int TS_IN, int TS_B, int TS_C, int HC_ROWS, bool IN_FROM_RING_HISTORY, bool RAG>
__device__ __forceinline__ void prologue(const StageArgs& a, const int hop, const int b0, float* __restrict__ X, float* __restrict__ Y, float* __restrict__ hin,
                                         float* __restrict__ hc, const int tid, const int* shop, const int fb = 0, const bool first = true, const bool last = true) {
  constexpr int F4 = C / 4, ROWS = T + 2, N = S * ROWS * F4, NIT = (N + NTHR - 1) / NTHR;
  const int pos = ring_pos(a.in, hop);
  float4 v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = tid + it * NTHR;
    const int s = e / (ROWS * F4), q = e % (ROWS * F4), t = q / F4 - 2, c4 = q % F4, b = b0 + s;
    v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < N && live_s<RAG>(shop, b0, s, a.B)) {
      if (IN_FROM_RING_HISTORY || t + fb >= 0) v[it] = *reinterpret_cast<const float4*>(ring_frame(a.in, b, stream_pos<RAG>(a.in, pos, shop, s), t + fb) + 4 * c4);
      else v[it] = *reinterpret_cast<const float4*>(a.state + (size_t)b * TAIL_STATE_FLOATS + TS_IN + (t + 2) * C + 4 * c4);
    }
  }
  float4 hb = make_float4(0.f, 0.f, 0.f, 0.f), hcv = hb;
  if (first) {
    hb = state_load<S, 6 * C, RAG>(a.state, TS_B, b0, shop, a.B, tid);
    hcv = state_load<S, HC_ROWS * C, RAG>(a.state, TS_C, b0, shop, a.B, tid);
  }
  // ---- every load above is in flight; now the stores
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = tid + it * NTHR;
    if (e < N) {
      const int s = e / (ROWS * F4), q = e % (ROWS * F4), t = q / F4 - 2, c4 = q % F4;
      store_act4(X + row_off<C, T>(s, t) + 4 * c4, v[it]);
      if (!IN_FROM_RING_HISTORY && last && t >= T - 2) *reinterpret_cast<float4*>(hin + (s * 2 + (t - (T - 2))) * C + 4 * c4) = v[it];
    }
  }
  if (first) {
    if (tid < S * 6 * C / 4) {   // (zeros past the batch)
      const int s = tid / (6 * C / 4), q = tid % (6 * C / 4);
      store_act4(Y + row_off<C, T>(s, -6 + (4 * q) / C) + (4 * q) % C, hb);
    }
    if (tid < S * HC_ROWS * C / 4) {
      float4 w;
      w.x = lrelu_max(hcv.x); w.y = lrelu_max(hcv.y); w.z = lrelu_max(hcv.z); w.w = lrelu_max(hcv.w);
      *reinterpret_cast<float4*>(hc + 4 * tid) = w;
    }
  }
}
// between two sub-steps: what the next one's prologue would have read from the state block, taken from this one's LDS buffers --
// the second layer's six history rows = the last six rows of Y (activated y), the third layer's HC_ROWS = the last rows of X
// (activated z).  The caller brackets it with barriers (every wavefront is done with X and Y; nobody has started the next prologue).
template <int C, int T, int S, int HC_ROWS>
__device__ __forceinline__ void carry_histories(float* __restrict__ X, float* __restrict__ Y, float* __restrict__ hc, const int tid) {
  static_assert(T >= 12, "history rows and their sources do not overlap");
  for (int e = tid; e < S * 6 * C; e += NTHR) {
    const int s = e / (6 * C), q = e % (6 * C);
    Y[row_off<C, T>(s, -6 + q / C) + q % C] = Y[row_off<C, T>(s, T - 6 + q / C) + q % C];
  }
  for (int e = tid; e < S * HC_ROWS * C; e += NTHR) {
    const int s = e / (HC_ROWS * C), q = e % (HC_ROWS * C);
    hc[e] = X[row_off<C, T>(s, T - HC_ROWS + q / C) + q % C];
  }
}
// a stash [S][ROWS][C] (contiguous) -> rows [t_first, t_first + ROWS) of every stream's block (LDS to LDS)
template <int C, int T, int S, int ROWS>
__device__ __forceinline__ void stash_to_rows(float* __restrict__ buf, const float* __restrict__ stash, const int t_first, const int tid) {
  for (int e = tid; e < S * ROWS * C; e += NTHR) {
    const int s = e / (ROWS * C), q = e % (ROWS * C);
    buf[row_off<C, T>(s, t_first + q / C) + q % C] = stash[e];
  }
}
template <int C, int S, int TS_IN, bool RAG>
__device__ __forceinline__ void hin_to_state(const StageArgs& a, const float* __restrict__ hin, const int b0, const int tid, const int* shop) {
  for (int e = tid; e < S * 2 * C / 4; e += NTHR) {
    const int s = e / (2 * C / 4), q = e % (2 * C / 4);
    if (live_s<RAG>(shop, b0, s, a.B)) *reinterpret_cast<float4*>(a.state + (size_t)(b0 + s) * TAIL_STATE_FLOATS + TS_IN + 4 * q) = *reinterpret_cast<const float4*>(hin + 4 * e);
  }
}

// The two residual layers every stage starts with (k3; dilation 1: X -> Y, dilation 3: Y -> X), MODEL_SPEC 4.4.3:
//   y = x + (conv(lrelu(x)) + bias)      x raw from the stage's input ring, y raw STAYS IN REGISTERS (its last 6 frames -> TS_B)
//   z = y + (conv(lrelu(y)) + bias)      y raw from those registers; of z only lrelu(z) is used again (-> X) and its last HC_ROWS
//                                        frames, raw, as the next step's history (-> TS_C)
// after_a() / between(): called right after the first layer's MFMAs and after the barrier that follows (the caller's fetches of
// the next layers' weights, see fetch_b).
template <int C, int T, int S, int TS_B, int TS_C, int HC_ROWS, bool RAG, class AfterA, class Between>
__device__ __forceinline__ void two_res_layers(const StageArgs& a, const int hop, const int b0, const int n_rows, float* __restrict__ X, float* __restrict__ Y,
                                               const float* __restrict__ hc, const float4 (&bfa)[Split<C>::CT][3 * C / 16],
                                               const float4 (&bfb)[Split<C>::CT][3 * C / 16], const int wave, const int lane, const int tid, AfterA after_a, Between between,
                                               const int* shop, float* __restrict__ yr = nullptr /* C > 32: LDS [S][T][C], the raw y (registers are short at 64 channels) */,
                                               const int fb = 0 /* sub-steps: first frame inside the step */, const bool last = true /* the step's last sub-step: histories -> state block */) {
  static_assert(Split<C>::CT == 1, "one column tile per wavefront");
  constexpr bool IN_LDS = C > 32;
  const int n_lane = (wave % Split<C>::NWN) * 16 + (lane & 15);
  const float bias_a = a.b[0][n_lane], bias_b = a.b[1][n_lane];
  const int pos = ring_pos(a.in, hop);
  float keep[n_pass<C, T, S>()][2][Split<C>::CT][4];
  layer<C, C, 3, 1, T, S, (C <= 32 ? 1 : 2)>(X, bfa, n_rows, wave, lane, keep,
    [&](int s, int t0, int n, int, float (&res)[4]) {
      if constexpr (RAG) {
        if (shop[s] < 0) { res[0] = res[1] = res[2] = res[3] = 0.0f; return; }   // (a stream that sits the step out: its rows compute on zeros, nothing of it is written)
      }
      const float* p = ring_frame(a.in, b0 + s, stream_pos<RAG>(a.in, pos, shop, s), t0 + fb) + n;   // (frames t0 .. t0 + 3 of a step are contiguous in its ring slot)
#pragma unroll
      for (int e = 0; e < 4; ++e) res[e] = p[e * C];
    },
    [&](int s, int t0, int n, int, const tail_f32x4& acc, float (&res)[4]) {
      float* yo = Y + row_off<C, T>(s, t0) + n;
      float* st = a.state + (size_t)(b0 + s) * TAIL_STATE_FLOATS + TS_B + (t0 - (T - 6)) * C + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = res[e] + (acc[e] + bias_a);
        yo[e * cs<C>()] = lrelu_max(y);
        if (IN_LDS) yr[((s * T) + t0 + e) * C + n] = y; else res[e] = y;   // the second layer's residual
        if (last && t0 + e >= T - 6 && (!RAG || shop[s] >= 0)) st[e * C] = y;
      }
    });
  after_a();
  TST_STAMP(3);
  __syncthreads();
  TST_STAMP(4);
  between();
  stash_to_rows<C, T, S, HC_ROWS>(X, hc, -HC_ROWS, tid);
  layer<C, C, 3, 3, T, S, (IN_LDS ? 2 : 0)>(Y, bfb, n_rows, wave, lane, keep,
    [&](int s, int t0, int n, int, float (&res)[4]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) res[e] = yr[((s * T) + t0 + e) * C + n];
    },
    [&](int s, int t0, int n, int, const tail_f32x4& acc, float (&res)[4]) {
      float* xo = X + row_off<C, T>(s, t0) + n;
      float* st = a.state + (size_t)(b0 + s) * TAIL_STATE_FLOATS + TS_C + (t0 - (T - HC_ROWS)) * C + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = res[e] + (acc[e] + bias_b);
        xo[e * cs<C>()] = lrelu_max(z);
        if (last && t0 + e >= T - HC_ROWS && (!RAG || shop[s] >= 0)) st[e * C] = z;
      }
    });
  TST_STAMP(5);
  __syncthreads();
  TST_STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------------------
// T1 and T2: the two residual layers, then the polyphase transposed conv (k2 over input frames, rate UPR, COUT channels)
// into the next stage's ring.  IN_FROM_RING_HISTORY: the first layer's two history frames come from the input ring itself
// (T1: the ring of up2 keeps them); otherwise from the state block at TS_IN (T2).
template <int C, int T, int S, int COUT, int UPR, int TS_IN, int TS_B, int TS_C, bool IN_FROM_RING_HISTORY, bool RAG = false, int NSUB = 1>
__device__ __forceinline__ void res_res_up_body(const StageArgs& a, const int g, float* __restrict__ lds) {
  // (T = frames per SUB-step; the step has NSUB * T frames per stream, prologue())
  constexpr int NUP = UPR * COUT;
  float* X = lds;
  float* Y = X + buf_floats<C, T, S>();
  float* HIN = Y + buf_floats<C, T, S>();   // [S][2][C] raw
  float* HC = HIN + S * 2 * C;              // [S][1][C] activated
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
#ifdef TST_SETPRIO
  __builtin_amdgcn_s_setprio(TST_SETPRIO);   // experiment: issue priority over the co-resident workgroup's wavefronts
#endif
  const int b0 = g * S;
  const int n_rows = (a.B - b0 < S ? a.B - b0 : S) * T;
  int* shop = reinterpret_cast<int*>(lds + stage_lds<C, T, S, 1>() - 8);
  auto sub_step = [&](const int fb, const bool first, const bool last) {
    float4 bfa[Split<C>::CT][3 * C / 16], bfb[Split<C>::CT][3 * C / 16], bfu[Split<NUP>::CT][2 * C / 16];
    TST_STAMP(0);
    fetch_b<3 * C, C>(a.w[0], bfa, wave, lane);
    if (first) stream_hops<S, RAG>(a, hop, b0, shop);
    prologue<C, T, S, TS_IN, TS_B, TS_C, 1, IN_FROM_RING_HISTORY, RAG>(a, hop, b0, X, Y, HIN, HC, tid, shop, fb, first, last);
    TST_STAMP(1);
    __syncthreads();
    TST_STAMP(2);
    constexpr int KBR = 3 * C / 16, KBU = 2 * C / 16;
    fetch_b<3 * C, C, 0, KBR / 2>(a.w[1], bfb, wave, lane);
    two_res_layers<C, T, S, TS_B, TS_C, 1, RAG>(a, hop, b0, n_rows, X, Y, HC, bfa, bfb, wave, lane, tid,
                                           [&] { fetch_b<3 * C, C, KBR / 2, KBR>(a.w[1], bfb, wave, lane); },
                                           [&] { fetch_b<2 * C, NUP, 0, KBU / 2>(a.w[2], bfu, wave, lane); }, shop, C > 32 ? HC + S * C : nullptr, fb, last);
    fetch_b<2 * C, NUP, KBU / 2, KBU>(a.w[2], bfu, wave, lane);
    // ---- transposed conv (polyphase k2): X -> the next stage's ring, frame t UPR + n / COUT, channel n % COUT
    {
      using SU = Split<NUP>;
      float bias_u[SU::CT];
#pragma unroll
      for (int ct = 0; ct < SU::CT; ++ct) bias_u[ct] = a.b[2][(SU::POW2 ? wave % SU::NWN : ct) * 16 + (lane & 15)];
      const int pos_o = ring_pos(a.out, hop);
      float none[n_pass<NUP, T, S>()][2][SU::CT][4];   // (no residual in this layer: never read, costs no registers)
      layer<C, NUP, 2, 1, T, S, 0>(X, bfu, n_rows, wave, lane, none, [](int, int, int, int, float (&)[4]) {},
        [&](int s, int t0, int n, int ct, const tail_f32x4& acc, float (&)[4]) {
          const float bu = bias_u[ct];   // (ct is a compile-time index after unrolling)
          if constexpr (RAG) { if (shop[s] < 0) return; }
          float* o = ring_frame(a.out, b0 + s, stream_pos<RAG>(a.out, pos_o, shop, s), (t0 + fb) * UPR + n / COUT) + n % COUT;   // (output frames of one input frame are UPR apart)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e * UPR * COUT] = acc[e] + bu;
        });
    }
    TST_STAMP(7);
    if constexpr (NSUB > 1) {
      if (!last) {
        __syncthreads();
        carry_histories<C, T, S, 1>(X, Y, HC, tid);
        __syncthreads();
      }
    }
  };
  if constexpr (NSUB == 1) sub_step(0, true, true);
  else {
#pragma unroll 1
    for (int sub = 0; sub < NSUB; ++sub) sub_step(sub * T, sub == 0, sub == NSUB - 1);
  }
  if (!IN_FROM_RING_HISTORY) hin_to_state<C, S, TS_IN, RAG>(a, HIN, b0, tid, shop);
  TST_STAMP(8);
}

constexpr int kT1Streams = 4, kT2Streams = 3, kT3Streams = 2;
// S = streams per workgroup.  The tick launch's table for FULL ticks uses 4 (T1) and 3 (T2): fewer, fuller workgroups; its table
// for the sparse ticks of fill and drain uses 2: half as long, and a sparse launch lasts as long as its longest workgroup
// (profiles/r04_notes.md section 4).  Same arithmetic, same state block, same rings: interchangeable from one tick to the next.
// HOPS = hops per step (tick mode with several hops per stage, tick.hip.h): a stream's frames of a step are HOPS times as many --
// the same bodies with T = HOPS x the per-hop count (the histories a layer keeps across steps do not depend on T).
template <int S = kT1Streams, int HOPS = 1>
struct T1OpS {
  using Args = StageArgs;
  static constexpr int T = 20 * HOPS;
  static constexpr int NTHR = tst::NTHR, LDS_FLOATS = stage_lds<64, T, S, 1>();
  static inline dim3 grid(const Args& a) { return dim3((a.B + S - 1) / S, 1); }
  static inline bhip::LaunchInfo info(const Args& a) {
    const double macs = 2.0 * T * 192 * 64 + 1.0 * T * 128 * 128;
    return bhip::LaunchInfo{"wave.tail1", 2.0 * a.B * macs, 4.0 * (2.0 * 192 * 64 + 128.0 * 128 + a.B * ((T + 2.0) * 64 + 4 * T * 32 + 2 * 7 * 64))};
  }
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) {
    res_res_up_body<64, T, S, 32, 4, 0, TS_YB2, TS_YC2, true>(a, bx, lds);
  }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int, float* lds) {
    res_res_up_body<64, T, S, 32, 4, 0, TS_YB2, TS_YC2, true, RAG>(a, bx, lds);
  }
};
using T1Op = T1OpS<kT1Streams>;
// NSUB: the step's 80 HOPS frames per stream in NSUB sub-steps, one after the other inside the workgroup (res_res_up_body)
template <int S = kT2Streams, int HOPS = 1, int NSUB = 1>
struct T2OpS {
  using Args = StageArgs;
  static_assert(HOPS % NSUB == 0, "whole hops per sub-step");
  static constexpr int T = 80 * HOPS;          // frames per stream and step
  static constexpr int TSUB = T / NSUB;        // ... per sub-step: what the LDS holds
  static constexpr int NTHR = tst::NTHR, LDS_FLOATS = stage_lds<32, TSUB, S, 1>();
  static inline dim3 grid(const Args& a) { return dim3((a.B + S - 1) / S, 1); }
  static inline bhip::LaunchInfo info(const Args& a) {
    const double macs = 2.0 * T * 96 * 32 + 1.0 * T * 64 * 48;
    return bhip::LaunchInfo{"wave.tail2", 2.0 * a.B * macs, 4.0 * (2.0 * 96 * 32 + 64.0 * 48 + a.B * (1.0 * T * 32 + 3 * T * 16 + 2 * 9 * 32))};
  }
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) {
    res_res_up_body<32, TSUB, S, 16, 3, TS_YA3, TS_YB3, TS_YC3, false, false, NSUB>(a, bx, lds);
  }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int, float* lds) {
    res_res_up_body<32, TSUB, S, 16, 3, TS_YA3, TS_YB3, TS_YC3, false, RAG, NSUB>(a, bx, lds);
  }
};
using T2Op = T2OpS<kT2Streams>;
constexpr int kT1Lds = T1Op::LDS_FLOATS, kT2Lds = T2Op::LDS_FLOATS;
template <int S, int HOPS> constexpr int t3_lds() { return stage_lds<16, 240 * HOPS, S, 6>() + 7 * 16; }
constexpr int kT3Lds = t3_lds<kT3Streams, 1>();

// ---------------------------------------------------------------------------------------------------------------------
// T3: res4a, res4b (16 channels, 240 frames per stream) and the output conv: lrelu, Conv1d(16 -> 1, k7), tanh.
template <bool RAG = false, int S = kT3Streams, int HOPS = 1, int NSUB = 1>
__device__ __forceinline__ void t3_body(const StageArgs& a, const int g, float* __restrict__ lds) {
  static_assert(HOPS % NSUB == 0, "whole hops per sub-step");
  constexpr int C = 16, T = 240 * HOPS / NSUB;   // T = frames per SUB-step (prologue()): what the LDS holds
  static_assert(S * T <= NTHR, "one thread per output sample");
  float* X = lds;
  float* Y = X + buf_floats<C, T, S>();
  float* HIN = Y + buf_floats<C, T, S>();   // [S][2][C] raw
  float* HC = HIN + S * 2 * C;              // [S][6][C] activated: history of the output conv's input
  float* FW = HC + S * 6 * C + 8;   // (behind the streams' step counters)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
#ifdef TST_SETPRIO
  __builtin_amdgcn_s_setprio(TST_SETPRIO);
#endif
  const int io = a.io_stride != 0 ? stepc::slot(a.hop) : 0;
  float* __restrict__ d_out = a.d_out + (size_t)io * a.io_stride;
  const int b0 = g * S;
  const int n_rows = (a.B - b0 < S ? a.B - b0 : S) * T;
  int* shop = reinterpret_cast<int*>(lds + stage_lds<C, T, S, 6>() - 8);
  stream_hops<S, RAG>(a, hop, b0, shop);
  float4 bfa[1][3], bfb[1][3];
  fetch_b<48, 16>(a.w[0], bfa, wave, lane);
  fetch_b<48, 16>(a.w[1], bfb, wave, lane);
  const float fin_b = a.fin_b[0];
  const float fw = tid < 7 * 16 ? a.fin_w[tid] : 0.0f;
#pragma unroll 1
  for (int sub = 0; sub < NSUB; ++sub) {
    const int fb = NSUB > 1 ? sub * T : 0;
    const bool first = NSUB == 1 || sub == 0, last = NSUB == 1 || sub == NSUB - 1;
    prologue<C, T, S, TS_YA4, TS_YB4, TS_YC4, 6, false, RAG>(a, hop, b0, X, Y, HIN, HC, tid, shop, fb, first, last);
    if (first && tid < 7 * 16) FW[tid] = fw;
    __syncthreads();
    two_res_layers<C, T, S, TS_YB4, TS_YC4, 6, RAG>(a, hop, b0, n_rows, X, Y, HC, bfa, bfb, wave, lane, tid, [] {}, [] {}, shop, nullptr, fb, last);
    // ---- output conv over the ACTIVATED frames: one thread per sample, the multiply-adds of wave_tail.hip.h in the same order
    if (tid < S * T) {
      const int s = tid / T, t = tid % T;
      if (live_s<RAG>(shop, b0, s, a.B)) {
        float acc = 0.0f;
        const float* x = X + row_off<C, T>(s, t - 6);
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
          for (int c = 0; c < 16; ++c) acc = bsp::fma(x[j * cs<C>() + c], FW[j * 16 + c], acc);
        d_out[(size_t)(b0 + s) * (T * NSUB) + fb + t] = bsp::tanh2(bsp::splat2(acc + fin_b)).x;   // (the packed form is the shorter one even for a single value)
      }
    }
    if constexpr (NSUB > 1) {
      if (!last) {
        __syncthreads();
        carry_histories<C, T, S, 6>(X, Y, HC, tid);
        __syncthreads();
      }
    }
  }
  hin_to_state<C, S, TS_YA4, RAG>(a, HIN, b0, tid, shop);
}
template <int S = kT3Streams, int HOPS = 1, int NSUB = 1>
struct T3OpS {
  using Args = StageArgs;
  static constexpr int T = 240 * HOPS;
  static constexpr int NTHR = tst::NTHR, LDS_FLOATS = t3_lds<S, HOPS / NSUB>();
  static inline dim3 grid(const Args& a) { return dim3((a.B + S - 1) / S, 1); }
  static inline bhip::LaunchInfo info(const Args& a) {
    const double macs = 2.0 * T * 48 * 16 + 1.0 * T * 112;
    return bhip::LaunchInfo{"wave.tail3", 2.0 * a.B * macs, 4.0 * (2.0 * 48 * 16 + 112.0 + a.B * (1.0 * T * 16 + T + 2 * 14 * 16))};
  }
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) { t3_body<false, S, HOPS, NSUB>(a, bx, lds); }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int, float* lds) { t3_body<RAG, S, HOPS, NSUB>(a, bx, lds); }
};
using T3Op = T3OpS<kT3Streams, 1>;

}  // namespace tst
