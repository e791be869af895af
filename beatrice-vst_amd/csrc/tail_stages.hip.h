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
// epilogue is paid once per 2-4 streams, and the LDS holds two ping-pong buffers of RAW activations only (lrelu is two VALU
// operations on the A operand on its way to the MFMA: max(x, 0.1 x), hidden behind the 32-cycle MFMA issue).
// More stages cost a throughput pipeline nothing (a step's latency grows by two ticks).
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

// LDS buffer of S streams: [S][HP + T][C + 2] raw values; row (s, t), t in [-HP, T).  Stride C + 2: the A-operand read of
// a wavefront (16 rows x 4 k) hits 32 distinct banks per half (bank = 2 row + k, as in wave_tail.hip.h)
template <int C> __host__ __device__ constexpr int cs() { return C + 2; }
template <int C, int T, int S> __host__ __device__ constexpr int buf_floats() { return S * (HP + T) * cs<C>(); }
template <int C, int T> __device__ __forceinline__ int row_off(int s, int t) { return (s * (HP + T) + HP + t) * cs<C>(); }

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

template <int K, int NOUT>
__device__ __forceinline__ void fetch_b(const float* __restrict__ wpacked, float4 (&bf)[Split<NOUT>::CT][K / 16], int wave, int lane) {
  using SP = Split<NOUT>;
  const int wn = wave % SP::NWN;
#pragma unroll
  for (int ct = 0; ct < SP::CT; ++ct) {
    const int nt = SP::POW2 ? wn : ct;
    const float4* p = reinterpret_cast<const float4*>(wpacked) + (size_t)nt * (K / 16) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < K / 16; ++kb) bf[ct][kb] = p[(size_t)kb * 64];
  }
}

#ifndef TST_PIN
#define TST_PIN 1
#endif
// One pass of a layer: NT (1 or 2, compile time) row tiles of 16 that share every B operand.  base[u] = the lane's A address
// of tile u (row lane & 15 of the tile, k offset lane >> 4); the reduction runs in groups of four MFMA steps (16 k = one B
// fragment record), the raw A operands of group g + 1 are read from LDS BEFORE the MFMAs of group g issue and the order is
// pinned (sched_barrier; the asm keeps IR passes from undoing it), lrelu = max(x, 0.1 x) is applied on the way.
template <int CIN, int NOUT, int KSZ, int DIL, int NT>
__device__ __forceinline__ void pass(const float* __restrict__ in, const int (&base)[2], const float4 (&bf)[Split<NOUT>::CT][KSZ * CIN / 16],
                                     tail_f32x4 (&acc)[2][Split<NOUT>::CT]) {
  using SP = Split<NOUT>;
  constexpr int CS = cs<CIN>(), NS = KSZ * CIN / 4, NG = NS / 4;
  static_assert(NS % 4 == 0, "reduction length in blocks of 16");
  auto a_off = [](int ks) { const int kk = ks * 4, j = kk / CIN, c = kk % CIN; return c - (KSZ - 1 - j) * DIL * CS; };
  // (two groups ahead: the scheduler places a group's LDS reads at the END of the region they are issued in, behind that
  //  region's MFMAs, so one group of distance leaves them no time to complete)
  float xq[2][4][NT];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int u = 0; u < NT; ++u) xq[q][e][u] = q < NG ? in[base[u] + a_off(q * 4 + e)] : 0.0f;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    float ac[4][NT];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int u = 0; u < NT; ++u) ac[e][u] = fmaxf(xq[g & 1][e][u], 0.1f * xq[g & 1][e][u]);  // == lrelu(x) bit for bit
    if (g + 2 < NG) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int u = 0; u < NT; ++u) xq[g & 1][e][u] = in[base[u] + a_off((g + 2) * 4 + e)];
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
// (taps are row offsets inside a stream's block, history rows included), N = NOUT.  epi(s, t, n, acc) receives the
// finished chain of output (stream s, frame t, column n); rows >= n_rows are padding (recomputed, never handed out).
template <int CIN, int NOUT, int KSZ, int DIL, int T, int S, class Epi>
__device__ __forceinline__ void layer(const float* __restrict__ in, const float4 (&bf)[Split<NOUT>::CT][KSZ * CIN / 16], const int n_rows,
                                      const int wave, const int lane, Epi epi) {
  using SP = Split<NOUT>;
  constexpr int R = S * T, NRT = (R + 15) / 16;
  const int i = lane & 15, kq = lane >> 4;
  // (the wavefront index as a SCALAR: the row-tile loop and its "second tile?" test must be scalar branches)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wn = wave_u % SP::NWN, wm = wave_u / SP::NWN;
#pragma unroll 1
  for (int t0 = wm; t0 < NRT; t0 += 2 * SP::NWM) {
    const bool two = t0 + SP::NWM < NRT;  // a second row tile shares every B operand of this pass
    int base[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int r = (t0 + u * SP::NWM) * 16 + i;
      r = r > R - 1 ? R - 1 : r;
      base[u] = row_off<CIN, T>(r / T, r % T) + kq;
    }
    tail_f32x4 acc[2][SP::CT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int ct = 0; ct < SP::CT; ++ct) acc[u][ct] = tail_f32x4{0.f, 0.f, 0.f, 0.f};
    if (two) pass<CIN, NOUT, KSZ, DIL, 2>(in, base, bf, acc);
    else pass<CIN, NOUT, KSZ, DIL, 1>(in, base, bf, acc);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
#pragma unroll
      for (int ct = 0; ct < SP::CT; ++ct) {
        const int n = (SP::POW2 ? wn : ct) * 16 + i;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = (t0 + u * SP::NWM) * 16 + kq * 4 + e;   // D layout of 16x16x4: row = (lane >> 4) * 4 + reg, column = lane & 15
          if (r < n_rows) epi(r / T, r % T, n, acc[u][ct][e]);
        }
      }
    }
  }
}

// ---- global memory <-> LDS, latency-aware.  A workgroup of these stages is a serial chain of phases; inside the tick launch
// a global round trip takes 1-2 us (the memory system is shared with ~500 other workgroups), so a body may afford very few
// of them on its critical path: EVERYTHING a stage reads from global memory -- its input frames, its three pieces of the
// state block, its biases -- is requested in one burst before the first barrier (every thread issues all of its loads, then
// stores them to LDS), and everything it writes to the state block goes out after the last layer, behind nothing.

// S x N floats (N a multiple of 4, S * N / 4 <= 512: one float4 per thread) of the state block at `ts_off` -> dst[s * N ...]
template <int S, int N>
__device__ __forceinline__ float4 state_load(const float* __restrict__ state, const int ts_off, const int b0, const int B, const int tid, bool* live) {
  static_assert(N % 4 == 0 && S * N / 4 <= NTHR, "one float4 per thread");
  const int s = tid / (N / 4), q = tid % (N / 4);
  *live = tid < S * N / 4 && b0 + s < B;
  return *live ? *reinterpret_cast<const float4*>(state + (size_t)(b0 + s) * TAIL_STATE_FLOATS + ts_off + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
}
// rows [t_first, t_first + ROWS) of every stream's block in `buf` -> the state block
template <int C, int T, int S, int ROWS>
__device__ __forceinline__ void state_store_rows(float* __restrict__ state, const int ts_off, const float* __restrict__ buf, const int t_first, const int b0,
                                                 const int B, const int tid) {
  constexpr int F2 = C / 2;
  for (int e = tid; e < S * ROWS * F2; e += NTHR) {
    const int s = e / (ROWS * F2), q = e % (ROWS * F2), row = q / F2, c2 = q % F2;
    if (b0 + s < B)
      *reinterpret_cast<float2*>(state + (size_t)(b0 + s) * TAIL_STATE_FLOATS + ts_off + row * C + 2 * c2) =
          *reinterpret_cast<const float2*>(buf + row_off<C, T>(s, t_first + row) + 2 * c2);
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

// The prologue shared by the three stages: weights of the first layer, the step's input frames -> X rows [-2, T) (history
// rows from the ring itself, or from the state block at TS_IN -- then the frames that will be the NEXT step's history are
// stashed in `hin` and reach the state block at the end), the six history rows of the second layer -> Y rows [-6, 0), the
// history of the third layer -> stash `hc`, and NB bias floats -> `bias_lds`.
template <int C, int T, int S, int TS_IN, int TS_B, int TS_C, int HC_ROWS, bool IN_FROM_RING_HISTORY, int NB0, int NB1, int NB2>
__device__ __forceinline__ void prologue(const StageArgs& a, const int hop, const int b0, float* __restrict__ X, float* __restrict__ Y, float* __restrict__ hin,
                                         float* __restrict__ hc, float* __restrict__ bias_lds, const int tid) {
  constexpr int F4 = C / 4, ROWS = T + 2, N = S * ROWS * F4, NIT = (N + NTHR - 1) / NTHR;
  const int pos = ring_pos(a.in, hop);
  float4 v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = tid + it * NTHR;
    const int s = e / (ROWS * F4), q = e % (ROWS * F4), t = q / F4 - 2, c4 = q % F4, b = b0 + s;
    v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < N && b < a.B) {
      if (IN_FROM_RING_HISTORY || t >= 0) v[it] = *reinterpret_cast<const float4*>(ring_frame(a.in, b, pos, t) + 4 * c4);
      else v[it] = *reinterpret_cast<const float4*>(a.state + (size_t)b * TAIL_STATE_FLOATS + TS_IN + (t + 2) * C + 4 * c4);
    }
  }
  bool live_b, live_c;
  const float4 hb = state_load<S, 6 * C>(a.state, TS_B, b0, a.B, tid, &live_b);
  const float4 hcv = state_load<S, HC_ROWS * C>(a.state, TS_C, b0, a.B, tid, &live_c);
  constexpr int NB = NB0 + NB1 + NB2;
  static_assert(NB <= NTHR, "biases: one float per thread");
  float bv = 0.0f;
  if (tid < NB) bv = tid < NB0 ? a.b[0][tid] : (tid < NB0 + NB1 ? a.b[1][tid - NB0] : a.b[2][tid - NB0 - NB1]);
  // ---- every load above is in flight; now the stores
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = tid + it * NTHR;
    if (e < N) {
      const int s = e / (ROWS * F4), q = e % (ROWS * F4), t = q / F4 - 2, c4 = q % F4;
      float* d = X + row_off<C, T>(s, t) + 4 * c4;   // (row stride C + 2: 8-byte aligned, not 16)
      reinterpret_cast<float2*>(d)[0] = make_float2(v[it].x, v[it].y);
      reinterpret_cast<float2*>(d)[1] = make_float2(v[it].z, v[it].w);
      if (!IN_FROM_RING_HISTORY && t >= T - 2) *reinterpret_cast<float4*>(hin + (s * 2 + (t - (T - 2))) * C + 4 * c4) = v[it];
    }
  }
  if (tid < S * 6 * C / 4) {   // (zeros past the batch)
    const int s = tid / (6 * C / 4), q = tid % (6 * C / 4);
    float* d = Y + row_off<C, T>(s, -6 + (4 * q) / C) + (4 * q) % C;
    reinterpret_cast<float2*>(d)[0] = make_float2(hb.x, hb.y);
    reinterpret_cast<float2*>(d)[1] = make_float2(hb.z, hb.w);
  }
  if (tid < S * HC_ROWS * C / 4) *reinterpret_cast<float4*>(hc + 4 * tid) = hcv;
  if (tid < NB) bias_lds[tid] = bv;
}

// ---------------------------------------------------------------------------------------------------------------------
// The common shape of T1 and T2: two residual convs (k3; dilation 1, 3) over C channels, then the polyphase transposed conv
// (k2 over input frames, rate UPR, COUT channels) into the next stage's ring.  IN_FROM_RING_HISTORY: the first layer's
// two history frames come from the input ring itself (T1: the ring of up2 keeps them); otherwise from the state block at
// TS_IN, which then receives this step's last two input frames (T2).
template <int C, int T, int S, int COUT, int UPR, int TS_IN, int TS_B, int TS_C, bool IN_FROM_RING_HISTORY>
__device__ __forceinline__ void res_res_up_body(const StageArgs& a, const int g, float* __restrict__ lds) {
  constexpr int NUP = UPR * COUT;
  float* X = lds;
  float* Y = X + buf_floats<C, T, S>();
  float* HIN = Y + buf_floats<C, T, S>();   // [S][2][C]
  float* HC = HIN + S * 2 * C;              // [S][1][C]
  float* BIAS = HC + S * C;                 // C | C | NUP
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
#ifdef TST_SETPRIO
  __builtin_amdgcn_s_setprio(TST_SETPRIO);   // experiment: issue priority over the co-resident workgroup's wavefronts
#endif
  const int b0 = g * S;
  const int n_rows = (a.B - b0 < S ? a.B - b0 : S) * T;
  float4 bfa[Split<C>::CT][3 * C / 16], bfb[Split<C>::CT][3 * C / 16], bfu[Split<NUP>::CT][2 * C / 16];
  TST_STAMP(0);
  fetch_b<3 * C, C>(a.w[0], bfa, wave, lane);
  prologue<C, T, S, TS_IN, TS_B, TS_C, 1, IN_FROM_RING_HISTORY, C, C, NUP>(a, hop, b0, X, Y, HIN, HC, BIAS, tid);
  TST_STAMP(1);
  __syncthreads();
  TST_STAMP(2);
  // ---- resA (k3, dilation 1): X -> Y
  fetch_b<3 * C, C>(a.w[1], bfb, wave, lane);
  layer<C, C, 3, 1, T, S>(X, bfa, n_rows, wave, lane, [&](int s, int t, int n, float v) {
    const int o = row_off<C, T>(s, t) + n;
    Y[o] = X[o] + (v + BIAS[n]);
  });
  TST_STAMP(3);
  __syncthreads();
  TST_STAMP(4);
  // ---- resB (k3, dilation 3): Y -> X (X's history row -1 <- the stash: input history of the transposed conv)
  fetch_b<2 * C, NUP>(a.w[2], bfu, wave, lane);
  stash_to_rows<C, T, S, 1>(X, HC, -1, tid);
  layer<C, C, 3, 3, T, S>(Y, bfb, n_rows, wave, lane, [&](int s, int t, int n, float v) {
    const int o = row_off<C, T>(s, t) + n;
    X[o] = Y[o] + (v + BIAS[C + n]);
  });
  TST_STAMP(5);
  __syncthreads();
  TST_STAMP(6);
  // ---- transposed conv (polyphase k2): X -> the next stage's ring, frame t UPR + n / COUT, channel n % COUT
  {
    const int pos_o = ring_pos(a.out, hop);
    layer<C, NUP, 2, 1, T, S>(X, bfu, n_rows, wave, lane, [&](int s, int t, int n, float v) {
      ring_frame(a.out, b0 + s, pos_o, t * UPR + n / COUT)[n % COUT] = v + BIAS[2 * C + n];
    });
  }
  TST_STAMP(7);
  // ---- the histories of the next step -> state block (Y and X still hold this step's resA / resB outputs)
  state_store_rows<C, T, S, 6>(a.state, TS_B, Y, T - 6, b0, a.B, tid);
  state_store_rows<C, T, S, 1>(a.state, TS_C, X, T - 1, b0, a.B, tid);
  if (!IN_FROM_RING_HISTORY)
    for (int e = tid; e < S * 2 * C / 4; e += NTHR) {
      const int s = e / (2 * C / 4), q = e % (2 * C / 4);
      if (b0 + s < a.B) *reinterpret_cast<float4*>(a.state + (size_t)(b0 + s) * TAIL_STATE_FLOATS + TS_IN + 4 * q) = *reinterpret_cast<const float4*>(HIN + 4 * e);
    }
  TST_STAMP(8);
}

constexpr int kT1Streams = 4, kT2Streams = 3, kT3Streams = 2;
template <int C, int T, int S, int NUP> constexpr int rru_lds() { return 2 * buf_floats<C, T, S>() + S * 3 * C + 2 * C + NUP; }
constexpr int kT1Lds = rru_lds<64, 20, kT1Streams, 128>();
constexpr int kT2Lds = rru_lds<32, 80, kT2Streams, 48>();
constexpr int kT3Lds = 2 * buf_floats<16, 240, kT3Streams>() + kT3Streams * 8 * 16 + 2 * 16 + 7 * 16;

struct T1Op {
  using Args = StageArgs;
  static constexpr int NTHR = tst::NTHR, LDS_FLOATS = kT1Lds;
  static inline dim3 grid(const Args& a) { return dim3((a.B + kT1Streams - 1) / kT1Streams, 1); }
  static inline bhip::LaunchInfo info(const Args& a) {
    const double macs = 2.0 * 20 * 192 * 64 + 20.0 * 128 * 128;
    return bhip::LaunchInfo{"wave.tail1", 2.0 * a.B * macs, 4.0 * (2.0 * 192 * 64 + 128.0 * 128 + a.B * (22.0 * 64 + 80 * 32 + 2 * 7 * 64))};
  }
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) {
    res_res_up_body<64, 20, kT1Streams, 32, 4, 0, TS_YB2, TS_YC2, true>(a, bx, lds);
  }
};
struct T2Op {
  using Args = StageArgs;
  static constexpr int NTHR = tst::NTHR, LDS_FLOATS = kT2Lds;
  static inline dim3 grid(const Args& a) { return dim3((a.B + kT2Streams - 1) / kT2Streams, 1); }
  static inline bhip::LaunchInfo info(const Args& a) {
    const double macs = 2.0 * 80 * 96 * 32 + 80.0 * 64 * 48;
    return bhip::LaunchInfo{"wave.tail2", 2.0 * a.B * macs, 4.0 * (2.0 * 96 * 32 + 64.0 * 48 + a.B * (80.0 * 32 + 240 * 16 + 2 * 9 * 32))};
  }
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) {
    res_res_up_body<32, 80, kT2Streams, 16, 3, TS_YA3, TS_YB3, TS_YC3, false>(a, bx, lds);
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// T3: res4a, res4b (16 channels, 240 frames per stream) and the output conv: lrelu, Conv1d(16 -> 1, k7), tanh.
__device__ __forceinline__ void t3_body(const StageArgs& a, const int g, float* __restrict__ lds) {
  constexpr int C = 16, T = 240, S = kT3Streams;
  float* X = lds;
  float* Y = X + buf_floats<C, T, S>();
  float* HIN = Y + buf_floats<C, T, S>();   // [S][2][C]
  float* HC = HIN + S * 2 * C;              // [S][6][C]: history of the output conv's input
  float* BIAS = HC + S * 6 * C;             // C | C
  float* FW = BIAS + 2 * C;
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
  float4 bfa[1][3], bfb[1][3];
  fetch_b<48, 16>(a.w[0], bfa, wave, lane);
  fetch_b<48, 16>(a.w[1], bfb, wave, lane);
  const float fin_b = a.fin_b[0];
  const float fw = tid < 7 * 16 ? a.fin_w[tid] : 0.0f;
  prologue<C, T, S, TS_YA4, TS_YB4, TS_YC4, 6, false, C, C, 0>(a, hop, b0, X, Y, HIN, HC, BIAS, tid);
  if (tid < 7 * 16) FW[tid] = fw;
  __syncthreads();
  layer<C, C, 3, 1, T, S>(X, bfa, n_rows, wave, lane, [&](int s, int t, int n, float v) {
    const int o = row_off<C, T>(s, t) + n;
    Y[o] = X[o] + (v + BIAS[n]);
  });
  __syncthreads();
  stash_to_rows<C, T, S, 6>(X, HC, -6, tid);
  layer<C, C, 3, 3, T, S>(Y, bfb, n_rows, wave, lane, [&](int s, int t, int n, float v) {
    const int o = row_off<C, T>(s, t) + n;
    X[o] = Y[o] + (v + BIAS[C + n]);
  });
  __syncthreads();
  // ---- output conv: one thread per sample, the operations of wave_tail.hip.h in the same order
  if (tid < S * T) {
    const int s = tid / T, t = tid % T;
    if (b0 + s < a.B) {
      float acc = 0.0f;
      const float* x = X + row_off<C, T>(s, t - 6);
#pragma unroll
      for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc = bsp::fma(bsp::lrelu(x[j * cs<C>() + c]), FW[j * 16 + c], acc);
      d_out[(size_t)(b0 + s) * B_OUT_HOP + t] = bsp::tanh(acc + fin_b);
    }
  }
  state_store_rows<C, T, S, 6>(a.state, TS_YB4, Y, T - 6, b0, a.B, tid);
  state_store_rows<C, T, S, 6>(a.state, TS_YC4, X, T - 6, b0, a.B, tid);
  for (int e = tid; e < S * 2 * C / 4; e += NTHR) {
    const int s = e / (2 * C / 4), q = e % (2 * C / 4);
    if (b0 + s < a.B) *reinterpret_cast<float4*>(a.state + (size_t)(b0 + s) * TAIL_STATE_FLOATS + TS_YA4 + 4 * q) = *reinterpret_cast<const float4*>(HIN + 4 * e);
  }
}
struct T3Op {
  using Args = StageArgs;
  static constexpr int NTHR = tst::NTHR, LDS_FLOATS = kT3Lds;
  static inline dim3 grid(const Args& a) { return dim3((a.B + kT3Streams - 1) / kT3Streams, 1); }
  static inline bhip::LaunchInfo info(const Args& a) {
    const double macs = 2.0 * 240 * 48 * 16 + 240.0 * 112;
    return bhip::LaunchInfo{"wave.tail3", 2.0 * a.B * macs, 4.0 * (2.0 * 48 * 16 + 112.0 + a.B * (240.0 * 16 + 240 + 2 * 14 * 16))};
  }
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) { t3_body(a, bx, lds); }
};

}  // namespace tst
