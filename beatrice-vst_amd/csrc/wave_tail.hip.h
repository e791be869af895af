// wave_tail.hip.h -- the upsampler tail of the waveform generator as ONE kernel per hop
// (MODEL_SPEC 4.4.3, from the first residual conv of stage 2 to the output samples):
//
//   res2a, res2b (64 ch, 20 frames) -> up3 -> res3a, res3b (32 ch, 80 frames)
//   -> up4 -> res4a, res4b (16 ch, 240 frames) -> lrelu, Conv1d(16->1, k7), tanh -> 240 samples
//
// Why fused: these nine layers are time-local per stream (row t of a layer needs rows t-6..t of the
// previous one, all of the same stream), their weights total 208 KB, and per-hop activations of one
// stream fit in LDS.  As nine launches they cost ~80 us per hop at B = 256 (launch + fill/drain per
// layer); as one workgroup per stream they are bound by the CU's own MFMA pipe (~2400 16x16x4 MFMAs).
//
// Layout: one 512-thread workgroup per stream, 78 KB of LDS (two workgroups per CU).  Three LDS
// activation buffers rotate through the layers, rows = [history | new frames], row stride C+2 floats
// (bank = 2*row + k: conflict-free A operand reads).  Weights never touch LDS: every wavefront holds
// the pre-packed B fragments of its column tile(s) in registers, fetched one layer ahead (L2 latency
// off the critical path), and keeps up to 3 independent accumulators in flight.  Cross-hop history
// (the last 1..6 frames of every intermediate, 960 floats per stream) is read into LDS once at the
// start and written back once at the end.  Numerics: every K here is <= 256, so each output is one k-ascending MFMA chain
// (MODEL_SPEC 2.2), residual added after bias -- identical bits to the layer-by-layer path.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_misc.hip.h"
#include "ring.h"
#include "spec_math.hip.h"

typedef float tail_f32x4 __attribute__((ext_vector_type(4)));

// per-stream state block (floats): histories of the intermediates, in consumption order
enum {
  TS_YB2 = 0,              // 6 x 64   input history of res2b (dil 3)
  TS_YC2 = TS_YB2 + 384,   // 1 x 64   input history of up3
  TS_YA3 = TS_YC2 + 64,    // 2 x 32   input history of res3a
  TS_YB3 = TS_YA3 + 64,    // 6 x 32
  TS_YC3 = TS_YB3 + 192,   // 1 x 32
  TS_YA4 = TS_YC3 + 32,    // 2 x 16
  TS_YB4 = TS_YA4 + 32,    // 6 x 16
  TS_YC4 = TS_YB4 + 96,    // 6 x 16   input history of the output conv (k7)
  TAIL_STATE_FLOATS = TS_YC4 + 96
};

struct TailArgs {
  Ring in;       // output of up2: C = 64, n = 20*H, history 2
  float* state;  // [B][TAIL_STATE_FLOATS]
  const float *w[8], *b[8];  // res2a, res2b, up3, res3a, res3b, up4, res4a, res4b
  const float *fin_w, *fin_b;
  float* d_out;  // [B][H*240]
  const int* hop;
  size_t io_stride;   // 0, or floats between the slots of a resident multi-step output buffer (slot = hop[1])
  int* host_flag;     // 1-stream ABI or null: d_out is then a pinned host block, and this word behind it gets the call's sequence word (*seq) once the 240 samples
  const int* seq;     // are written (system-scope fence): the host polls it -- no copy command, no stream query (csrc/abi.hip flag_wait)
#ifdef TAIL_TIMING
  unsigned long long* stamps;  // tools/microbench/tail_timing.hip: [B][16] wall-clock stamps per phase
#endif
};
__device__ __forceinline__ void globalize(TailArgs& a) {
  globalize(a.in); a.state = as_global(a.state);
  for (int i = 0; i < 8; ++i) { a.w[i] = as_global(a.w[i]); a.b[i] = as_global(a.b[i]); }
  a.fin_w = as_global(a.fin_w); a.fin_b = as_global(a.fin_b); a.d_out = as_global(a.d_out); a.hop = as_global(a.hop);
}
#ifdef TAIL_TIMING
#define TAIL_STAMP(i) do { if (tid == 0) a.stamps[b * 16 + (i)] = wall_clock64(); } while (0)
// finer split, accumulated over the layers by wavefront 0 (tools/microbench/tail_timing.hip): k = 0 MFMA loop, 1 epilogue,
// 2 history rows in + barrier wait, 3 history rows out (shader cycles)
namespace tail { __shared__ unsigned long long sub_acc[4]; __shared__ unsigned long long sub_t; }
#define TAIL_SUB(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); tail::sub_acc[k] += now_ - tail::sub_t; tail::sub_t = now_; } } while (0)
#else
#define TAIL_STAMP(i) do { } while (0)
#define TAIL_SUB(k) do { } while (0)
#endif

namespace tail {
constexpr int NTHR = 512, NWAVE = 8;
constexpr int BUF_FLOATS = (80 + 6) * (2 * 32 + 2);  // largest activation buffer: stage 3 with activated copy (5676) > stage 4 raw (4428)
constexpr int BIAS_FLOATS = 64 + 64 + 128 + 32 + 32 + 48 + 16 + 16;
// bias offsets inside the LDS bias block, per layer
constexpr int BO[8] = {0, 64, 128, 256, 288, 320, 368, 384};

// Work split of one layer over the 8 wavefronts: NWN column groups x NWM row groups.  A wavefront owns
// CT column tiles (1, or all of them when the tile count is not a power of two) and up to RT row
// tiles, i.e. RT*CT independent accumulators sharing the B fragments it holds in registers.
template <int T, int NOUT>
struct Split {
  static constexpr int MT = (T + 15) / 16, NTL = NOUT / 16;
  static constexpr bool POW2 = (NTL & (NTL - 1)) == 0;
  static constexpr int NWN = POW2 ? (NTL < 8 ? NTL : 8) : 1;
  static constexpr int NWM = 8 / NWN;
  static constexpr int CT = POW2 ? 1 : NTL;
  static constexpr int RT = (MT + NWM - 1) / NWM;
  static_assert(!POW2 || NTL <= 8, "at most 8 column groups");
};

// B fragments of one layer for this wavefront: CT column tiles x K/16 packed records (float4 each),
// fetched from the pre-packed global weights (conv_gemm.hip.h) one layer ahead of their use.
template <int K, int NOUT, int T>
__device__ __forceinline__ void fetch_b(const float* __restrict__ wpacked, float4 (&bf)[Split<T, NOUT>::CT][K / 16], int wave, int lane) {
  using S = Split<T, NOUT>;
  const int wn = wave % S::NWN;
#pragma unroll
  for (int ct = 0; ct < S::CT; ++ct) {
    const int nt = S::POW2 ? wn : ct;
    const float4* p = reinterpret_cast<const float4*>(wpacked) + (size_t)nt * (K / 16) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < K / 16; ++kb) bf[ct][kb] = p[(size_t)kb * 64];
  }
}

// One layer as a small GEMM: rows = frames, K = KSZ*CIN, N = NOUT; A from LDS, B from registers.
//   conv  (UPR == 0): out[H_OUT + t][n]            = in[H_IN + t][n] + (acc + bias[n])       (residual)
//   convT (UPR  > 0): out[H_OUT + t*UPR + n/COUT][n%COUT] = acc + bias[n],  COUT = NOUT / UPR
// Buffers of the 64- and 32-channel stages keep an ACTIVATED copy next to the raw values
// (row = [raw C | lrelu(raw) C | 2 pad]): the MFMA loop then reads its A operand ready-made instead of
// spending three VALU instructions per step on lrelu (measured: 0.8 us of every 2.6 us phase); the raw
// half serves the residual add and the history.  The 16-channel stage stays raw-only (it would not fit
// two workgroups per CU otherwise) and applies lrelu in the loop as max(x, 0.1x).
__host__ __device__ constexpr int row_stride(int C) { return C >= 32 ? 2 * C + 2 : C + 2; }
__host__ __device__ constexpr bool has_act(int C) { return C >= 32; }

template <int CIN, int NOUT, int KSZ, int DIL, int T, int H_IN, int H_OUT, int UPR>
__device__ __forceinline__ void layer(const float* __restrict__ in, float* __restrict__ out,
                                      const float4 (&bf)[Split<T, NOUT>::CT][KSZ * CIN / 16],
                                      const float* __restrict__ bias, int wave, int lane) {
  using S = Split<T, NOUT>;
  constexpr int SI = row_stride(CIN);
  constexpr int COUT = UPR > 0 ? NOUT / UPR : NOUT;
  constexpr int SO = row_stride(COUT);
  constexpr int AOFF = has_act(CIN) ? CIN : 0;  // where the A operand lives inside an input row
  const int i = lane & 15, kq = lane >> 4;
  const int wn = wave % S::NWN, wm = wave / S::NWN;
  tail_f32x4 acc[S::RT][S::CT];
  int row[S::RT];
#pragma unroll
  for (int rt = 0; rt < S::RT; ++rt) {
    int r = (wm + rt * S::NWM) * 16 + i;
    row[rt] = r > T - 1 ? T - 1 : r;  // padded rows recompute the last frame; never stored
#pragma unroll
    for (int ct = 0; ct < S::CT; ++ct) acc[rt][ct] = tail_f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < KSZ; ++j) {
#pragma unroll
    for (int c = 0; c < CIN; c += 4) {
      constexpr int dummy = 0; (void)dummy;
      const int ks = (j * CIN + c) / 4;  // MFMA step index within the layer's reduction
      float av[S::RT];
#pragma unroll
      for (int rt = 0; rt < S::RT; ++rt) {
        const float x = in[(H_IN + row[rt] - (KSZ - 1 - j) * DIL) * SI + AOFF + c + kq];
        av[rt] = has_act(CIN) ? x : fmaxf(x, 0.1f * x);  // == lrelu(x) bit for bit
      }
#pragma unroll
      for (int ct = 0; ct < S::CT; ++ct) {
        const float4 f = bf[ct][ks >> 2];
        const float bv = (ks & 3) == 0 ? f.x : ((ks & 3) == 1 ? f.y : ((ks & 3) == 2 ? f.z : f.w));
#pragma unroll
        for (int rt = 0; rt < S::RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt], bv, acc[rt][ct], 0, 0, 0);
      }
    }
  }
  TAIL_SUB(0);
#pragma unroll
  for (int rt = 0; rt < S::RT; ++rt) {
    const int mt = wm + rt * S::NWM;
    if (mt >= S::MT) continue;
#pragma unroll
    for (int ct = 0; ct < S::CT; ++ct) {
      const int n = (S::POW2 ? wn : ct) * 16 + i;
      const float bn = bias[n];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t = mt * 16 + kq * 4 + e;
        if (t < T) {
          float v = acc[rt][ct][e] + bn;
          float* dst;
          if constexpr (UPR == 0) {
            v = in[(H_IN + t) * SI + n] + v;
            dst = out + (H_OUT + t) * SO + n;
          } else {
            dst = out + (H_OUT + t * UPR + n / COUT) * SO + (n % COUT);
          }
          dst[0] = v;
          if constexpr (has_act(COUT)) dst[COUT] = bsp::lrelu(v);
        }
      }
    }
  }
  TAIL_SUB(1);
}

// history rows: LDS state block <-> LDS activation buffer rows
template <int C, int ROWS>
__device__ __forceinline__ void hist_in(float* __restrict__ buf, const float* __restrict__ st, int tid) {
  for (int e = tid; e < ROWS * C; e += NTHR) {
    float* d = buf + (e / C) * row_stride(C) + (e % C);
    const float v = st[e];
    d[0] = v;
    if constexpr (has_act(C)) d[C] = bsp::lrelu(v);
  }
}
template <int C, int ROWS>
__device__ __forceinline__ void hist_out(float* __restrict__ st, const float* __restrict__ buf, int first_row, int tid) {
  for (int e = tid; e < ROWS * C; e += NTHR) st[e] = buf[(first_row + e / C) * row_stride(C) + (e % C)];
}
}  // namespace tail

// 2 workgroups per CU (59 KB of LDS each): while one waits at a barrier the other computes
constexpr int kTailLdsFloats = 3 * tail::BUF_FLOATS + 2 * TAIL_STATE_FLOATS + tail::BIAS_FLOATS + 7 * 16;
template <int H>  // hops per step, compile time: H = 1 keeps the single-hop kernel free of loop state
__device__ __forceinline__ void wave_tail_body(const TailArgs& a, const int b, float* __restrict__ lds) {
  using namespace tail;
  float* R0 = lds;
  float* R1 = lds + BUF_FLOATS;
  float* R2 = lds + 2 * BUF_FLOATS;
  float* SI_ = lds + 3 * BUF_FLOATS;         // state of the previous hop (read)
  float* SO_ = SI_ + TAIL_STATE_FLOATS;      // state after this hop (written, stored at the end)
  float* BIAS = SO_ + TAIL_STATE_FLOATS;
  float* FW = BIAS + BIAS_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* st = a.state + (size_t)b * TAIL_STATE_FLOATS;
  TAIL_STAMP(0);
#ifdef TAIL_SETPRIO
  __builtin_amdgcn_s_setprio(TAIL_SETPRIO);   // experiment: issue priority over the co-resident workgroup's wavefronts
#endif

  // biases, output-conv taps and the stream's state block: one round of global loads
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
  const int io = a.io_stride != 0 ? stepc::slot(a.hop) : 0;
  float* __restrict__ d_out = a.d_out + (size_t)io * a.io_stride;
  for (int e = tid; e < TAIL_STATE_FLOATS; e += NTHR) SI_[e] = st[e];
  if (tid < BIAS_FLOATS) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) l += tid >= BO[k] ? 1 : 0;
    BIAS[tid] = a.b[l][tid - BO[l]];
  }
  if (tid < 7 * 16) FW[tid] = a.fin_w[tid];

#pragma unroll 1
  for (int hh = 0; hh < H; ++hh) {
  // B fragments, one layer ahead (a.w[] are pre-packed)
  float4 b_r2a[1][12], b_r2b[1][12], b_u3[1][8], b_r3a[1][6], b_r3b[1][6], b_u4[3][4], b_r4a[1][3], b_r4b[1][3];
  fetch_b<192, 64, 20>(a.w[0], b_r2a, wave, lane);
  fetch_b<192, 64, 20>(a.w[1], b_r2b, wave, lane);

  // ---- input frames of this hop (history 2 + 20 new, 64 channels), raw + activated
  {
    const int pos = ring_pos(a.in, hop);
    for (int e = tid; e < 22 * 16; e += NTHR) {
      const int fr = e >> 4, q = e & 15;
      const float4 v = *reinterpret_cast<const float4*>(ring_frame(a.in, b, pos, hh * 20 + fr - 2) + 4 * q);
      float* d = R0 + fr * row_stride(64) + 4 * q;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      d[64] = bsp::lrelu(v.x); d[65] = bsp::lrelu(v.y); d[66] = bsp::lrelu(v.z); d[67] = bsp::lrelu(v.w);
    }
  }
  __syncthreads();
  TAIL_STAMP(1);
#ifdef TAIL_TIMING
  if (tid == 0) { for (int k = 0; k < 4; ++k) tail::sub_acc[k] = 0; tail::sub_t = __builtin_readcyclecounter(); }
#endif
  hist_in<64, 6>(R1, SI_ + TS_YB2, tid);
  // (R1's history rows are only read by res2b, after the next barrier)

  // ---- res2a: R0 (H 2) -> R1 (H 6)
  fetch_b<128, 128, 20>(a.w[2], b_u3, wave, lane);
  layer<64, 64, 3, 1, 20, 2, 6, 0>(R0, R1, b_r2a, BIAS + BO[0], wave, lane);
  hist_in<64, 1>(R2, SI_ + TS_YC2, tid);
  __syncthreads();
  TAIL_SUB(2);
  hist_out<64, 6>(SO_ + TS_YB2, R1, 20, tid);
  TAIL_SUB(3);
  TAIL_STAMP(2);

  // ---- res2b (dil 3): R1 (H 6) -> R2 (H 1)
  fetch_b<96, 32, 80>(a.w[3], b_r3a, wave, lane);
  layer<64, 64, 3, 3, 20, 6, 1, 0>(R1, R2, b_r2b, BIAS + BO[1], wave, lane);
  hist_in<32, 2>(R0, SI_ + TS_YA3, tid);
  __syncthreads();
  TAIL_SUB(2);
  hist_out<64, 1>(SO_ + TS_YC2, R2, 20, tid);
  TAIL_SUB(3);
  TAIL_STAMP(3);

  // ---- up3 (x4): R2 (H 1, 20 frames of 64) -> R0 (H 2, 80 frames of 32)
  fetch_b<96, 32, 80>(a.w[4], b_r3b, wave, lane);
  layer<64, 128, 2, 1, 20, 1, 2, 4>(R2, R0, b_u3, BIAS + BO[2], wave, lane);
  hist_in<32, 6>(R1, SI_ + TS_YB3, tid);
  __syncthreads();
  TAIL_SUB(2);
  hist_out<32, 2>(SO_ + TS_YA3, R0, 80, tid);
  TAIL_SUB(3);
  TAIL_STAMP(4);

  // ---- res3a: R0 (H 2) -> R1 (H 6)
  fetch_b<64, 48, 80>(a.w[5], b_u4, wave, lane);
  layer<32, 32, 3, 1, 80, 2, 6, 0>(R0, R1, b_r3a, BIAS + BO[3], wave, lane);
  hist_in<32, 1>(R2, SI_ + TS_YC3, tid);
  __syncthreads();
  TAIL_SUB(2);
  hist_out<32, 6>(SO_ + TS_YB3, R1, 80, tid);
  TAIL_SUB(3);
  TAIL_STAMP(5);

  // ---- res3b (dil 3): R1 (H 6) -> R2 (H 1)
  fetch_b<48, 16, 240>(a.w[6], b_r4a, wave, lane);
  layer<32, 32, 3, 3, 80, 6, 1, 0>(R1, R2, b_r3b, BIAS + BO[4], wave, lane);
  hist_in<16, 2>(R0, SI_ + TS_YA4, tid);
  __syncthreads();
  TAIL_SUB(2);
  hist_out<32, 1>(SO_ + TS_YC3, R2, 80, tid);
  TAIL_SUB(3);
  TAIL_STAMP(6);

  // ---- up4 (x3): R2 (H 1, 80 frames of 32) -> R0 (H 2, 240 frames of 16)
  fetch_b<48, 16, 240>(a.w[7], b_r4b, wave, lane);
  layer<32, 48, 2, 1, 80, 1, 2, 3>(R2, R0, b_u4, BIAS + BO[5], wave, lane);
  hist_in<16, 6>(R1, SI_ + TS_YB4, tid);
  __syncthreads();
  TAIL_SUB(2);
  hist_out<16, 2>(SO_ + TS_YA4, R0, 240, tid);
  TAIL_SUB(3);
  TAIL_STAMP(7);

  // ---- res4a: R0 (H 2) -> R1 (H 6)
  layer<16, 16, 3, 1, 240, 2, 6, 0>(R0, R1, b_r4a, BIAS + BO[6], wave, lane);
  hist_in<16, 6>(R2, SI_ + TS_YC4, tid);
  __syncthreads();
  TAIL_SUB(2);
  hist_out<16, 6>(SO_ + TS_YB4, R1, 240, tid);
  TAIL_SUB(3);
  TAIL_STAMP(8);

  // ---- res4b (dil 3): R1 (H 6) -> R2 (H 6)
  layer<16, 16, 3, 3, 240, 6, 6, 0>(R1, R2, b_r4b, BIAS + BO[7], wave, lane);
  __syncthreads();
  TAIL_SUB(2);
  hist_out<16, 6>(SO_ + TS_YC4, R2, 240, tid);
  TAIL_SUB(3);
  TAIL_STAMP(9);

  // ---- output conv: lrelu, Conv1d(16 -> 1, k7), tanh; one thread per sample, coalesced store
  if (tid < B_OUT_HOP) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
      for (int c = 0; c < 16; ++c) acc = bsp::fma(bsp::lrelu(R2[(tid + j) * 18 + c]), FW[j * 16 + c], acc);
    d_out[((size_t)b * H + hh) * B_OUT_HOP + tid] = bsp::tanh(acc + a.fin_b[0]);
    if (a.host_flag != nullptr) __threadfence_system();
  }
  __syncthreads();
  if (a.host_flag != nullptr && tid == 0) __hip_atomic_store(a.host_flag, *a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // (the state below is the next call's business)
  { float* tmp = SI_; SI_ = SO_; SO_ = tmp; }  // this hop's histories are the next hop's state
  }  // hops of the step
  for (int e = tid; e < TAIL_STATE_FLOATS; e += NTHR) st[e] = SI_[e];
  TAIL_STAMP(10);
#ifdef TAIL_TIMING
  if (tid == 0) for (int k = 0; k < 4; ++k) a.stamps[b * 16 + 11 + k] = tail::sub_acc[k];
#endif
}

template <int H>
static __global__ __launch_bounds__(tail::NTHR, 3) void wave_tail_kernel(const TailArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kTailLdsFloats];
  wave_tail_body<H>(a, blockIdx.x, lds);
}
template <int H>
struct TailOp {
  using Args = TailArgs;
  static constexpr int NTHR = tail::NTHR;
  static constexpr int LDS_FLOATS = kTailLdsFloats;
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) { wave_tail_body<H>(a, bx, lds); }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int, float* lds) { wave_tail_body<H>(a, bx, lds); }   // (not a stage of ragged ticks: the tail runs as three stages there)
};
