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
// Layout: one 512-thread workgroup per stream.  Three LDS activation buffers rotate through the
// layers, rows = [history | new frames], row stride C+2 floats (bank = 2*row + k: conflict-free A
// operand reads).  Each layer's weights are staged whole in LDS; the NEXT layer's weights are
// fetched into registers while the current layer computes (L2 latency off the critical path).
// Cross-hop history (the last 1..6 frames of every intermediate) lives in a 960-float state block
// per stream in HBM.  Numerics: every K here is <= 256, so each output is one k-ascending MFMA chain
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
  Ring in;       // output of up2: C = 64, n = 20, history 2
  float* state;  // [B][TAIL_STATE_FLOATS]
  const float *w[8], *b[8];  // res2a, res2b, up3, res3a, res3b, up4, res4a, res4b
  const float *fin_w, *fin_b;
  float* d_out;  // [B][240]
  const int* hop;
};

namespace tail {
constexpr int NTHR = 512, NWAVE = 8;
constexpr int BUF_FLOATS = (240 + 6) * 18;  // largest activation buffer (stage 4)
constexpr int W_FLOATS = 128 * 128;         // largest weight matrix (up3)

// One layer as a small GEMM: rows = frames, K = KSZ*CIN, N = NOUT, operands from LDS.
//   conv  (UPR == 0): out[H_OUT + t][n]            = in[H_IN + t][n] + (acc + bias[n])       (residual)
//   convT (UPR  > 0): out[H_OUT + t*UPR + n/COUT][n%COUT] = acc + bias[n],  COUT = NOUT / UPR
template <int CIN, int NOUT, int KSZ, int DIL, int T, int H_IN, int H_OUT, int UPR>
__device__ __forceinline__ void layer(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ wl,
                                      const float* __restrict__ bias, int wave, int lane) {
  constexpr int SI = CIN + 2;
  constexpr int COUT = UPR > 0 ? NOUT / UPR : NOUT;
  constexpr int SO = COUT + 2;
  constexpr int MT = (T + 15) / 16, NTL = NOUT / 16, TILES = MT * NTL;
  const int i = lane & 15, kq = lane >> 4;
  for (int tile = wave; tile < TILES; tile += NWAVE) {
    const int mt = tile / NTL, nt = tile % NTL;
    int row = mt * 16 + i;
    if (row > T - 1) row = T - 1;  // padded rows recompute the last frame; never stored
    tail_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KSZ; ++j) {
      const float* arow = in + (H_IN + row - (KSZ - 1 - j) * DIL) * SI;
      const float* wrow = wl + (size_t)(j * CIN) * NOUT + nt * 16 + i;
#pragma unroll 4
      for (int c = 0; c < CIN; c += 4) {
        const float a = bsp::lrelu(arow[c + kq]);
        const float b = wrow[(size_t)(c + kq) * NOUT];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
    }
    const int n = nt * 16 + i;
    const float bn = bias[n];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = mt * 16 + kq * 4 + e;
      if (t < T) {
        float v = acc[e] + bn;
        if constexpr (UPR == 0) {
          v = in[(H_IN + t) * SI + n] + v;
          out[(H_OUT + t) * SO + n] = v;
        } else {
          out[(H_OUT + t * UPR + n / COUT) * SO + (n % COUT)] = v;
        }
      }
    }
  }
}

// Weight prefetch: a [K][N] matrix (TOTAL4 float4s) is fetched into named float4 registers (no
// arrays: hipcc keeps indexed private arrays in scratch/LDS here) and dropped into LDS one phase later.
#define TAIL_F1(R, S, GPTR, TOTAL4)                                                              \
  {                                                                                              \
    const int idx_ = tid + (S) * tail::NTHR;                                                     \
    R = reinterpret_cast<const float4*>(GPTR)[idx_ < (TOTAL4) ? idx_ : 0];                       \
  }
#define TAIL_S1(R, S, LPTR, TOTAL4)                                                              \
  {                                                                                              \
    const int idx_ = tid + (S) * tail::NTHR;                                                     \
    if (idx_ < (TOTAL4)) reinterpret_cast<float4*>(LPTR)[idx_] = R;                              \
  }

// history rows: state block <-> LDS buffer rows
template <int C, int ROWS>
__device__ __forceinline__ void hist_load(float* __restrict__ buf, const float* __restrict__ st, int tid) {
  for (int e = tid; e < ROWS * C; e += NTHR) buf[(e / C) * (C + 2) + (e % C)] = st[e];
}
template <int C, int ROWS>
__device__ __forceinline__ void hist_save(float* __restrict__ st, const float* __restrict__ buf, int first_row, int tid) {
  for (int e = tid; e < ROWS * C; e += NTHR) st[e] = buf[(first_row + e / C) * (C + 2) + (e % C)];
}
}  // namespace tail

static __global__ __launch_bounds__(tail::NTHR) void wave_tail_kernel(const TailArgs a) {
  using namespace tail;
  __shared__ __attribute__((aligned(16))) float lds[3 * BUF_FLOATS + W_FLOATS + 7 * 16 + 16];
  float* R0 = lds;
  float* R1 = lds + BUF_FLOATS;
  float* R2 = lds + 2 * BUF_FLOATS;
  float* W = lds + 3 * BUF_FLOATS;
  float* FW = W + W_FLOATS;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hop = *a.hop;
  float* st = a.state + (size_t)b * TAIL_STATE_FLOATS;

  float4 wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7;  // next layer's weights in flight (<= 128*128 floats / 512 threads)

  // ---- prologue: weights of res2a, input frames (history 2 + 20 new, 64 ch) and output histories
  TAIL_F1(wr0, 0, a.w[0], 3072) TAIL_F1(wr1, 1, a.w[0], 3072) TAIL_F1(wr2, 2, a.w[0], 3072) TAIL_F1(wr3, 3, a.w[0], 3072) TAIL_F1(wr4, 4, a.w[0], 3072) TAIL_F1(wr5, 5, a.w[0], 3072)
  {
    const int pos = ring_pos(a.in, hop);
    for (int e = tid; e < 22 * 16; e += NTHR) {
      const int fr = e >> 4, q = e & 15;
      const float4 v = *reinterpret_cast<const float4*>(ring_frame(a.in, b, pos, fr - 2) + 4 * q);
      float* d = R0 + fr * 66 + 4 * q;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    hist_load<64, 6>(R1, st + TS_YB2, tid);
    if (tid < 7 * 16) FW[tid] = a.fin_w[tid];
  }
  TAIL_S1(wr0, 0, W, 3072) TAIL_S1(wr1, 1, W, 3072) TAIL_S1(wr2, 2, W, 3072) TAIL_S1(wr3, 3, W, 3072) TAIL_S1(wr4, 4, W, 3072) TAIL_S1(wr5, 5, W, 3072)
  __syncthreads();

  // ---- res2a: R0 (H 2) -> R1 (H 6)
  TAIL_F1(wr0, 0, a.w[1], 3072) TAIL_F1(wr1, 1, a.w[1], 3072) TAIL_F1(wr2, 2, a.w[1], 3072) TAIL_F1(wr3, 3, a.w[1], 3072) TAIL_F1(wr4, 4, a.w[1], 3072) TAIL_F1(wr5, 5, a.w[1], 3072)
  layer<64, 64, 3, 1, 20, 2, 6, 0>(R0, R1, W, a.b[0], wave, lane);
  hist_load<64, 1>(R2, st + TS_YC2, tid);
  __syncthreads();
  TAIL_S1(wr0, 0, W, 3072) TAIL_S1(wr1, 1, W, 3072) TAIL_S1(wr2, 2, W, 3072) TAIL_S1(wr3, 3, W, 3072) TAIL_S1(wr4, 4, W, 3072) TAIL_S1(wr5, 5, W, 3072)
  hist_save<64, 6>(st + TS_YB2, R1, 20, tid);
  __syncthreads();

  // ---- res2b (dil 3): R1 (H 6) -> R2 (H 1)
  TAIL_F1(wr0, 0, a.w[2], 4096) TAIL_F1(wr1, 1, a.w[2], 4096) TAIL_F1(wr2, 2, a.w[2], 4096) TAIL_F1(wr3, 3, a.w[2], 4096) TAIL_F1(wr4, 4, a.w[2], 4096) TAIL_F1(wr5, 5, a.w[2], 4096) TAIL_F1(wr6, 6, a.w[2], 4096) TAIL_F1(wr7, 7, a.w[2], 4096)
  layer<64, 64, 3, 3, 20, 6, 1, 0>(R1, R2, W, a.b[1], wave, lane);
  hist_load<32, 2>(R0, st + TS_YA3, tid);
  __syncthreads();
  TAIL_S1(wr0, 0, W, 4096) TAIL_S1(wr1, 1, W, 4096) TAIL_S1(wr2, 2, W, 4096) TAIL_S1(wr3, 3, W, 4096) TAIL_S1(wr4, 4, W, 4096) TAIL_S1(wr5, 5, W, 4096) TAIL_S1(wr6, 6, W, 4096) TAIL_S1(wr7, 7, W, 4096)
  hist_save<64, 1>(st + TS_YC2, R2, 20, tid);
  __syncthreads();

  // ---- up3 (x4): R2 (H 1, 20 frames of 64) -> R0 (H 2, 80 frames of 32)
  TAIL_F1(wr0, 0, a.w[3], 768) TAIL_F1(wr1, 1, a.w[3], 768)
  layer<64, 128, 2, 1, 20, 1, 2, 4>(R2, R0, W, a.b[2], wave, lane);
  hist_load<32, 6>(R1, st + TS_YB3, tid);
  __syncthreads();
  TAIL_S1(wr0, 0, W, 768) TAIL_S1(wr1, 1, W, 768)
  hist_save<32, 2>(st + TS_YA3, R0, 80, tid);
  __syncthreads();

  // ---- res3a: R0 (H 2) -> R1 (H 6)
  TAIL_F1(wr0, 0, a.w[4], 768) TAIL_F1(wr1, 1, a.w[4], 768)
  layer<32, 32, 3, 1, 80, 2, 6, 0>(R0, R1, W, a.b[3], wave, lane);
  hist_load<32, 1>(R2, st + TS_YC3, tid);
  __syncthreads();
  TAIL_S1(wr0, 0, W, 768) TAIL_S1(wr1, 1, W, 768)
  hist_save<32, 6>(st + TS_YB3, R1, 80, tid);
  __syncthreads();

  // ---- res3b (dil 3): R1 (H 6) -> R2 (H 1)
  TAIL_F1(wr0, 0, a.w[5], 768) TAIL_F1(wr1, 1, a.w[5], 768)
  layer<32, 32, 3, 3, 80, 6, 1, 0>(R1, R2, W, a.b[4], wave, lane);
  hist_load<16, 2>(R0, st + TS_YA4, tid);
  __syncthreads();
  TAIL_S1(wr0, 0, W, 768) TAIL_S1(wr1, 1, W, 768)
  hist_save<32, 1>(st + TS_YC3, R2, 80, tid);
  __syncthreads();

  // ---- up4 (x3): R2 (H 1, 80 frames of 32) -> R0 (H 2, 240 frames of 16)
  TAIL_F1(wr0, 0, a.w[6], 192)
  layer<32, 48, 2, 1, 80, 1, 2, 3>(R2, R0, W, a.b[5], wave, lane);
  hist_load<16, 6>(R1, st + TS_YB4, tid);
  __syncthreads();
  TAIL_S1(wr0, 0, W, 192)
  hist_save<16, 2>(st + TS_YA4, R0, 240, tid);
  __syncthreads();

  // ---- res4a: R0 (H 2) -> R1 (H 6)
  TAIL_F1(wr0, 0, a.w[7], 192)
  layer<16, 16, 3, 1, 240, 2, 6, 0>(R0, R1, W, a.b[6], wave, lane);
  hist_load<16, 6>(R2, st + TS_YC4, tid);
  __syncthreads();
  TAIL_S1(wr0, 0, W, 192)
  hist_save<16, 6>(st + TS_YB4, R1, 240, tid);
  __syncthreads();

  // ---- res4b (dil 3): R1 (H 6) -> R2 (H 6)
  layer<16, 16, 3, 3, 240, 6, 6, 0>(R1, R2, W, a.b[7], wave, lane);
  __syncthreads();
  hist_save<16, 6>(st + TS_YC4, R2, 240, tid);

  // ---- output conv: lrelu, Conv1d(16 -> 1, k7), tanh; one thread per sample, coalesced store
  if (tid < B_OUT_HOP) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
      for (int c = 0; c < 16; ++c) acc = bsp::fma(bsp::lrelu(R2[(tid + j) * 18 + c]), FW[j * 16 + c], acc);
    a.d_out[(size_t)b * B_OUT_HOP + tid] = bsp::tanh(acc + a.fin_b[0]);
  }
}
