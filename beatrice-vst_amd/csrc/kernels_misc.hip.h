// kernels_misc.hip.h -- the non-GEMM kernels of the per-hop path (MODEL_SPEC section 4) and the
// set-time kernels of the embedding setter.  All of them are HBM/LDS-bound elementwise, gather or
// reduction work: one workgroup (or one wavefront) per stream, coalesced channel-last accesses,
// cross-lane reductions through wavefront shuffles in the order the spec fixes.
#pragma once
#include <hip/hip_runtime.h>

#include "ring.h"
#include "spec_math.hip.h"

#define B_IN_HOP 160
#define B_OUT_HOP 240
#define B_HID 256
#define B_PHONE_CH 128
#define B_PITCH_BINS 448
#define B_CODEBOOK 512
#define B_KV_LEN 384
#define B_KV_CH 128
#define B_NBLOCKS 4
#define B_FFT_N 1024
#define B_SPEC_BINS 512
#define B_PITCH_HIST (B_FFT_N - B_IN_HOP)

// ---------------------------------------------------------------------------------------------
// The step counter only ever selects ring slots (counter mod m, m <= 17 slots): it wraps at lcm(1..17)
// so that a stream can run forever without the modulus glitching at an integer overflow.
#define B_HOP_WRAP 12252240
__host__ __device__ inline int hop_next(int hop) { return hop + 1 >= B_HOP_WRAP ? 0 : hop + 1; }
static __global__ void hop_advance_kernel(int* hop) { *hop = hop_next(*hop); }

// ---------------------------------------------------------------------------------------------
// Phone front-end layer 1 (MODEL_SPEC 4.1.1): Conv1d(1 -> 64, k=10, stride=5) + GELU.
// K is only 10, so this is VALU work: one workgroup per stream, thread = (frame t, 8 channels).
// Also appends the hop's 160 samples to the audio ring (5 samples of history are re-read).
struct F1Args {
  const float* d_in;   // [B][H*160]
  Ring audio, out;
  const float *w, *bias;
  const int* hop;      // step counter this kernel reads ([1] = resident-I/O slot, read only if io_stride != 0)
  int* hop_publish;    // optional: workgroup (0,0) copies the counter (and the slot) here for the rest of the chain
  int* hop_publish_wave;  // optional, int[4][2]: the same pair again at [counter & 3], for the waveform generator's stages,
                          // which may still be working on earlier steps when the next one's front end starts
  int H;
  size_t io_stride;    // 0, or floats between the slots of a resident multi-step input buffer (batch.hip)
};
__device__ __forceinline__ void globalize(F1Args& a) {
  a.d_in = as_global(a.d_in); globalize(a.audio); globalize(a.out); a.w = as_global(a.w); a.bias = as_global(a.bias);
  a.hop = as_global(a.hop); a.hop_publish = as_global(a.hop_publish); a.hop_publish_wave = as_global(a.hop_publish_wave);
}
// grid (stream, hop-in-step).  Samples before the step come from the audio ring, the rest from d_in.
constexpr int kF1LdsFloats = 168 + 10 * 64;
// PACK streams per workgroup, 256 threads each (the tick launch runs 512-thread workgroups: two streams share one).
// n_streams bounds the stream index when PACK > 1 (every thread still reaches the barrier).
template <int PACK, bool RAG = false>
__device__ __forceinline__ void phone_f1_body_t(const F1Args& a, const int bx, const int hh, float* __restrict__ lds, const int n_streams) {
  const int tid = threadIdx.x & 255, part = PACK > 1 ? (int)(threadIdx.x >> 8) : 0;
  float* x = lds + 168 * part;     // [5 + 160] per stream
  float* ws = lds + 168 * PACK;    // [10][64], shared
  const int b = bx * PACK + part;
  const bool in_batch = PACK == 1 || b < n_streams;
  const int step = stepc::step(a.hop), H = a.H;
  if (step < 0) return;
  const int hop = in_batch ? stepc::of_t<RAG>(step, b) : -1;   // (a stream that sits the step out: -1)
  const bool live = hop >= 0;
  const int io = a.io_stride != 0 ? stepc::slot(a.hop) : 0;
  if (a.hop_publish != nullptr && b == 0 && hh == 0 && tid == 0) {
    a.hop_publish[0] = hop; a.hop_publish[1] = io;
    if (a.hop_publish_wave != nullptr) { a.hop_publish_wave[(hop & 3) * 2] = hop; a.hop_publish_wave[(hop & 3) * 2 + 1] = io; }
  }
  const Ring& audio = a.audio;
  const Ring& out = a.out;
  const float* __restrict__ d_in = a.d_in + (size_t)io * a.io_stride;
  const float* __restrict__ w = a.w;
  const float* __restrict__ bias = a.bias;
  const int pos = live ? ring_pos(audio, hop) : 0;
  const float* src = d_in + (size_t)b * H * B_IN_HOP;
  for (int i = threadIdx.x; i < 10 * 64; i += 256 * PACK) ws[i] = w[i];
  if (live && tid < 5) {
    const int i = hh * B_IN_HOP + tid - 5;
    x[tid] = i < 0 ? *ring_frame(audio, b, pos, i) : src[i];
  }
  if (live && tid < B_IN_HOP) {
    const float v = src[hh * B_IN_HOP + tid];
    x[5 + tid] = v;
    *ring_frame(audio, b, pos, hh * B_IN_HOP + tid) = v;
  }
  __syncthreads();
  if (!live) return;
  const int t = tid >> 3, n0 = (tid & 7) * 8;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const float a = x[5 * t + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = bsp::fma(a, ws[j * 64 + n0 + u], acc[u]);
  }
  float* o = ring_frame(out, b, ring_pos(out, hop), hh * 32 + t) + n0;
#pragma unroll
  for (int u = 0; u < 8; u += 2) {   // pairs through the packed gelu (spec_math.hip.h): the same bits, fewer instructions
    const bsp::f32x2 g = bsp::gelu2(bsp::f32x2{acc[u] + bias[n0 + u], acc[u + 1] + bias[n0 + u + 1]});
    o[u] = g.x; o[u + 1] = g.y;
  }
}
__device__ __forceinline__ void phone_f1_body(const F1Args& a, const int b, const int hh, float* __restrict__ lds) {
  phone_f1_body_t<1>(a, b, hh, lds, 0);
}
static __global__ __launch_bounds__(256) void phone_f1_kernel(const F1Args a) {
  __shared__ __attribute__((aligned(16))) float lds[kF1LdsFloats];
  phone_f1_body(a, blockIdx.x, blockIdx.y, lds);
}
struct F1Op {
  using Args = F1Args;
  static constexpr int NTHR = 256;
  static constexpr int LDS_FLOATS = kF1LdsFloats;
  __device__ static __forceinline__ void run(const Args& a, int bx, int by, float* lds) { phone_f1_body(a, bx, by, lds); }
};
// two streams per 512-thread workgroup (H = 1): grid ((n_streams + 1) / 2, 1)
struct F1Args2 { F1Args a; int n_streams; };
__device__ __forceinline__ void globalize(F1Args2& a) { globalize(a.a); }
struct F1Op2 {
  using Args = F1Args2;
  static constexpr int NTHR = 512;
  static constexpr int LDS_FLOATS = 2 * 168 + 10 * 64;
  // (grid y = hop within the step: ((n_streams + 1) / 2, H))
  __device__ static __forceinline__ void run(const Args& a, int bx, int by, float* lds) { phone_f1_body_t<2>(a.a, bx, by, lds, a.n_streams); }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int by, float* lds) { phone_f1_body_t<2, RAG>(a.a, bx, by, lds, a.n_streams); }
};

// ---------------------------------------------------------------------------------------------
// k-nearest-neighbour codebook lookup (MODEL_SPEC 4.1.3).  One workgroup of 512 threads per
// stream; thread j owns codebook row j.  cbT is the codebook transposed to [128][512] so the
// distance loop reads coalesced rows.  Streams with k == 0 pass the raw vector through.
struct VqArgs {
  int H;                       // hops per step: rows are (stream, hop), codebook per stream
  Ring raw;                    // C = 128, n = H: the encoder's output vectors
  Ring out;                    // C = 128, n = H frames per step, m = 1 or 2 step slots
  const int* hop;
  const float* const* cbT;     // per row (stream, hop): [128][512]
  const float* const* cnorm;   // per row: [512]
  const int* k;                // per stream
};
__device__ __forceinline__ void globalize(VqArgs& a) {
  globalize(a.raw); globalize(a.out); a.hop = as_global(a.hop); a.cbT = as_global(a.cbT); a.cnorm = as_global(a.cnorm); a.k = as_global(a.k);
}
constexpr int kVqLdsFloats = B_PHONE_CH + 8 + 8 + 8;
template <bool RAG = false>
__device__ __forceinline__ void phone_vq_body(const VqArgs& a, const int row, float* __restrict__ lds) {
  float* x = lds;                                               // [128]
  float* red_d = lds + B_PHONE_CH;                              // [8]
  int* red_j = reinterpret_cast<int*>(lds + B_PHONE_CH + 8);    // [8]
  int& winner = *reinterpret_cast<int*>(lds + B_PHONE_CH + 16);
  const int b = row / a.H, j = threadIdx.x, lane = j & 63, wave = j >> 6;
  const int step = stepc::step(a.hop);
  if (step < 0) return;
  const int hop = stepc::of_t<RAG>(step, b);
  if (hop < 0) return;   // (the stream sits this step out)
  float* out = ring_frame(a.out, b, ring_pos(a.out, hop), row % a.H);
  const int k = a.k[b];
  const float* cbT = as_global_v(a.cbT[row]);   // (device allocations: global memory)
  if (j < B_PHONE_CH) x[j] = ring_frame(a.raw, b, ring_pos(a.raw, hop), row % a.H)[j];
  if (k <= 0 || cbT == nullptr) {
    if (j < B_PHONE_CH) out[j] = x[j];
    return;
  }
  __syncthreads();
  float dot = 0.0f;
#pragma unroll 8
  for (int c = 0; c < B_PHONE_CH; ++c) dot = bsp::fma(x[c], cbT[c * B_CODEBOOK + j], dot);
  float d = bsp::fma(-2.0f, dot, as_global_v(a.cnorm[row])[j]);
  float acc = 0.0f;
  for (int r = 0; r < k; ++r) {
    float bd = d;
    int bj = j;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float od = __shfl_xor(bd, off, 64);
      const int oj = __shfl_xor(bj, off, 64);
      if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
    }
    if (lane == 0) { red_d[wave] = bd; red_j[wave] = bj; }
    __syncthreads();
    if (j == 0) {
      float wd = red_d[0];
      int wj = red_j[0];
      for (int w = 1; w < 8; ++w)
        if (red_d[w] < wd || (red_d[w] == wd && red_j[w] < wj)) { wd = red_d[w]; wj = red_j[w]; }
      winner = wj;
    }
    __syncthreads();
    const int wj = winner;
    if (j == wj) d = __builtin_huge_valf();
    if (j < B_PHONE_CH) acc = acc + cbT[j * B_CODEBOOK + wj];
    __syncthreads();
  }
  if (j < B_PHONE_CH) out[j] = acc / (float)k;
}
static __global__ __launch_bounds__(512) void phone_vq_kernel(const VqArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kVqLdsFloats];
  phone_vq_body(a, blockIdx.x, lds);
}
// The same for the NH hops of a step in ONE workgroup (the tick launch with several hops per stage): a lookup is bound by the
// 256 KB of codebook a workgroup pulls through its CU's L1, and the hops of a stream almost always share their codebook -- then
// every codebook element is read once and feeds NH distances.  Per row the operations of phone_vq_body, in its order.
template <bool RAG, int NH>
__device__ __forceinline__ void phone_vq_rows_body(const VqArgs& a, const int b, float* __restrict__ lds) {
  float* x = lds;                                                    // [NH][128]
  float* red_d = lds + NH * B_PHONE_CH;                              // [8]
  int* red_j = reinterpret_cast<int*>(lds + NH * B_PHONE_CH + 8);    // [8]
  int& winner = *reinterpret_cast<int*>(lds + NH * B_PHONE_CH + 16);
  const int j = threadIdx.x, lane = j & 63, wave = j >> 6;
  const int step = stepc::step(a.hop);
  if (step < 0) return;
  const int hop = stepc::of_t<RAG>(step, b);
  if (hop < 0) return;
  const int k = a.k[b];
  const float* cb[NH];
  bool same = true;
#pragma unroll
  for (int h = 0; h < NH; ++h) { cb[h] = as_global_v(a.cbT[b * NH + h]); same = same && cb[h] == cb[0]; }
  if (j < B_PHONE_CH) {
#pragma unroll
    for (int h = 0; h < NH; ++h) x[h * B_PHONE_CH + j] = ring_frame(a.raw, b, ring_pos(a.raw, hop), h)[j];
  }
  __syncthreads();
  float dot[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) dot[h] = 0.0f;
  if (k > 0 && same && cb[0] != nullptr) {
#pragma unroll 16
    for (int c = 0; c < B_PHONE_CH; ++c) {
      const float v = cb[0][c * B_CODEBOOK + j];
#pragma unroll
      for (int h = 0; h < NH; ++h) dot[h] = bsp::fma(x[h * B_PHONE_CH + c], v, dot[h]);
    }
  } else if (k > 0) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
      if (cb[h] != nullptr) {
#pragma unroll 8
        for (int c = 0; c < B_PHONE_CH; ++c) dot[h] = bsp::fma(x[h * B_PHONE_CH + c], cb[h][c * B_CODEBOOK + j], dot[h]);
      }
  }
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    float* out = ring_frame(a.out, b, ring_pos(a.out, hop), h);
    if (k <= 0 || cb[h] == nullptr) {   // (uniform over the workgroup)
      if (j < B_PHONE_CH) out[j] = x[h * B_PHONE_CH + j];
      continue;
    }
    float d = bsp::fma(-2.0f, dot[h], as_global_v(a.cnorm[b * NH + h])[j]);
    float acc = 0.0f;
    for (int r = 0; r < k; ++r) {
      float bd = d;
      int bj = j;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const float od = __shfl_xor(bd, off, 64);
        const int oj = __shfl_xor(bj, off, 64);
        if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
      }
      if (lane == 0) { red_d[wave] = bd; red_j[wave] = bj; }
      __syncthreads();
      if (j == 0) {
        float wd = red_d[0];
        int wj = red_j[0];
        for (int w = 1; w < 8; ++w)
          if (red_d[w] < wd || (red_d[w] == wd && red_j[w] < wj)) { wd = red_d[w]; wj = red_j[w]; }
        winner = wj;
      }
      __syncthreads();
      const int wj = winner;
      if (j == wj) d = __builtin_huge_valf();
      if (j < B_PHONE_CH) acc = acc + cb[h][j * B_CODEBOOK + wj];
      __syncthreads();
    }
    if (j < B_PHONE_CH) out[j] = acc / (float)k;
  }
}
template <int NH>
struct VqRowsOp {   // grid (streams, 1)
  using Args = VqArgs;
  static constexpr int NTHR = 512;
  static constexpr int LDS_FLOATS = NH * B_PHONE_CH + 24;
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) { phone_vq_rows_body<false, NH>(a, bx, lds); }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int, float* lds) { phone_vq_rows_body<RAG, NH>(a, bx, lds); }
};
struct VqOp {
  using Args = VqArgs;
  static constexpr int NTHR = 512;
  static constexpr int LDS_FLOATS = kVqLdsFloats;
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) { phone_vq_body(a, bx, lds); }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int, float* lds) { phone_vq_body<RAG>(a, bx, lds); }
};

// ---------------------------------------------------------------------------------------------
// Pitch front-end (MODEL_SPEC 4.2.1): window, 1024-point radix-2 DIT FFT in LDS, log power.
// One workgroup of 256 threads per stream; each thread does 2 butterflies per stage.
struct FftArgs {
  const float* d_in;  // [B][H*160]
  Ring audio, spec;
  const float *window, *twiddle;
  const int* hop;
  int H;
  size_t io_stride;  // see F1Args
};
__device__ __forceinline__ void globalize(FftArgs& a) {
  a.d_in = as_global(a.d_in); globalize(a.audio); globalize(a.spec); a.window = as_global(a.window); a.twiddle = as_global(a.twiddle);
  a.hop = as_global(a.hop);
}
// Round 5: the same butterflies (MODEL_SPEC 4.2.1, operation for operation, the same twiddle table: bit-identical), scheduled as FIVE
// passes of two stages each instead of ten passes through LDS.  A thread owns the four elements {base + m h1} of a pass (h1 = 4^p)
// and runs the stage-h1 and the stage-2 h1 butterflies on them in registers.  Pass 0 takes its elements straight from the audio
// (element 4 t + m of the bit-reversed order is the sample at rev8(t) + 256 rev2(m)): the bit-reversal scatter into LDS -- whose 1 024
// stores landed on one bank in groups of 32, half of the tick launch's LDS bank conflicts, profiles/r05_tick_inst_by_body.txt -- is
// gone; pass 4 leaves bins t and t + 256 in registers for the log-power.  Four LDS exchanges and barriers instead of eleven; LDS
// indices are padded by 4 floats per 16 (fft_at): the strided accesses of passes 1 and 2 then spread over all banks.
constexpr int kFftArr = B_FFT_N + 4 * (B_FFT_N / 16);   // one padded array of 1 024 floats
constexpr int kFftLdsFloats = 2 * kFftArr + B_FFT_N;
__device__ __forceinline__ int fft_at(const int i) { return i + 4 * (i >> 4); }
__device__ __forceinline__ void fft_bfly(float& ar, float& ai, float& br, float& bi, const float wr, const float wi) {
  const float tr = bsp::fma(-wi, bi, wr * br);
  const float ti = bsp::fma(wi, br, wr * bi);
  const float xr = ar, xi = ai;
  ar = xr + tr; ai = xi + ti;
  br = xr - tr; bi = xi - ti;
}
// stages h1 and 2 h1 on the four elements i_m = base + m h1 (j1 = base & (h1 - 1)); tw = the table (cos, -sin)(2 pi k / 1024) in LDS
__device__ __forceinline__ void fft_two_stages(float (&xr)[4], float (&xi)[4], const float* __restrict__ tw, const int h1, const int low) {
  const int s1 = (B_FFT_N / 2) / h1;               // twiddle step of stage h1: N / (2 h1)
  const float w1r = tw[2 * (low * s1)], w1i = tw[2 * (low * s1) + 1];
  fft_bfly(xr[0], xi[0], xr[1], xi[1], w1r, w1i);
  fft_bfly(xr[2], xi[2], xr[3], xi[3], w1r, w1i);
  const int s2 = s1 >> 1;                          // stage 2 h1: j = low (elements 0, 2) and low + h1 (elements 1, 3)
  const float war = tw[2 * (low * s2)], wai = tw[2 * (low * s2) + 1];
  const float wbr = tw[2 * ((low + h1) * s2)], wbi = tw[2 * ((low + h1) * s2) + 1];
  fft_bfly(xr[0], xi[0], xr[2], xi[2], war, wai);
  fft_bfly(xr[1], xi[1], xr[3], xi[3], wbr, wbi);
}
// PACK streams per workgroup, 256 threads each (see phone_f1_body_t); the twiddle table is shared
template <int PACK, bool RAG = false>
__device__ __forceinline__ void pitch_fft_body_t(const FftArgs& a, const int bx, const int hh, float* __restrict__ lds, const int n_streams) {
  const int tid = threadIdx.x & 255, part = PACK > 1 ? (int)(threadIdx.x >> 8) : 0;
  float* re = lds + 2 * kFftArr * part;
  float* im = re + kFftArr;
  float* tw = lds + 2 * kFftArr * PACK;
  const int b = bx * PACK + part;
  const bool in_batch = PACK == 1 || b < n_streams;
  const int step = stepc::step(a.hop), H = a.H;
  if (step < 0) return;
  const int hop = in_batch ? stepc::of_t<RAG>(step, b) : -1;   // (a stream that sits the step out: -1)
  const bool live = hop >= 0;
  const Ring& audio = a.audio;
  const Ring& spec = a.spec;
  const float* __restrict__ d_in = a.d_in + (a.io_stride != 0 ? (size_t)stepc::slot(a.hop) * a.io_stride : 0);
  const float* __restrict__ window = a.window;
  const float* __restrict__ twiddle = a.twiddle;
  const int pos = live ? ring_pos(audio, hop) : 0;
  const float* src = d_in + (size_t)b * H * B_IN_HOP;
  for (int i = threadIdx.x; i < B_FFT_N; i += 256 * PACK) tw[i] = twiddle[i];
  float xr[4] = {0.f, 0.f, 0.f, 0.f}, xi[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    // the frame: 864 samples of history + this hop's 160; sample index relative to the start of the step: negative = the ring,
    // otherwise the step's input (earlier hops of the step included).  Element m of this thread = position r8 + 256 rev2(m).
    const int r8 = (int)(__brev((unsigned)tid) >> 24);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int i = r8 + 256 * ((m >> 1) | ((m & 1) << 1));
      const int si = hh * B_IN_HOP + i - B_PITCH_HIST;
      const float s = si < 0 ? *ring_frame(audio, b, pos, si) : src[si];
      xr[m] = s * window[i];
    }
    if (tid < B_IN_HOP) *ring_frame(audio, b, pos, hh * B_IN_HOP + tid) = src[hh * B_IN_HOP + tid];   // each hop block appends its own 160 samples
  }
  __syncthreads();   // (the twiddles are in LDS)
  if (live) {
    fft_two_stages(xr, xi, tw, 1, 0);
    // elements 4 t .. 4 t + 3: one 16-byte store per array (fft_at keeps groups of 16 together)
    *reinterpret_cast<float4*>(re + fft_at(4 * tid)) = make_float4(xr[0], xr[1], xr[2], xr[3]);
    *reinterpret_cast<float4*>(im + fft_at(4 * tid)) = make_float4(xi[0], xi[1], xi[2], xi[3]);
  }
#pragma unroll
  for (int p = 1; p < 5; ++p) {
    constexpr int kH1[5] = {1, 4, 16, 64, 256};
    const int h1 = kH1[p];
    __syncthreads();   // (the pass before has written; inside a pass a thread reads and writes its own four elements only)
    if (live) {
      const int low = tid & (h1 - 1), base = ((tid >> (2 * p)) << (2 * p + 2)) + low;
#pragma unroll
      for (int m = 0; m < 4; ++m) { xr[m] = re[fft_at(base + m * h1)]; xi[m] = im[fft_at(base + m * h1)]; }
      fft_two_stages(xr, xi, tw, h1, low);
      if (p < 4) {
#pragma unroll
        for (int m = 0; m < 4; ++m) { re[fft_at(base + m * h1)] = xr[m]; im[fft_at(base + m * h1)] = xi[m]; }
      }
    }
  }
  if (!live) return;
  // after the last pass this thread holds bins t + 256 m; the feature wants bins 0 .. 511
  float* o = ring_frame(spec, b, ring_pos(spec, hop), hh);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float pw = bsp::fma(xi[u], xi[u], xr[u] * xr[u]);
    o[tid + u * 256] = 0.5f * bsp::log(pw + 1e-5f);
  }
}
__device__ __forceinline__ void pitch_fft_body(const FftArgs& a, const int b, const int hh, float* __restrict__ lds) {
  pitch_fft_body_t<1>(a, b, hh, lds, 0);
}
static __global__ __launch_bounds__(256) void pitch_fft_kernel(const FftArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kFftLdsFloats];
  pitch_fft_body(a, blockIdx.x, blockIdx.y, lds);
}
struct FftOp {
  using Args = FftArgs;
  static constexpr int NTHR = 256;
  static constexpr int LDS_FLOATS = kFftLdsFloats;
  __device__ static __forceinline__ void run(const Args& a, int bx, int by, float* lds) { pitch_fft_body(a, bx, by, lds); }
};
struct FftArgs2 { FftArgs a; int n_streams; };
__device__ __forceinline__ void globalize(FftArgs2& a) { globalize(a.a); }
struct FftOp2 {  // two streams per 512-thread workgroup (H = 1): grid ((n_streams + 1) / 2, 1)
  using Args = FftArgs2;
  static constexpr int NTHR = 512;
  static constexpr int LDS_FLOATS = 4 * kFftArr + B_FFT_N;
  __device__ static __forceinline__ void run(const Args& a, int bx, int by, float* lds) { pitch_fft_body_t<2>(a.a, bx, by, lds, a.n_streams); }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int by, float* lds) { pitch_fft_body_t<2, RAG>(a.a, bx, by, lds, a.n_streams); }
};

// ---------------------------------------------------------------------------------------------
// Pitch head (MODEL_SPEC 4.2.3): masked argmax + 4 features, one wavefront per stream; optional
// per-stream pitch transform (double precision, restating reference
// src/common/processor_core_2.cc:190-252) so that the batched path needs no host round trip.
struct PitchParams {  // per stream
  double average_source_pitch, intonation_intensity, pitch_shift, pitch_correction;
  int pitch_correction_type, pad;
};
struct PitchHeadArgs {
  int H;                // hops per step; all per-hop arrays below are [B][H]...
  Ring logits;          // C = 448, n = H
  Ring h;               // GRU state ring (C=128)
  const float* d_in;    // [B][160]
  const float* voi_w;   // [128]
  const float* voi_b;   // [1]
  const int* min_q;
  const int* max_q;
  int* prev_q;
  int* q_raw;           // [q_slots][B][H]
  int* q_out;           // [q_slots][B][H] (after transform; == q_raw when params == nullptr)
  float* feat;          // [q_slots][B][H][4]
  const PitchParams* params;
  const int* hop;
  size_t io_stride;     // see F1Args (d_in is read for the frame energy)
  int q_slots;          // step slots of the three outputs (step t -> slot t mod q_slots): 1, or 2 when the consumer may lag a step
  int B;
  float* host_out;      // 1-stream ABI (B = 1, H = 1) or null: pinned host block [4 feat | raw bin | sequence word]; the kernel writes the results there itself and the
                        // call's sequence word (mailbox word 3 behind the audio) LAST, behind a system-scope fence: the host polls that word, no copy command, no stream query
};
__device__ __forceinline__ void globalize(PitchHeadArgs& a) {
  globalize(a.logits); globalize(a.h); a.d_in = as_global(a.d_in); a.voi_w = as_global(a.voi_w); a.voi_b = as_global(a.voi_b);
  a.min_q = as_global(a.min_q); a.max_q = as_global(a.max_q); a.prev_q = as_global(a.prev_q); a.q_raw = as_global(a.q_raw);
  a.q_out = as_global(a.q_out); a.feat = as_global(a.feat); a.params = as_global(a.params); a.hop = as_global(a.hop);
}

__device__ inline double pitch_round_half_away(double v) { return v >= 0.0 ? floor(v + 0.5) : -floor(-v + 0.5); }

__device__ inline int pitch_transform_device(int q, const PitchParams& p) {
  const double per = 8.0;
  double t = p.average_source_pitch + ((double)q - p.average_source_pitch) * p.intonation_intensity + per * p.pitch_shift;
  if (p.pitch_correction != 0.0) {
    if (p.pitch_correction_type == 0) {
      const double near = (floor(t / per) + 0.5) * per;
      const double d = (t - near) * (2.0 / per);
      if (fabs(d) < 1e-4) t = near;
      else t = near + d * pow(fabs(d), -p.pitch_correction) * (per / 2.0);
    } else {
      const double near = pitch_round_half_away(t / per) * per;
      const double d = (t - near) * (2.0 / per);
      if (p.pitch_correction > 1 - 1e-4) t = near;
      else if (d >= 0.0) t = near + pow(d, 1.0 / (1.0 - p.pitch_correction)) * (per / 2.0);
      else t = near - pow(-d, 1.0 / (1.0 - p.pitch_correction)) * (per / 2.0);
    }
  }
  // the reference converts to int first (static_cast<int>(std::round(t))) and clamps after
  const double r = pitch_round_half_away(t);
  int qi = r > 2147483647.0 ? 2147483647 : (r < -2147483648.0 ? (int)(-2147483647 - 1) : (int)r);
  return qi < 1 ? 1 : (qi > B_PITCH_BINS - 1 ? B_PITCH_BINS - 1 : qi);
}

template <bool RAG = false>
__device__ __forceinline__ void pitch_head_body(const PitchHeadArgs& a, const int b, const int l = threadIdx.x) {
  const int step = stepc::step(a.hop);
  if (step < 0) return;
  const int hop = stepc::of_t<RAG>(step, b);
  if (hop < 0) return;   // (the stream sits this step out: previous bin, outputs and slots stay as they are)
  const size_t qoff = (size_t)(hop % a.q_slots) * a.B * a.H;
  const int pos_l = ring_pos(a.logits, hop);
  int lo = a.min_q[b], hi = a.max_q[b];
  if (hi < lo) hi = lo;
  int prev = a.prev_q[b];
  for (int hh = 0; hh < a.H; ++hh) {
    const size_t row = (size_t)b * a.H + hh;
    const float* lg = ring_frame(a.logits, b, pos_l, hh);
    const int n_slot = a.logits.C >> 6;   // logits per lane: 7 (448 bins) or 6 (384 bins, the legacy generations)
    float v[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) v[i] = i < n_slot ? lg[l + 64 * i] : 0.0f;
    float bv = -__builtin_huge_valf();
    int bj = 0x7fffffff;
    float mx = v[0];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      if (i >= n_slot) break;
      const int j = l + 64 * i;
      mx = fmaxf(mx, v[i]);
      if (j >= lo && j <= hi && (bj == 0x7fffffff || v[i] > bv)) { bv = v[i]; bj = j; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oj = __shfl_xor(bj, off, 64);
      if (oj != 0x7fffffff && (bj == 0x7fffffff || ov > bv || (ov == bv && oj < bj))) { bv = ov; bj = oj; }
    }
    const int q = bj;
    mx = bsp::wmax64(mx);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {   // (pairs through the packed exp; summed in index order as before)
      const bsp::f32x2 e = bsp::exp2(bsp::f32x2{v[i < 7 ? i : 6] - mx, v[i + 1 < 7 ? i + 1 : 6] - mx});
      if (i < n_slot) s = s + e.x;
      if (i + 1 < 7 && i + 1 < n_slot) s = s + e.y;
    }
    const float f0 = bsp::exp(lg[q] - mx) / bsp::wsum64(s);
    const float* x = a.d_in + (a.io_stride != 0 ? (size_t)stepc::slot(a.hop) * a.io_stride : 0) + row * B_IN_HOP;
    float en = 0.0f;
    for (int i = l; i < B_IN_HOP; i += 64) en = bsp::fma(x[i], x[i], en);
    const float f1 = 0.1f * bsp::log(bsp::fma(bsp::wsum64(en), 1.0f / 160.0f, 1e-8f));
    const float* hv = ring_frame(a.h, b, ring_pos(a.h, hop), hh);
    const float pv = bsp::fma(hv[l + 64], a.voi_w[l + 64], bsp::fma(hv[l], a.voi_w[l], 0.0f));
    const float f3 = bsp::sigmoid(bsp::wsum64(pv) + a.voi_b[0]);
    if (l == 0) {
      float dq = (float)(q - prev) * 0.125f;
      dq = dq < -1.0f ? -1.0f : (dq > 1.0f ? 1.0f : dq);
      float* f = a.feat + (qoff + row) * 4;
      f[0] = f0; f[1] = f1; f[2] = dq; f[3] = f3;
      a.q_raw[qoff + row] = q;
      a.q_out[qoff + row] = a.params ? pitch_transform_device(q, a.params[b]) : q;
      if (a.host_out != nullptr) {   // (one stream, one hop per step)
        a.host_out[0] = f0; a.host_out[1] = f1; a.host_out[2] = dq; a.host_out[3] = f3;
        reinterpret_cast<int*>(a.host_out)[4] = q;
      }
    }
    prev = q;
  }
  if (l == 0) a.prev_q[b] = prev;
  if (l == 0 && a.host_out != nullptr) {
    const int seq = reinterpret_cast<const int*>(a.d_in + B_IN_HOP)[3];
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<int*>(a.host_out) + 5, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static __global__ __launch_bounds__(64) void pitch_head_kernel(const PitchHeadArgs a) { pitch_head_body(a, blockIdx.x); }
struct HeadOp {
  using Args = PitchHeadArgs;
  static constexpr int NTHR = 64;
  static constexpr int LDS_FLOATS = 0;
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float*) { pitch_head_body(a, bx); }
};
struct HeadOp8 {  // one wavefront per stream, eight streams per 512-thread workgroup: grid ((B + 7) / 8, 1)
  using Args = PitchHeadArgs;
  static constexpr int NTHR = 512;
  static constexpr int LDS_FLOATS = 0;
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float*) {
    const int b = bx * 8 + (int)(threadIdx.x >> 6);
    if (b < a.B) pitch_head_body(a, b, (int)(threadIdx.x & 63));
  }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int, float*) {
    const int b = bx * 8 + (int)(threadIdx.x >> 6);
    if (b < a.B) pitch_head_body<RAG>(a, b, (int)(threadIdx.x & 63));
  }
};

// ---------------------------------------------------------------------------------------------
// Waveform input mix, conditioning part (MODEL_SPEC 4.4.1):
//   e[b][n] = (pitch_emb[q][n] + Wf.feat[b]) + (add_tab[add_idx[b]][n] + frm_tab[frm_idx[b]][n])
struct CondArgs {
  int H;               // hops per step; q/feat/e rows are (stream, hop)
  const int* q;        // [q_slots][B][H]
  const float* feat;   // [q_slots][B][H][4]
  int q_slots, B;      // see PitchHeadArgs
  const float* pitch_emb;
  const float* feat_w; // [4][256]
  const float* add_tab; const int* add_idx;
  const float* frm_tab; const int* frm_idx;
  Ring e;              // C = 256, n = H, m = 1 or 2 step slots
  const int* hop;
  int* hop_next_out;   // optional (batch): row 0 stores {counter + 1, next resident-I/O slot} for the next step's first kernels;
                       // no kernel of this step's front end reads that pair after its first launch
  int io_slots;
  int n_bins;          // rows of pitch_emb: 448, or 384 in the legacy generations -- which also have no formant table
                       // (frm_tab == nullptr): their conditioning is the ONE row add_tab[add_idx] the host hands over per hop
};
__device__ __forceinline__ void globalize(CondArgs& a) {
  a.q = as_global(a.q); a.feat = as_global(a.feat); a.pitch_emb = as_global(a.pitch_emb); a.feat_w = as_global(a.feat_w);
  a.add_tab = as_global(a.add_tab); a.add_idx = as_global(a.add_idx); a.frm_tab = as_global(a.frm_tab); a.frm_idx = as_global(a.frm_idx);
  globalize(a.e); a.hop = as_global(a.hop); a.hop_next_out = as_global(a.hop_next_out);
}
template <bool RAG = false>
__device__ __forceinline__ void wave_cond_body(const CondArgs& a, const int row, const int n = threadIdx.x) {
  const int b = row / a.H;
  const int step = stepc::step(a.hop);
  if (step < 0) return;
  const int hop = stepc::of_t<RAG>(step, b);
  if (hop < 0) return;   // (the stream sits this step out)
  const size_t qoff = (size_t)(hop % a.q_slots) * a.B * a.H;
  int q = a.q[qoff + row];
  q = q < 0 ? 0 : (q > a.n_bins - 1 ? a.n_bins - 1 : q);
  const float* f = a.feat + (qoff + row) * 4;
  float fp = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) fp = bsp::fma(f[i], a.feat_w[i * B_HID + n], fp);
  float c = a.add_tab[(size_t)a.add_idx[b] * B_HID + n];
  if (a.frm_tab != nullptr) c = c + a.frm_tab[(size_t)a.frm_idx[b] * B_HID + n];
  ring_frame(a.e, b, ring_pos(a.e, hop), row % a.H)[n] = (a.pitch_emb[(size_t)q * B_HID + n] + fp) + c;
  if (a.hop_next_out != nullptr && row == 0 && n == 0) {
    const int io = a.io_slots > 0 ? stepc::slot(a.hop) : 0;
    a.hop_next_out[0] = hop_next(step);
    a.hop_next_out[1] = a.io_slots > 0 ? (io + 1 >= a.io_slots ? 0 : io + 1) : 0;
  }
}
static __global__ __launch_bounds__(256) void wave_cond_kernel(const CondArgs a) { wave_cond_body(a, blockIdx.x); }
struct CondOp {
  using Args = CondArgs;
  static constexpr int NTHR = 256;
  static constexpr int LDS_FLOATS = 0;
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float*) { wave_cond_body(a, bx); }
};
struct CondOp2 {  // two rows per 512-thread workgroup: grid ((rows + 1) / 2, 1), rows = B * H
  using Args = CondArgs;
  static constexpr int NTHR = 512;
  static constexpr int LDS_FLOATS = 0;
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float*) {
    const int row = bx * 2 + (int)(threadIdx.x >> 8);
    if (row < a.B * a.H) wave_cond_body(a, row, (int)(threadIdx.x & 255));
  }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int, float*) {
    const int row = bx * 2 + (int)(threadIdx.x >> 8);
    if (row < a.B * a.H) wave_cond_body<RAG>(a, row, (int)(threadIdx.x & 255));
  }
};

// ---------------------------------------------------------------------------------------------
// Set-time kernels (embedding setter, MODEL_SPEC 4.3).  Not on the per-hop path.
// y[row][n] = bias[n] + chain_c x[row][c] * w[c][n]
static __global__ void dense_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ bias, float* __restrict__ y, int rows,
                                  int cin, int cout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cout) return;
  const int r = idx / cout, n = idx % cout;
  float acc = 0.0f;
  for (int c = 0; c < cin; ++c) acc = bsp::fma(x[(size_t)r * cin + c], w[(size_t)c * cout + n], acc);
  y[idx] = acc + bias[n];
}

__device__ __forceinline__ size_t packed_w_offset_dev(int K, int kk, int n) {
  return ((((size_t)(n >> 4) * (K >> 4) + (kk >> 4)) * 64 + ((kk & 3) << 4) + (n & 15)) << 2) + ((kk & 15) >> 2);
}
// K^T (k = channel, n = token) and V (k = token, n = channel) of one block for `slots` raw embeddings:
// grid (384, slots), 256 threads.
static __global__ __launch_bounds__(256) void kv_project_kernel(const float* __restrict__ kv_raw,
                                                         const float* __restrict__ kw, const float* __restrict__ kb,
                                                         const float* __restrict__ vw, const float* __restrict__ vb,
                                                         float* __restrict__ kt, float* __restrict__ v,
                                                         float* __restrict__ kt_plain, float* __restrict__ v_plain) {
  __shared__ float row[B_KV_CH];
  const int j = blockIdx.x, slot = blockIdx.y, c = threadIdx.x;
  if (c < B_KV_CH) row[c] = kv_raw[((size_t)slot * B_KV_LEN + j) * B_KV_CH + c];
  __syncthreads();
  float ak = 0.0f, av = 0.0f;
  for (int e = 0; e < B_KV_CH; ++e) {
    ak = bsp::fma(row[e], kw[e * B_HID + c], ak);
    av = bsp::fma(row[e], vw[e * B_HID + c], av);
  }
  // both tables are GEMM "weights" of the attention products and are written in MFMA B-fragment
  // order (conv_gemm.hip.h packed_w_offset): scores = q . K^T has (k = channel c, n = token j),
  // P . V has (k = token j, n = channel c)
  kt[(size_t)slot * B_HID * B_KV_LEN + packed_w_offset_dev(B_HID, c, j)] = ak + kb[c];
  v[(size_t)slot * B_KV_LEN * B_HID + packed_w_offset_dev(B_KV_LEN, j, c)] = av + vb[c];
  // the same values in plain order for the multi-block (4x4x1) attention path: K^T [channel][token], V [token][channel]
  if (kt_plain) kt_plain[(size_t)slot * B_HID * B_KV_LEN + (size_t)c * B_KV_LEN + j] = ak + kb[c];
  if (v_plain) v_plain[(size_t)slot * B_KV_LEN * B_HID + (size_t)j * B_HID + c] = av + vb[c];
}

// codebook [512][128] -> transposed [128][512] + squared norms; grid = n codebooks, 512 threads.
static __global__ __launch_bounds__(512) void codebook_prep_kernel(const float* __restrict__ cb, float* __restrict__ cbT,
                                                            float* __restrict__ cnorm) {
  const int s = blockIdx.x, j = threadIdx.x;
  const float* src = cb + ((size_t)s * B_CODEBOOK + j) * B_PHONE_CH;
  float* dstT = cbT + (size_t)s * B_PHONE_CH * B_CODEBOOK;
  float a = 0.0f;
  for (int c = 0; c < B_PHONE_CH; ++c) {
    const float x = src[c];
    a = bsp::fma(x, x, a);
    dstT[(size_t)c * B_CODEBOOK + j] = x;
  }
  cnorm[(size_t)s * B_CODEBOOK + j] = a;
}

// ---------------------------------------------------------------------------------------------
// 48 kHz host-rate wrapper on the device (BASELINE.json configs[4]; reference chain for a 48 kHz
// host: src/vst/processor.cc:183-192 downmix, src/common/resample.h Downsample/Upsample at ratio
// 1/1, :343-363 block FIFO, :380-394 decimate by 3 / zero-stuff by 2).  Gains are the 0 dB identity
// here (non-zero gains stay in the C++ host layer).  Accumulations are mul-then-add in the
// reference's tap order, so results equal the host chain bit for bit.
struct Wrap48State {   // per stream, floats
  float hist_in[30];   // newest 30 mono input samples of the previous block
  float ztail[16];     // last 16 model outputs of the block emitted one call earlier
  float fpend[240];    // model output waiting in the 480-sample FIFO (emitted by the NEXT call)
};

// ---- the shell's silent-block rule per stream (reference src/vst/processor.cc:204-214: a block whose down-mix is all zeros is
// not converted -- the core is not called, its state and its 10 ms FIFO stand still, the output is that down-mix) ------------
// A batch advances all of its streams together and indexes every ring by the common step counter, so a stream cannot simply
// be left out of a step.  Instead the step runs for every stream and the silent ones are put back afterwards: a ring with
// m >= 2 step slots is ROTATED (slot of step t - k <- slot of step t - k - 1, k = 0 .. m - 2: seen from step t + 1 the
// stream's history is what it was before the silent block; the slot the step wrote held step t - m, which nothing reads any
// more), a single-slot ring (state updated in place: the tail's history block; scratch) and the pitch head's previous bin
// are RESTORED from a copy taken before the step.
struct FreezeRing { float* base; int slot_floats; int m; unsigned long long keep_off; };   // slot_floats = n C; keep_off: m == 1 only
static __global__ __launch_bounds__(256) void freeze_save_kernel(const FreezeRing* __restrict__ rings, float* __restrict__ keep, const int B,
                                                                 const int* __restrict__ prev_q, int* __restrict__ keep_prev_q) {
  const FreezeRing r = rings[blockIdx.x];
  const int b = blockIdx.y;
  if (blockIdx.x == 0 && threadIdx.x == 0) keep_prev_q[b] = prev_q[b];
  if (r.m != 1) return;
  const float* src = r.base + (size_t)b * r.slot_floats;
  float* dst = keep + r.keep_off + (size_t)b * r.slot_floats;
  for (int i = threadIdx.x; i < r.slot_floats; i += 256) dst[i] = src[i];
}
static __global__ __launch_bounds__(256) void freeze_fix_kernel(const FreezeRing* __restrict__ rings, const float* __restrict__ keep, const int B,
                                                                const unsigned char* __restrict__ frozen, const int hop,
                                                                int* __restrict__ prev_q, const int* __restrict__ keep_prev_q) {
  const int b = blockIdx.y;
  if (!frozen[b]) return;
  const FreezeRing r = rings[blockIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) prev_q[b] = keep_prev_q[b];
  float* base = r.base + (size_t)b * r.slot_floats * r.m;
  if (r.m == 1) {
    const float* src = keep + r.keep_off + (size_t)b * r.slot_floats;
    for (int i = threadIdx.x; i < r.slot_floats; i += 256) base[i] = src[i];
    return;
  }
  const int cur = hop % r.m;
  for (int i = threadIdx.x; i < r.slot_floats; i += 256) {
    int d = cur;
    for (int k = 0; k + 1 < r.m; ++k) {   // (a thread touches element i of every slot only: no cross-thread ordering needed)
      const int s = d == 0 ? r.m - 1 : d - 1;
      base[(size_t)d * r.slot_floats + i] = base[(size_t)s * r.slot_floats + i];
      d = s;
    }
  }
}

// Streams of a batch back to ONE step counter (tick mode with ragged steps, at a drained point): stream b has sat shift[b] steps out
// in total, i.e. its rings are indexed by a counter that lags the batch's by shift[b].  Every slot of every ring moves forward by
// shift[b] (mod the ring's slot count): what the stream wrote at its own step h then sits where the common counter h + shift[b]
// points, so the next step at the common counter finds the stream's whole history in the right places.  Single-slot rings are not
// indexed by the counter.  Grid (rings, streams); a thread moves element i of every slot (no cross-thread ordering needed).
constexpr int kMaxRotateSlots = 64;
static __global__ __launch_bounds__(256) void ring_rotate_kernel(const FreezeRing* __restrict__ rings, const int* __restrict__ shift) {
  const FreezeRing r = rings[blockIdx.x];
  const int b = blockIdx.y;
  const int d = r.m > 1 ? shift[b] % r.m : 0;
  if (d == 0) return;
  float* base = r.base + (size_t)b * r.slot_floats * r.m;
  for (int i = threadIdx.x; i < r.slot_floats; i += 256) {
    float v[kMaxRotateSlots];
    for (int j = 0; j < r.m; ++j) v[j] = base[(size_t)j * r.slot_floats + i];
    for (int j = 0; j < r.m; ++j) { const int t = j + d >= r.m ? j + d - r.m : j + d; base[(size_t)t * r.slot_floats + i] = v[j]; }
  }
}

// mono = (L + R) * 0.5 (or L), 31-tap low-pass evaluated only at the samples the decimator keeps
// (frozen != nullptr and frozen[b]: the stream's block is silent by the shell's rule -- nothing of its state moves)
// (row: the block's index in in48 / in16 -- the stream, or (stream, hop in step) with several blocks per step; b: the stream's state)
__device__ __forceinline__ void wrap48_pre_body(const int b, const float* __restrict__ in48, int channels, Wrap48State* __restrict__ st,
                                                const float* __restrict__ coef_down, float* __restrict__ in16, float* __restrict__ lds,
                                                const unsigned char* __restrict__ frozen = nullptr, int row = -1) {
  if (row < 0) row = b;
  float* g = lds;        // [30 + 480]
  float* cd = lds + 512; // [33]
  const int tid = threadIdx.x;
  const float* src = in48 + (size_t)row * channels * 480;
  if (tid < 33) cd[tid] = coef_down[tid];
  if (tid < 30) g[tid] = st[b].hist_in[tid];
  for (int i = tid; i < 480; i += 256) {
    float m = src[i];
    if (channels >= 2) { m = m + src[480 + i]; m = m * 0.5f; }
    g[30 + i] = m;
  }
  __syncthreads();
  if (tid < 160) {
    const int p = 3 * tid + 2;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 31; ++i) acc = acc + g[30 + p - i] * cd[1 + i];
    in16[(size_t)row * 160 + tid] = acc * 1.0f;
  }
  if (tid < 30 && !(frozen && frozen[b])) st[b].hist_in[tid] = g[480 + tid];
}
static __global__ __launch_bounds__(256) void wrap48_pre_kernel(const float* __restrict__ in48, int channels,
                                                                Wrap48State* __restrict__ st, const float* __restrict__ coef_down,
                                                                float* __restrict__ in16, const unsigned char* __restrict__ frozen = nullptr) {
  __shared__ float lds[512 + 33];
  wrap48_pre_body(blockIdx.x, in48, channels, st, coef_down, in16, lds, frozen);
}

// zero-stuffed previous model output through the 32-tap low-pass; writes every channel.  latch != nullptr: the model
// output of this step ([B][240]) then takes the FIFO's place for the next block (what wrap48_latch_kernel does).
__device__ __forceinline__ void wrap48_post_body(const int b, Wrap48State* __restrict__ st, const float* __restrict__ coef_up,
                                                 float* __restrict__ out48, int channels, const float* __restrict__ latch,
                                                 float* __restrict__ lds, int row = -1) {
  if (row < 0) row = b;
  float* f = lds;         // [16 + 240]
  float* cu = lds + 256;  // [33]
  const int tid = threadIdx.x;
  if (tid < 33) cu[tid] = coef_up[tid];
  if (tid < 16) f[tid] = st[b].ztail[tid];
  if (tid < 240) f[16 + tid] = st[b].fpend[tid];
  __syncthreads();
  float* dst = out48 + (size_t)row * channels * 480;
  for (int n = tid; n < 480; n += 256) {
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int m = n - i;                     // index into the zero-stuffed stream; odd positions are zero
      if ((m & 1) == 0) acc = acc + f[16 + (m >> 1)] * cu[i];   // m >= -31 -> (m >> 1) >= -16
    }
    for (int c = 0; c < channels; ++c) dst[c * 480 + n] = acc;
  }
  __syncthreads();
  if (tid < 16) st[b].ztail[tid] = f[16 + 224 + tid];
  if (latch != nullptr && tid < 240) st[b].fpend[tid] = latch[(size_t)row * 240 + tid];
}
// (frozen streams: the output block is the block's own down-mix -- zeros -- on every channel, the FIFO stands still)
static __global__ __launch_bounds__(256) void wrap48_post_kernel(Wrap48State* __restrict__ st, const float* __restrict__ coef_up,
                                                                 float* __restrict__ out48, int channels, const unsigned char* __restrict__ frozen = nullptr,
                                                                 const float* __restrict__ in48 = nullptr) {
  __shared__ float lds[256 + 33];
  if (frozen && frozen[blockIdx.x]) {
    const float* src = in48 + (size_t)blockIdx.x * channels * 480;
    float* dst = out48 + (size_t)blockIdx.x * channels * 480;
    for (int n = threadIdx.x; n < 480; n += 256) {
      float m = src[n];
      if (channels >= 2) { m = m + src[480 + n]; m = m * 0.5f; }
      for (int c = 0; c < channels; ++c) dst[c * 480 + n] = m;
    }
    return;
  }
  wrap48_post_body(blockIdx.x, st, coef_up, out48, channels, nullptr, lds);
}

static __global__ void wrap48_latch_kernel(Wrap48State* __restrict__ st, const float* __restrict__ model_out, int B, const unsigned char* __restrict__ frozen = nullptr) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < B * 240 && !(frozen && frozen[idx / 240])) st[idx / 240].fpend[idx % 240] = model_out[idx];
}

// Both ends of the wrapper around one tick of the pipelined batch, one launch: workgroups [0, n_pre) turn the 48 kHz block
// entering the pipeline into its 16 kHz hop, workgroups [n_pre, n_pre + n_post) emit the 48 kHz block of the step that
// left the pipeline one tick earlier and latch its model output (they touch disjoint parts of Wrap48State).
struct Wrap48TickArgs {
  int n_pre, n_post, channels;
  Wrap48State* st;
  const float *coef_down, *coef_up;
  const float* in48; float* in16;          // pre: slot of the step fed by this tick
  float* out48; const float* model_out;    // post: slot of the step completed by the previous tick
  // ragged steps (the silent-block rule per stream): the step counters of the entering / the completed step's streams, -1 =
  // the stream's block is silent (pre: its filter history stands still; post: its output block is its own zero down-mix,
  // FIFO and latch untouched); null: every stream takes part
  const int *hv_pre, *hv_post;
  const float* in48_post;                  // the completed step's own input block (read for silent streams only)
  int H;                                   // blocks per stream and step (a batch with several hops per step): rows are (stream, hop), in order
};
// The step's H blocks of a stream, in order, operation for operation wrap48_pre_body / wrap48_post_body -- with what goes from one
// block to the next (the decimator's 30 samples of history; the up-sampler's 16-sample tail and the latched model output) carried in
// registers instead of through the state in global memory: a store -> load round trip per block was the launch's critical path at
// four blocks per step.
__device__ __forceinline__ void wrap48_pre_blocks(const int b, const Wrap48TickArgs& a, float* __restrict__ lds) {
  float* g = lds;        // [30 + 480]
  float* cd = lds + 512; // [33]
  const int tid = threadIdx.x;
  if (tid < 33) cd[tid] = a.coef_down[tid];
  float hk = tid < 30 ? a.st[b].hist_in[tid] : 0.0f;
  for (int hh = 0; hh < a.H; ++hh) {
    const int row = b * a.H + hh;
    const float* src = a.in48 + (size_t)row * a.channels * 480;
    if (hh > 0) __syncthreads();
    if (tid < 30) g[tid] = hk;
    for (int i = tid; i < 480; i += 256) {
      float m = src[i];
      if (a.channels >= 2) { m = m + src[480 + i]; m = m * 0.5f; }
      g[30 + i] = m;
    }
    __syncthreads();
    if (tid < 160) {
      const int p = 3 * tid + 2;
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < 31; ++i) acc = acc + g[30 + p - i] * cd[1 + i];
      a.in16[(size_t)row * 160 + tid] = acc * 1.0f;
    }
    if (tid < 30) hk = g[480 + tid];
  }
  if (tid < 30) a.st[b].hist_in[tid] = hk;
}
__device__ __forceinline__ void wrap48_post_blocks(const int b, const Wrap48TickArgs& a, float* __restrict__ lds) {
  float* f = lds;         // [16 + 240]
  float* cu = lds + 256;  // [33]
  const int tid = threadIdx.x;
  if (tid < 33) cu[tid] = a.coef_up[tid];
  float zt = tid < 16 ? a.st[b].ztail[tid] : 0.0f;
  float fp = tid < 240 ? a.st[b].fpend[tid] : 0.0f;
  for (int hh = 0; hh < a.H; ++hh) {
    const int row = b * a.H + hh;
    if (hh > 0) __syncthreads();
    if (tid < 16) f[tid] = zt;
    if (tid < 240) f[16 + tid] = fp;
    __syncthreads();
    float* dst = a.out48 + (size_t)row * a.channels * 480;
    for (int n = tid; n < 480; n += 256) {
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int m = n - i;                     // index into the zero-stuffed stream; odd positions are zero
        if ((m & 1) == 0) acc = acc + f[16 + (m >> 1)] * cu[i];
      }
      for (int c = 0; c < a.channels; ++c) dst[c * 480 + n] = acc;
    }
    if (tid < 16) zt = f[16 + 224 + tid];
    if (tid < 240) fp = a.model_out[(size_t)row * 240 + tid];   // this block's model output is the next block's FIFO content
  }
  if (tid < 16) a.st[b].ztail[tid] = zt;
  if (tid < 240) a.st[b].fpend[tid] = fp;
}
// Round 6: the blocks of a step side by side.  What one block hands the next is a function of the step's INPUTS alone -- the decimator's history
// = the last 30 down-mixed samples of the block before, the up-sampler's FIFO = the model output of the hop before, its 16-sample tail = the end of
// the FIFO content before that -- so a workgroup can fetch it instead of waiting for a neighbour: one workgroup per (stream, block) for the input half,
// and for the output half one per block except that blocks 0 and 1 share one (both read the state the launch found; that workgroup also writes the
// state the launch leaves, from the inputs, after it has read the old one).  Operation for operation the bodies above: the same samples.  A step's
// wrapper launch was a chain of H blocks per workgroup (18 us at four blocks, 128 workgroups on 256 compute units): now of one or two.
__device__ __forceinline__ float wrap48_mono(const float* __restrict__ src, const int channels, const int i) {
  float m = src[i];
  if (channels >= 2) { m = m + src[480 + i]; m = m * 0.5f; }
  return m;
}
__device__ __forceinline__ void wrap48_pre_block_par(const int b, const int hh, const Wrap48TickArgs& a, float* __restrict__ lds) {
  float* g = lds;        // [30 + 480]
  float* cd = lds + 512; // [33]
  const int tid = threadIdx.x;
  const int row = b * a.H + hh;
  const float* src = a.in48 + (size_t)row * a.channels * 480;
  if (tid < 33) cd[tid] = a.coef_down[tid];
  if (tid < 30) g[tid] = hh == 0 ? a.st[b].hist_in[tid] : wrap48_mono(src - (size_t)a.channels * 480, a.channels, 450 + tid);
  for (int i = tid; i < 480; i += 256) g[30 + i] = wrap48_mono(src, a.channels, i);
  __syncthreads();
  if (tid < 160) {
    const int p = 3 * tid + 2;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 31; ++i) acc = acc + g[30 + p - i] * cd[1 + i];
    a.in16[(size_t)row * 160 + tid] = acc * 1.0f;
  }
  // (the history the launch leaves: the end of the step's LAST block; written by the one workgroup that read the old one, behind that read)
  if (hh == 0 && tid < 30) a.st[b].hist_in[tid] = wrap48_mono(a.in48 + (size_t)(b * a.H + a.H - 1) * a.channels * 480, a.channels, 450 + tid);
}
__device__ __forceinline__ void wrap48_post_blocks_par(const int b, const int h0, const int nb, const Wrap48TickArgs& a, float* __restrict__ lds) {
  float* f = lds;         // [16 + 240]
  float* cu = lds + 256;  // [33]
  const int tid = threadIdx.x;
  if (tid < 33) cu[tid] = a.coef_up[tid];
  const float* mo = a.model_out + (size_t)b * a.H * 240;   // the step's model outputs of this stream, hop after hop
  float zt = 0.0f, fp = 0.0f;
  if (h0 == 0) {
    if (tid < 16) zt = a.st[b].ztail[tid];
    if (tid < 240) fp = a.st[b].fpend[tid];
  } else {   // (h0 >= 2: blocks 0 and 1 are one workgroup's)
    if (tid < 16) zt = mo[(size_t)(h0 - 2) * 240 + 224 + tid];
    if (tid < 240) fp = mo[(size_t)(h0 - 1) * 240 + tid];
  }
  for (int hh = h0; hh < h0 + nb; ++hh) {
    const int row = b * a.H + hh;
    if (hh > h0) __syncthreads();
    if (tid < 16) f[tid] = zt;
    if (tid < 240) f[16 + tid] = fp;
    __syncthreads();
    float* dst = a.out48 + (size_t)row * a.channels * 480;
    for (int n = tid; n < 480; n += 256) {
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int m = n - i;                     // index into the zero-stuffed stream; odd positions are zero
        if ((m & 1) == 0) acc = acc + f[16 + (m >> 1)] * cu[i];
      }
      for (int c = 0; c < a.channels; ++c) dst[c * 480 + n] = acc;
    }
    if (tid < 16) zt = f[16 + 224 + tid];
    if (tid < 240) fp = mo[(size_t)hh * 240 + tid];
  }
  if (h0 == 0) {   // the state the launch leaves (the old one was read above, by these threads)
    if (tid < 16) a.st[b].ztail[tid] = mo[(size_t)(a.H - 2) * 240 + 224 + tid];
    if (tid < 240) a.st[b].fpend[tid] = mo[(size_t)(a.H - 1) * 240 + tid];
  }
}
// workgroups of the launch: H > 1 without ragged steps = the side-by-side form (H per stream for the input half, H - 1 for the output half)
static inline int wrap48_tick_grid(const Wrap48TickArgs& a) {
  const bool par = a.H > 1 && a.hv_pre == nullptr && a.hv_post == nullptr;
  return par ? a.n_pre * a.H + a.n_post * (a.H - 1) : a.n_pre + a.n_post;
}
static __global__ __launch_bounds__(256) void wrap48_tick_kernel(const Wrap48TickArgs a) {
  __shared__ float lds[512 + 33];
  int w = blockIdx.x;
  if (a.H > 1 && a.hv_pre == nullptr && a.hv_post == nullptr) {
    const int npre = a.n_pre * a.H;
    if (w < npre) { wrap48_pre_block_par(w / a.H, w % a.H, a, lds); return; }
    w -= npre;
    const int b = w / (a.H - 1), j = w % (a.H - 1);
    wrap48_post_blocks_par(b, j == 0 ? 0 : j + 1, j == 0 ? 2 : 1, a, lds);
    return;
  }
  if (w < a.n_pre) {
    if (a.hv_pre != nullptr && a.hv_pre[w] < 0) return;   // (its 16 kHz hop is not read either: the model sits the step out)
    wrap48_pre_blocks(w, a, lds);
  } else {
    const int b = w - a.n_pre;
    if (a.hv_post != nullptr && a.hv_post[b] < 0) {   // (the stream sat the step out: every block of it)
      for (int hh = 0; hh < a.H; ++hh) {
        const float* src = a.in48_post + ((size_t)b * a.H + hh) * a.channels * 480;
        float* dst = a.out48 + ((size_t)b * a.H + hh) * a.channels * 480;
        for (int n = threadIdx.x; n < 480; n += 256) {
          float m = src[n];
          if (a.channels >= 2) { m = m + src[480 + n]; m = m * 0.5f; }
          for (int c = 0; c < a.channels; ++c) dst[c * 480 + n] = m;
        }
      }
      return;
    }
    wrap48_post_blocks(b, a, lds);
  }
}
