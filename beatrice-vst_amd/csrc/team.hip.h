// team.hip.h -- the 1-stream chain of a module as ONE launch: a TEAM of workgroups that hands every layer's output vector
// from its producers to all of its consumers inside the launch (BASELINE.json configs[1], the reference's own use: one stream,
// one hop per call, processor_core_2.cc:184,188,253).
//
// Why: with one stream a layer is a [1-8 rows] x [K x N] product; as a launch of its own it costs 4.6-8 us whatever it
// computes (launch + fill + a dependent chain of global round trips, profiles/r03_b1_per_kernel.txt) and the chain is 49 deep:
// 330 us per hop.  Inside one launch the price of a layer boundary is the all-to-all EDGE -- every consumer needs the whole
// vector -- measured at 1.1-1.2 us for 8-32 workgroups when the data itself carries the flag (tools/microbench/team_edge.hip,
// profiles/r04_notes.md section 2): each value travels as one naturally aligned 8-byte GRANULE {float value, int tag}, written
// with ONE agent-scope (sc1, write-through) store and polled by the consumers with agent-scope loads until the tag is the
// hop's.  No separate flag, no fence, no barrier: a stage's workgroups start as soon as THEIR inputs have landed (dataflow),
// and a workgroup requests its slice of the layer's weights BEFORE it starts to poll, so the weight stream (which does not
// depend on the previous layer) hides behind the edge.
//
// Numerics: MODEL_SPEC 2.2 as gemv.hip.h -- every 256-long reduction segment of every output is ONE lane's k-ascending
// v_fma_f32 chain from 0 (the f32 MFMA's rounding), segments added in order, then conv_gemm's epilogue operation for
// operation -- so the bits are those of the per-layer kernels and of the oracle.
//
// Layout of work: a stage (layer) is cut into column tiles of 8; workgroup w of the team owns tiles w, w + NWG, ...; inside a
// tile lane (c, p) runs the chain of column c and (row, segment) pair p.  A tensor that is produced inside the launch has an
// exchange buffer of granules for the frames of THIS hop and its ring in global memory for the hops to come (plain stores:
// the next launch sees them); older frames (taps that reach back) are read from the ring with plain loads.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "chain_layers.hip.h"
#include "conv_gemm.hip.h"
#include "ring.h"
#include "spec_math.hip.h"

namespace team {

constexpr int NWG = 32, NTHR = 256, COLS = 8;
constexpr int WS = 260;                 // LDS stride of one (segment, column) weight run: 256 k + 4 (conflict-free 16-byte reads)
constexpr int MAX_XS = 4096;            // floats of gathered input rows (the largest layer: phone.f2, 8 rows x 512)
constexpr int MAX_P = 6;                // reduction segments (pitch.p1: K = 1536)
constexpr int MAX_PAIRS = 32;           // (row, segment) pairs of a layer: one pass of the 256 lanes
constexpr int kLdsFloats = MAX_P * COLS * WS + MAX_XS + MAX_PAIRS * COLS + 64;
constexpr int kSpinLimit = 400000;      // polls before a workgroup gives up (a bug must not hang the GPU): ~0.3 s

using gran_t = unsigned long long;

struct Tensor {
  Ring ring;       // the tensor's history in global memory (frames of earlier hops; this hop's frames are also stored here)
  gran_t* xb;      // granules of this hop's frames [ring.n][ring.C], or nullptr: produced by an earlier launch
};

__device__ __forceinline__ void publish(gran_t* xb, const int idx, const float v, const int tag) {
  const gran_t g = ((gran_t)(unsigned)tag << 32) | (gran_t)__float_as_uint(v);
  __hip_atomic_store(xb + idx, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the value of granule idx once its tag is this hop's; *dead: set when the wait was given up
__device__ __forceinline__ float acquire(const gran_t* xb, const int idx, const int tag, int* dead) {
  gran_t g = __hip_atomic_load(xb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while ((int)(g >> 32) != tag) {
    if (++spins > kSpinLimit || *dead) { *dead = 1; return 0.0f; }
    __builtin_amdgcn_s_sleep(1);
    g = __hip_atomic_load(xb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return __uint_as_float((unsigned)g);
}

// measurement aid (BEATRICE_HIP_TEAM_TRACE, the 1-stream ABI's waveform context): workgroup 0 stamps the shader clock at the
// phase boundaries of every stage -- entered | inputs in LDS | weights in LDS | chains done | published.  The buffer pointer
// travels in the kernel arguments and is null in normal operation (one scalar test per stamp, no memory access).
struct Stamps { unsigned long long* buf; int at; };
__device__ __forceinline__ void stamp(Stamps& st, const int wg) {
  if (st.buf != nullptr) {
    if (wg == 0 && threadIdx.x == 0) st.buf[st.at] = __builtin_readcyclecounter();
    ++st.at;
  }
}

struct StageArgs {
  Tensor in, out, res;
  const float* w;      // packed [K][N] (conv_gemm.hip.h packed_w_offset)
  const float* bias;
  float scale;         // EPI_SCALE
};

// The input rows of a conv layer, [M][K] into LDS: element (row m, flat reduction index kk) = tap / channel / frame as in
// conv_gemm.hip.h; frames of this hop come from the granules, older ones from the ring.  Two phases: EVERY load of the thread is
// issued first (one round trip for all of them, shared with the weight fetch issued just before), then the granules whose tag
// is not yet this hop's are polled.
template <class L>
__device__ __forceinline__ void gather_conv(const Tensor& in, const int hop, const int tag, float* __restrict__ xs, int* dead) {
  constexpr int N_EL = L::T * L::K, NE = (N_EL + NTHR - 1) / NTHR;
  const int pos_in = ring_pos(in.ring, hop);
  gran_t g[NE];
  int gi[NE];   // granule index, or -1: the value came from the ring (in g's low word)
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = threadIdx.x + NTHR * i;
    gi[i] = -1; g[i] = 0;
    if (e < N_EL) {
      const int m = e / L::K, kk = e % L::K;
      const int tap = kk / L::CIN, c = kk % L::CIN;
      const int rel = (m + 1) * L::STRIDE - 1 - (L::KSZ - 1 - tap) * L::DIL;
      if (rel >= 0 && in.xb != nullptr) {
        gi[i] = rel * L::CIN + c;
        g[i] = __hip_atomic_load(in.xb + gi[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        g[i] = (gran_t)__float_as_uint(ring_frame(in.ring, 0, pos_in, rel)[c]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = threadIdx.x + NTHR * i;
    if (e >= N_EL) continue;
    float v;
    if (gi[i] >= 0 && (int)(g[i] >> 32) != tag) v = acquire(in.xb, gi[i], tag, dead);
    else v = __uint_as_float((unsigned)g[i]);
    if constexpr (L::PRE == PRE_LRELU) v = bsp::lrelu(v);
    xs[e] = v;
  }
}

// One reduction segment of one output: acc <- fma(x[k], w[k], acc), k ascending from 0 (MODEL_SPEC 2.2), operands in LDS.
// The chain is one dependent v_fma_f32 per k; what the code around it must not do is make a group of multiply-adds wait for
// its own LDS reads (14 cycles per k measured with a plain loop): straight-line code, batches of 32 k, the NEXT batch's sixteen
// 16-byte reads issued before the current batch's multiply-adds, the order pinned.
__device__ __forceinline__ float chain(const float4* __restrict__ x4, const float4* __restrict__ w4, const int n4 /* k / 4, wave-uniform */) {
  constexpr int BK = 8;   // float4 per operand and batch (n4 is a multiple of it)
  float4 xa[BK], wa[BK], xn[BK], wn[BK];
#pragma unroll
  for (int i = 0; i < BK; ++i) { xa[i] = x4[i]; wa[i] = w4[i]; }
  float acc = 0.0f;
  // (the trip count is deliberately a run-time value: with a constant the compiler unrolls the loop and the chain runs at half
  //  the speed -- measured ~2 200 cycles per 256 k in this form, 3 500-4 800 in the unrolled ones)
  for (int k4 = 0; k4 < n4; k4 += BK) {
    const bool more = k4 + BK < n4;
    if (more) {
#pragma unroll
      for (int i = 0; i < BK; ++i) { xn[i] = x4[k4 + BK + i]; wn[i] = w4[k4 + BK + i]; }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < BK; ++i) {
      acc = bsp::fma(xa[i].x, wa[i].x, acc); acc = bsp::fma(xa[i].y, wa[i].y, acc); acc = bsp::fma(xa[i].z, wa[i].z, acc); acc = bsp::fma(xa[i].w, wa[i].w, acc);
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < BK; ++i) { xa[i] = xn[i]; wa[i] = wn[i]; }
    }
  }
  return acc;
}

// The weights of one column tile of a layer, as the registers that will carry them to LDS: record (16-column tile, k-block) of
// the packed blob = 64 lanes x float4, lane slot 16 kq + (n & 15), component e = W[16 kb + 4 e + kq][n]; a thread fetches two
// float4 per segment: idx = tid + 256 i -> (kb, kq, c).
template <class L>
__device__ __forceinline__ void load_w(const float* __restrict__ w, const int tile, float4 (&r)[MAX_P][2]) {
  constexpr int K = L::K, P = L::P, LAST = K - 256 * (P - 1);
  const int n0 = tile * COLS, tid = threadIdx.x;
  const float4* wrec = reinterpret_cast<const float4*>(w) + (size_t)(n0 >> 4) * (K / 16) * 64 + (n0 & 15);
#pragma unroll
  for (int s = 0; s < P; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + NTHR * i, kb = idx >> 5, kq = (idx >> 3) & 3, c = idx & 7;
      const bool live = s + 1 < P || kb < LAST / 16;
      r[s][i] = live ? wrec[(size_t)(s * 16 + kb) * 64 + kq * 16 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// A stage requests the weights of the NEXT stage's first tile while its own chains run (they depend on nothing the chain
// produces): the ~1-2 us from L2 / Infinity Cache are then off the critical path of every layer but the first.
struct Pre { float4 r[MAX_P][2]; bool ok; };
struct NoLayer {};

// One layer.  `fill(xs)`: null functor = the conv gather above; the attention's P.V stage supplies its own (softmax weights).
// rowscale: EPI_ROWSCALE's factor (LDS or register value, the same for the one row of the stage).
template <class L, class LN, class Fill>
__device__ __forceinline__ void stage(const StageArgs& a, const int hop, const int tag, const int wg, float* __restrict__ lds, int* dead, Stamps& st, Pre& pre,
                                      const float* w_next, Fill fill, const float* rowscale_lds = nullptr) {
  constexpr int K = L::K, P = L::P, M = L::T, N = L::NOUT;
  constexpr int LAST = K - 256 * (P - 1);
  constexpr int NT8 = N / COLS;
  static_assert(N % COLS == 0 && K % 16 == 0 && LAST % 32 == 0 && !L::GROUPED, "layer shape");
  static_assert(P <= MAX_P && M * P <= MAX_PAIRS && M * K <= MAX_XS, "stage does not fit the team's LDS plan");
  float* wbuf = lds;                                  // [P][COLS][WS]
  float* xs = lds + MAX_P * COLS * WS;                // [M][K]
  float* part = xs + MAX_XS;                          // [M * P][COLS]
  const int tid = threadIdx.x;
  if (wg >= NT8) { pre.ok = false; return; }          // (a narrow layer leaves workgroups without a tile: they move on)
  const int pos_out = ring_pos(a.out.ring, hop);
  int pos_res = 0;
  if constexpr (L::RES) pos_res = ring_pos(a.res.ring, hop);
  bool first = true, next_tile_in_pre = false;
  for (int tile = wg; tile < NT8; tile += NWG) {
    // ---- this tile's weights: every load issued now (registers), stored to LDS once the inputs are in -- the round trip
    // hides behind the wait for the producers.  Record (16-column tile, k-block): 64 lanes x float4, lane slot 16 kq + (n & 15),
    // component e of it = W[16 kb + 4 e + kq][n]; a thread fetches 2 float4 per segment: idx -> (kb, kq, c).
    stamp(st, wg);
    const int n0 = tile * COLS;
    float4 wr[MAX_P][2];
    if ((first && pre.ok) || next_tile_in_pre) {   // requested during the stage before / during this stage's previous tile
#pragma unroll
      for (int s = 0; s < P; ++s) { wr[s][0] = pre.r[s][0]; wr[s][1] = pre.r[s][1]; }
    } else {
      load_w<L>(a.w, tile, wr);
    }
    // what the epilogue will need from global memory, requested now as well: thread (m, c) of the epilogue
    float bias_v = 0.0f;
    gran_t res_g = 0;
    const int em = tid / COLS, en = n0 + tid % COLS;
    if (tid < M * COLS) {
      if constexpr (L::EPI == EPI_BIAS) bias_v = a.bias[en];
      if constexpr (L::RES) {
        if (a.res.xb != nullptr) res_g = __hip_atomic_load(a.res.xb + em * N + en, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else res_g = ((gran_t)(unsigned)tag << 32) | (gran_t)__float_as_uint(a.res.ring.base[(unsigned)pos_res * (unsigned)a.res.ring.C + (unsigned)(em * N + en)]);
      }
    }
    if (first) {  // the stage's input rows (once per workgroup and stage)
      fill(xs);
      first = false;
      stamp(st, wg);
    } else {
      __syncthreads();   // (the previous tile's chains are done with wbuf / part)
    }
#pragma unroll
    for (int s = 0; s < P; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + NTHR * i, kb = idx >> 5, kq = (idx >> 3) & 3, c = idx & 7;
        float* d = wbuf + (s * COLS + c) * WS + kb * 16 + kq;
        d[0] = wr[s][i].x; d[4] = wr[s][i].y; d[8] = wr[s][i].z; d[12] = wr[s][i].w;
      }
    __syncthreads();
    stamp(st, wg);
    if (tile + NWG >= NT8) {   // this workgroup's last tile of the stage: the next stage's weights travel during its chains
      pre.ok = false;
      next_tile_in_pre = false;
      if constexpr (!std::is_same<LN, NoLayer>::value) {
        if (wg < LN::NOUT / COLS) { load_w<LN>(w_next, wg, pre.r); pre.ok = true; }
      }
    } else {                   // a layer wider than the team (attention scores, up1): the next tile's weights, likewise
      load_w<L>(a.w, tile + NWG, pre.r);
      next_tile_in_pre = true;
    }
    // ---- the chains: lane (c, pair) = column c, (row m, segment s).  Pairs are dealt to the four wavefronts round-robin
    // (pair = wavefront + 4 j), so that with two segments of different lengths (K = 384) a wavefront's lanes all run the same length
    {
      const int c = tid & 7, pair = (tid >> 6) + 4 * ((tid >> 3) & 7);
      if (pair < M * P) {
        const int m = pair / P, s = pair % P;
        const float4* x4 = reinterpret_cast<const float4*>(xs + m * K + s * 256);
        const float4* w4 = reinterpret_cast<const float4*>(wbuf + (s * COLS + c) * WS);
        const int klen = __builtin_amdgcn_readfirstlane(s + 1 < P ? 256 : LAST);   // (wave-uniform: see above)
        part[pair * COLS + c] = chain(x4, w4, klen / 4);
      }
    }
    __syncthreads();
    stamp(st, wg);
    // ---- epilogue (conv_gemm's, operation for operation): thread (m, c)
    if (tid < M * COLS) {
      const int m = tid / COLS, c = tid % COLS, n = n0 + c;
      float v = part[(m * P) * COLS + c];
#pragma unroll
      for (int s = 1; s < P; ++s) v = v + part[(m * P + s) * COLS + c];
      if constexpr (L::EPI == EPI_BIAS) v = v + bias_v;
      if constexpr (L::EPI == EPI_SCALE) v = v * a.scale;
      if constexpr (L::EPI == EPI_ROWSCALE) v = v * rowscale_lds[m];
      if constexpr (L::ACT == ACT_GELU) v = bsp::gelu(v);
      if constexpr (L::RES) {
        const float r = (int)(res_g >> 32) == tag ? __uint_as_float((unsigned)res_g) : acquire(a.res.xb, m * N + n, tag, dead);
        v = r + v;
      }
      a.out.ring.base[(unsigned)pos_out * (unsigned)a.out.ring.C + (unsigned)(m * N + n)] = v;
      if (a.out.xb != nullptr) publish(a.out.xb, m * N + n, v, tag);
    }
    stamp(st, wg);
  }
  __syncthreads();   // (LDS is reused by the next stage)
}
template <class L, class LN = NoLayer>
__device__ __forceinline__ void conv_stage(const StageArgs& a, const int hop, const int tag, const int wg, float* __restrict__ lds, int* dead, Stamps& st, Pre& pre,
                                           const float* w_next = nullptr) {
  stage<L, LN>(a, hop, tag, wg, lds, dead, st, pre, w_next, [&](float* xs) { gather_conv<L>(a.in, hop, tag, xs, dead); });
}

// ---------------------------------------------------------------------------------------------------------------------
// The waveform generator from the input mix to the stage-2 transposed conv (MODEL_SPEC 4.4.1-4.4.3): 29 layers, one launch.
// In front of it: wave_cond_kernel (the conditioning vector e); behind it: the fused tail (wave_tail.hip.h).
struct WaveTeamArgs {
  Ring phone_in, e;                               // inputs produced by earlier launches
  Tensor x[B_NBLOCKS + 1];
  Tensor h1[B_NBLOCKS], xa[B_NBLOCKS], q[B_NBLOCKS], sc[B_NBLOCKS], o[B_NBLOCKS];   // (rings: the one scratch set; granules: per block)
  Tensor ya1, yb1, yc1, ya2;
  const float *inp_w, *inp_b;
  const float *c1_w[B_NBLOCKS], *c1_b[B_NBLOCKS], *c2_w[B_NBLOCKS], *c2_b[B_NBLOCKS], *q_w[B_NBLOCKS], *q_b[B_NBLOCKS], *o_w[B_NBLOCKS], *o_b[B_NBLOCKS];
  const float *kt[B_NBLOCKS], *v[B_NBLOCKS];      // packed per-slot tables
  const int* tile_slot[B_NBLOCKS];                // [0]: the K/V slot of the one stream's row
  const float *up_w[2], *up_b[2], *ra_w, *ra_b, *rb_w, *rb_b;
  const int* hop;
  int* dead;                                      // device flag: a wait was given up (outputs are garbage; the host reports failure)
  unsigned long long* stamps;                     // measurement aid, normally null
  int with_cond;                                  // the conditioning mix (wave.cond: 256 sums per hop) at the head of the launch, by every
  CondArgs cond;                                  // workgroup redundantly (see PhoneTeamArgs::with_f1): one launch fewer per hop
};

template <int D, class LNEXT>   // LNEXT: the layer that follows the block (the next block's dilated conv, or the first transposed conv)
__device__ __forceinline__ void wave_block(const WaveTeamArgs& a, const int blk, const int hop, const int tag, const int wg, float* lds, int* dead, Stamps& st,
                                           Pre& pre, const float* w_after) {
  using namespace bhip::wave_layers;
  StageArgs s{};
  // h = gelu(Conv(x))
  s.in = a.x[blk]; s.out = a.h1[blk]; s.w = a.c1_w[blk]; s.bias = a.c1_b[blk];
  using SC = Layer<B_HID, B_KV_LEN, 1, 1, 1, 1, PRE_NONE, ACT_NONE, EPI_SCALE, false>;
  using PV = Layer<B_KV_LEN, B_HID, 1, 1, 1, 1, PRE_NONE, ACT_NONE, EPI_ROWSCALE, false>;
  const int slot = a.tile_slot[blk][0];
  const float* kt = a.kt[blk] + (size_t)(slot < 0 ? 0 : slot) * B_HID * B_KV_LEN;
  const float* vv = a.v[blk] + (size_t)(slot < 0 ? 0 : slot) * B_KV_LEN * B_HID;
  conv_stage<C1<D, 1>, C2<1>>(s, hop, tag, wg, lds, dead, st, pre, a.c2_w[blk]);
  // xa = x + Linear(h)
  s.in = a.h1[blk]; s.res = a.x[blk]; s.out = a.xa[blk]; s.w = a.c2_w[blk]; s.bias = a.c2_b[blk];
  conv_stage<C2<1>, QL<1>>(s, hop, tag, wg, lds, dead, st, pre, a.q_w[blk]);
  // q = Linear(xa)
  s.in = a.xa[blk]; s.out = a.q[blk]; s.w = a.q_w[blk]; s.bias = a.q_b[blk];
  conv_stage<QL<1>, SC>(s, hop, tag, wg, lds, dead, st, pre, kt);
  // s_j = (q . K_j) / 16
  s.in = a.q[blk]; s.out = a.sc[blk]; s.w = kt; s.bias = nullptr; s.scale = 0.0625f;
  conv_stage<SC, PV>(s, hop, tag, wg, lds, dead, st, pre, vv);
  // o = (sum_j e_j V_j) / sum_j e_j, e_j = exp(s_j - max): every workgroup works the softmax weights out for itself (384 exps)
  float* inv = lds + kLdsFloats - 64;
  s.in = a.sc[blk]; s.out = a.o[blk]; s.w = vv;
  stage<PV, C2<1>>(s, hop, tag, wg, lds, dead, st, pre, a.o_w[blk], [&](float* xs) {
    if (threadIdx.x < 64) {   // one wavefront: attn_pv_body's statistics, operation for operation
      const int lane = threadIdx.x;
      float v[6];
      gran_t g[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) g[i] = __hip_atomic_load(a.sc[blk].xb + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = (int)(g[i] >> 32) == tag ? __uint_as_float((unsigned)g[i]) : acquire(a.sc[blk].xb, lane + 64 * i, tag, dead);
      float mx = -__builtin_huge_valf();
#pragma unroll
      for (int i = 0; i < 6; ++i) mx = fmaxf(mx, v[i]);
      mx = bsp::wmax64(mx);
      float sum = 0.0f;
#pragma unroll
      for (int i = 0; i < 6; i += 2) {
        const bsp::f32x2 e = bsp::exp2(bsp::f32x2{v[i] - mx, v[i + 1] - mx});
        sum = sum + e.x; sum = sum + e.y;
        xs[lane + 64 * i] = e.x; xs[lane + 64 * (i + 1)] = e.y;
      }
      const float tot = bsp::wsum64(sum);
      if (lane == 0) inv[0] = 1.0f / tot;
    }
  }, inv);
  // x' = xa + Linear(o)
  s.in = a.o[blk]; s.res = a.xa[blk]; s.out = a.x[blk + 1]; s.w = a.o_w[blk]; s.bias = a.o_b[blk];
  conv_stage<C2<1>, LNEXT>(s, hop, tag, wg, lds, dead, st, pre, w_after);
}

static __global__ __launch_bounds__(NTHR) void wave_team_kernel(const WaveTeamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using namespace bhip::wave_layers;
  const int wg = blockIdx.x;
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
  const int tag = hop + 1;
  Stamps st{a.stamps, 0};
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  __syncthreads();
  if (a.with_cond) {
    wave_cond_body(a.cond, 0);
    __threadfence();   // (e goes to its ring with plain stores and is read with plain loads by the first stage's epilogue)
    __syncthreads();
  }
  StageArgs s{};
  // x0 = Linear(phone) + e
  s.in = Tensor{a.phone_in, nullptr}; s.res = Tensor{a.e, nullptr}; s.out = a.x[0]; s.w = a.inp_w; s.bias = a.inp_b;
  Pre pre;
  pre.ok = false;
  conv_stage<INP<1>, C1<1, 1>>(s, hop, tag, wg, lds, &dead, st, pre, a.c1_w[0]);
  wave_block<1, C1<2, 1>>(a, 0, hop, tag, wg, lds, &dead, st, pre, a.c1_w[1]);
  wave_block<2, C1<4, 1>>(a, 1, hop, tag, wg, lds, &dead, st, pre, a.c1_w[2]);
  wave_block<4, C1<8, 1>>(a, 2, hop, tag, wg, lds, &dead, st, pre, a.c1_w[3]);
  wave_block<8, UP<256, 128, 5, 1>>(a, 3, hop, tag, wg, lds, &dead, st, pre, a.up_w[0]);
  s = StageArgs{};
  s.in = a.x[4]; s.out = a.ya1; s.w = a.up_w[0]; s.bias = a.up_b[0];
  conv_stage<UP<256, 128, 5, 1>, RES<128, 1, 5>>(s, hop, tag, wg, lds, &dead, st, pre, a.ra_w);
  s.in = a.ya1; s.res = a.ya1; s.out = a.yb1; s.w = a.ra_w; s.bias = a.ra_b;
  conv_stage<RES<128, 1, 5>, RES<128, 3, 5>>(s, hop, tag, wg, lds, &dead, st, pre, a.rb_w);
  s.in = a.yb1; s.res = a.yb1; s.out = a.yc1; s.w = a.rb_w; s.bias = a.rb_b;
  conv_stage<RES<128, 3, 5>, UP<128, 64, 4, 5>>(s, hop, tag, wg, lds, &dead, st, pre, a.up_w[1]);
  s.in = a.yc1; s.res = Tensor{}; s.out = a.ya2; s.w = a.up_w[1]; s.bias = a.up_b[1];
  conv_stage<UP<128, 64, 4, 5>>(s, hop, tag, wg, lds, &dead, st, pre);
  if (threadIdx.x == 0 && dead) *a.dead = 1;
  if (st.buf != nullptr && wg == 0 && threadIdx.x == 0) st.buf[1023] = (unsigned long long)st.at;
}

// ---------------------------------------------------------------------------------------------------------------------
// The content encoder's eight convolutions (MODEL_SPEC 4.1.1-4.1.2: f2 .. f5, four residual blocks) and the pitch estimator's
// three (4.2.2): the layers between the per-stream front kernels (phone.f1 / pitch.fft, earlier launches) and the GRUs.
struct PhoneTeamArgs {
  Tensor f[5], rb[4];       // f[0]: written by phone.f1 (an earlier launch: no granules); rb[3]: read by the GRU launch
  const float *f_w[4], *f_b[4], *rb_w[4], *rb_b[4];
  const int* hop;
  int* dead;
  // with_f1: the module's first layer (phone.f1: Conv1d(1 -> 64) on the hop's audio, VALU work of a few microseconds) runs at the
  // head of the launch instead of as a launch of its own -- by EVERY workgroup of the team, redundantly: each then reads the f[0]
  // frames (and the audio ring) it wrote itself, all write the same bits, nothing is exchanged, and a launch boundary (~3 us of
  // dependent-launch gap + the launch) is gone from the hop
  int with_f1;
  F1Args f1;
};
static __global__ __launch_bounds__(NTHR) void phone_team_kernel(const PhoneTeamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using PL = bhip::PhoneLayers<1>;
  const int wg = blockIdx.x;
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
  const int tag = hop + 1;
  Stamps st{nullptr, 0};
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  __syncthreads();
  if (a.with_f1) {
    phone_f1_body(a.f1, 0, 0, lds);
    __threadfence();   // (the frames go to the ring with plain stores and are gathered with plain loads below)
    __syncthreads();
  }
  StageArgs s{};
  s.in = a.f[0]; s.out = a.f[1]; s.w = a.f_w[0]; s.bias = a.f_b[0];
  Pre pre;
  pre.ok = false;
  conv_stage<PL::F2, PL::F3>(s, hop, tag, wg, lds, &dead, st, pre, a.f_w[1]);
  s.in = a.f[1]; s.out = a.f[2]; s.w = a.f_w[1]; s.bias = a.f_b[1];
  conv_stage<PL::F3, PL::F4>(s, hop, tag, wg, lds, &dead, st, pre, a.f_w[2]);
  s.in = a.f[2]; s.out = a.f[3]; s.w = a.f_w[2]; s.bias = a.f_b[2];
  conv_stage<PL::F4, PL::F5>(s, hop, tag, wg, lds, &dead, st, pre, a.f_w[3]);
  s.in = a.f[3]; s.out = a.f[4]; s.w = a.f_w[3]; s.bias = a.f_b[3];
  conv_stage<PL::F5, PL::RBL>(s, hop, tag, wg, lds, &dead, st, pre, a.rb_w[0]);
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    s.in = i == 0 ? a.f[4] : a.rb[i - 1]; s.res = s.in; s.out = a.rb[i]; s.w = a.rb_w[i]; s.bias = a.rb_b[i];
    conv_stage<PL::RBL, PL::RBL>(s, hop, tag, wg, lds, &dead, st, pre, a.rb_w[i < 3 ? i + 1 : 3]);   // (after the last block: a fetch nobody uses)
  }
  if (threadIdx.x == 0 && dead) *a.dead = 1;
}
constexpr size_t kPhoneGranules = 8 * 128 + 4 * 256 + 2 * 256 + 256 + 4 * 256;   // f[1..4], rb[0..3]

struct PitchTeamArgs {
  Tensor spec, p[3];        // spec: written by pitch.fft (an earlier launch); p[2]: read by the GRU launch
  const float *p_w[3], *p_b[3];
  const int* hop;
  int* dead;
};
static __global__ __launch_bounds__(NTHR) void pitch_team_kernel(const PitchTeamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using QL = bhip::PitchLayers<1>;
  const int wg = blockIdx.x;
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
  const int tag = hop + 1;
  Stamps st{nullptr, 0};
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  __syncthreads();
  StageArgs s{};
  s.in = a.spec; s.out = a.p[0]; s.w = a.p_w[0]; s.bias = a.p_b[0];
  Pre pre;
  pre.ok = false;
  conv_stage<QL::P1, QL::P23>(s, hop, tag, wg, lds, &dead, st, pre, a.p_w[1]);
  s.in = a.p[0]; s.res = a.p[0]; s.out = a.p[1]; s.w = a.p_w[1]; s.bias = a.p_b[1];
  conv_stage<QL::P23, QL::P23>(s, hop, tag, wg, lds, &dead, st, pre, a.p_w[2]);
  s.in = a.p[1]; s.res = a.p[1]; s.out = a.p[2]; s.w = a.p_w[2]; s.bias = a.p_b[2];
  conv_stage<QL::P23>(s, hop, tag, wg, lds, &dead, st, pre);
  if (threadIdx.x == 0 && dead) *a.dead = 1;
}
constexpr size_t kPitchGranules = 3 * 128;
constexpr int kPitchTeamWgs = 16;   // (every layer of it is 128 columns wide: 16 tiles)

// granules of the waveform team: x 5 x 256 | per block h1, xa, q, o (256 each), sc (384) | ya1, yb1, yc1 (640 each)
constexpr size_t kWaveGranules = 5 * 256 + B_NBLOCKS * (4 * 256 + 384) + 3 * 640;

}  // namespace team
