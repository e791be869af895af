// fused_small.hip.h -- two launch-count reductions for the few-row part of the chain
// (profiles/r01_notes.md: at B = 256 every launch costs >= ~4 us whatever it computes).
//
//  * gru_fused_kernel: both gate GEMMs (x.Wih, h.Whh) and the gate math of a GRU cell in one
//    launch (was 3).  Workgroup = 16 streams x 16 hidden units, six wavefronts = {x,h} x {r,z,n};
//    each wavefront runs one 16x16 MFMA chain over K <= 256 (a single MODEL_SPEC 2.2 segment) with
//    B fragments straight from the pre-packed weights and A from an LDS tile; the six partial tiles
//    meet in LDS and 256 threads apply the gates.
//  * attn_pv_kernel: softmax statistics + P.V product + 1/sum scaling in one launch (was 2).
//    Workgroup = 16 streams (one K/V slot) x 32 channels, two k-groups (keys 0..255 | 256..383);
//    every workgroup recomputes the 16 x 384 exponentials of its rows (24 per thread) instead of
//    reading them back from a separate launch.
// Arithmetic and its order are exactly those of the unfused kernels (conv_gemm + gru_gate_kernel,
// attn_softmax_kernel + conv_gemm<PV>), so results stay bit-identical to the oracle.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_gemm.hip.h"

struct GruArgs {
  Ring x, h;                  // x: C = IN, n = 1; h: C = H, n = 1, m = 2 (previous state = frame -1)
  const float *wih, *whh;     // packed [IN][3H], [H][3H]
  const float *bih, *bhh;     // [3H]
  const int* hop;
  int B;
  int t;  // hop within the step: x frame t, previous state = h frame t-1, new state -> h frame t
  // Hops of one step linked INSIDE a launch (the tick launch with several hops per stage, tick.hip.h; null elsewhere): the cell
  // of hop t publishes its state as tagged granules [B][H] of {value, step tag} (one 8-byte agent-scope store each: the data
  // carries its own flag, no fence), the cell of hop t + 1 -- workgroups LATER in the launch's dispatch order -- polls them.
  unsigned long long* link_out;
  const unsigned long long* link_in;
  int* link_dead;   // pinned host word, set when a wait was given up (a bug must not hang the GPU)
  // Row groups a workgroup walks one after the other with its column tile's weights kept in registers (0 or 1: one group per
  // workgroup).  A cell's weights pass a compute unit once per 16 RT rows; with two groups per workgroup once per 32 RT -- at
  // 1 024 streams x 4 hops the tick launch 947 -> 921 us, at 256 streams no gain (fewer, longer workgroups in the linked chains), so
  // the tick table asks for it from 512 streams on (profiles/r05_notes.md section 15).
  int passes;
};
namespace glink {
constexpr int kSpinLimit = 2000000;   // polls before a workgroup gives up: ~1 s
__device__ __forceinline__ void publish(unsigned long long* g, const float v, const int tag) {
  __hip_atomic_store(g, ((unsigned long long)(unsigned)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float acquire(const unsigned long long* g, const int tag, int* dead) {
  unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while ((int)(v >> 32) != tag) {
    if (++spins > kSpinLimit) { *dead = 1; return 0.0f; }
    __builtin_amdgcn_s_sleep(2);
    v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return __uint_as_float((unsigned)v);
}
}  // namespace glink

__device__ __forceinline__ void globalize(GruArgs& a) {
  globalize(a.x); globalize(a.h);
  a.wih = as_global(a.wih); a.whh = as_global(a.whh); a.bih = as_global(a.bih); a.bhh = as_global(a.bhh); a.hop = as_global(a.hop);
  a.link_out = as_global(a.link_out); a.link_in = as_global(a.link_in);
}

// RT = row tiles of 16 streams per workgroup: at 2 the wavefront's weight fragments (held in registers for the whole
// reduction) feed two independent MFMA chains -- half the weight traffic per stream and twice the work per dependent step.
// LINK: bit 0 = publish the new state as granules (a.link_out), bit 1 = take the previous state from granules (a.link_in)
template <int IN, int H, int RT = 1, bool RAG = false, int LINK = 0>
__device__ __forceinline__ void gru_fused_body(const GruArgs& a, const int bx, const int by, float* __restrict__ lds) {
  constexpr int XS = IN + 2, HS = H + 2;  // lds: 16 RT * XS + 16 RT * HS + RT * 6 * 256 floats
  constexpr int ROWS = 16 * RT;
  float* xs = lds;
  float* hs = lds + ROWS * XS;
  float* g6 = hs + ROWS * HS;  // [tile][src][gate][16][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int src = wave / 3, gate = wave % 3;
  const int j0 = by * 16;

  // B fragment of this wave: packed weights, column tile (gate*H + j0)/16, all k-blocks
  constexpr int KB_X = IN / 16, KB_H = H / 16;
  float4 bf[(KB_X > KB_H ? KB_X : KB_H)];
  {
    const int kbn = src == 0 ? KB_X : KB_H;
    const float4* wp = reinterpret_cast<const float4*>(src == 0 ? a.wih : a.whh) + (size_t)((gate * H + j0) >> 4) * kbn * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < (KB_X > KB_H ? KB_X : KB_H); ++kb) bf[kb] = wp[(size_t)(kb < kbn ? kb : 0) * 64];
  }
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
  const bool rag = stepc::rag_t<RAG>();   // (tick launch, ragged steps: every row at its stream's own counter, -1 = the stream sits the step out)
  const int px = ring_pos(a.x, hop), ph = ring_pos(a.h, hop);
  const int passes = a.passes > 1 ? a.passes : 1;   // (GruArgs::passes: row groups walked with the weights above kept in registers)
#pragma unroll 1
  for (int pass = 0; pass < passes; ++pass) {
  const int b0 = (bx * passes + pass) * ROWS;
  if (pass > 0) { if (b0 >= a.B) break; __syncthreads(); }   // (the tiles and the gate block are reused)
  // A tiles -> LDS
  for (int e = tid; e < ROWS * (IN / 4); e += 384) {
    const int r = e / (IN / 4), q = e % (IN / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b0 + r < a.B) {
      const int hr = rag ? stepc::hopv[b0 + r] : hop;
      if (hr >= 0) v = *reinterpret_cast<const float4*>(ring_frame(a.x, b0 + r, rag ? ring_pos(a.x, hr) : px, a.t) + 4 * q);
    }
    float2* d = reinterpret_cast<float2*>(&xs[r * XS + 4 * q]);
    d[0] = make_float2(v.x, v.y); d[1] = make_float2(v.z, v.w);
  }
  for (int e = tid; e < ROWS * (H / 4); e += 384) {
    const int r = e / (H / 4), q = e % (H / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b0 + r < a.B) {
      const int hr = rag ? stepc::hopv[b0 + r] : hop;
      if (hr < 0) {
        // (ragged step: the stream sits the step out -- nothing was published for it and nothing is read; zeros in the tile)
      } else if constexpr ((LINK & 2) != 0) {   // the state the cell of the hop before published in THIS launch
        const unsigned long long* g = a.link_in + (size_t)(b0 + r) * H + 4 * q;
        unsigned long long gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gv[i] = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (int)(gv[i] >> 32) == hop + 1 ? __uint_as_float((unsigned)gv[i]) : glink::acquire(g + i, hop + 1, a.link_dead);
        v = make_float4(f[0], f[1], f[2], f[3]);
      } else v = *reinterpret_cast<const float4*>(ring_frame(a.h, b0 + r, rag ? ring_pos(a.h, hr) : ph, a.t - 1) + 4 * q);
    }
    float2* d = reinterpret_cast<float2*>(&hs[r * HS + 4 * q]);
    d[0] = make_float2(v.x, v.y); d[1] = make_float2(v.z, v.w);
  }
  __syncthreads();
  f32x4 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (src == 0) {
    const float* ap = xs + (lane & 15) * XS + (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < IN / 4; ++ks) {
      const float4 f = bf[ks >> 2];
      const float bv = (ks & 3) == 0 ? f.x : ((ks & 3) == 1 ? f.y : ((ks & 3) == 2 ? f.z : f.w));
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[t * 16 * XS + 4 * ks], bv, acc[t], 0, 0, 0);
    }
  } else {
    const float* ap = hs + (lane & 15) * HS + (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < H / 4; ++ks) {
      const float4 f = bf[ks >> 2];
      const float bv = (ks & 3) == 0 ? f.x : ((ks & 3) == 1 ? f.y : ((ks & 3) == 2 ? f.z : f.w));
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[t * 16 * HS + 4 * ks], bv, acc[t], 0, 0, 0);
    }
  }
  {
    const float bias = (src == 0 ? a.bih : a.bhh)[gate * H + j0 + (lane & 15)];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) g6[t * 6 * 256 + (wave * 16 + (lane >> 4) * 4 + e) * 16 + (lane & 15)] = acc[t][e] + bias;
  }
  __syncthreads();
  for (int idx = tid; idx < ROWS * 16; idx += 384) {
    const int r = idx >> 4, j = idx & 15, t = r >> 4, rr_ = r & 15;
    const int hr = b0 + r < a.B ? (rag ? stepc::hopv[b0 + r] : hop) : -1;
    if (hr >= 0) {
      const float* g = g6 + t * 6 * 256;
      const float gi_r = g[(0 * 16 + rr_) * 16 + j], gi_z = g[(1 * 16 + rr_) * 16 + j], gi_n = g[(2 * 16 + rr_) * 16 + j];
      const float gh_r = g[(3 * 16 + rr_) * 16 + j], gh_z = g[(4 * 16 + rr_) * 16 + j], gh_n = g[(5 * 16 + rr_) * 16 + j];
      const bsp::f32x2 rz = bsp::sigmoid2(bsp::f32x2{gi_r + gh_r, gi_z + gh_z});   // (packed forms, spec_math.hip.h: the same bits)
      const float rr = rz.x, zz = rz.y;
      const float nn = bsp::tanh2(bsp::splat2(bsp::fma(rr, gh_n, gi_n))).x;
      const float hp = hs[r * HS + j0 + j];
      const float hv = bsp::fma(zz, hp - nn, nn);
      ring_frame(a.h, b0 + r, rag ? ring_pos(a.h, hr) : ph, a.t)[j0 + j] = hv;
      if constexpr ((LINK & 1) != 0) glink::publish(a.link_out + (size_t)(b0 + r) * H + j0 + j, hv, hop + 1);
    }
  }
  }
}

template <int IN, int H, int RT = 1>
static __global__ __launch_bounds__(384) void gru_fused_kernel(const GruArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[RT * (16 * (IN + 2) + 16 * (H + 2) + 6 * 256)];
  gru_fused_body<IN, H, RT>(a, blockIdx.x, blockIdx.y, lds);
}

template <int IN, int H, int RT = 1, int LINK = 0>
struct GruOp {
  using Args = GruArgs;
  static constexpr int NTHR = 384;
  static constexpr int LDS_FLOATS = RT * (16 * (IN + 2) + 16 * (H + 2) + 6 * 256);
  static inline dim3 grid(const GruArgs& a) { const int rows = 16 * RT * (a.passes > 1 ? a.passes : 1); return dim3((a.B + rows - 1) / rows, H / 16); }
  static inline bhip::LaunchInfo info(const char* name, const GruArgs& a) {
    return bhip::LaunchInfo{name, 2.0 * a.B * (IN + H) * 3.0 * H, 4.0 * ((IN + H) * 3.0 * H + a.B * (IN + 2.0 * H))};
  }
  __device__ static __forceinline__ void run(const GruArgs& a, int bx, int by, float* lds) { gru_fused_body<IN, H, RT, false, LINK>(a, bx, by, lds); }
  template <bool RAG> __device__ static __forceinline__ void run_t(const GruArgs& a, int bx, int by, float* lds) { gru_fused_body<IN, H, RT, RAG, LINK>(a, bx, by, lds); }
};

// (RT = 2 as a launch of its own measured slower at 8192 streams -- 140 vs 113 us: 78 KB of LDS leave two workgroups
//  per CU where the one-tile form has four; inside the tick launch, where two per CU is the rule anyway, it is the
//  better one: 1024 streams 3.98 -> 4.04 M frames/s)
template <int IN, int H>
static inline void launch_gru(const char* name, const GruArgs& a, hipStream_t stream) {
  const dim3 grid = GruOp<IN, H>::grid(a);
  bhip::launch_site(GruOp<IN, H>::info(name, a), stream,
                    [&] { hipLaunchKernelGGL((gru_fused_kernel<IN, H>), grid, dim3(384), 0, stream, a); });
}

// ---------------------------------------------------------------------------------------------
struct AttnPvArgs {
  Ring scores;           // C = 384, n = H frames per step: scores of row (stream, hop), already scaled by 1/16
  const float* v;        // packed V tables, slot stride 384*256
  Ring out;              // C = 256, n = H
  const int* perm;       // [n_tiles][16] row indices (stream * H + hop) or -1
  const int* tile_slot;  // [n_tiles]
  const int* hop;
};

constexpr int kAttnPvLdsFloats = 16 * (B_KV_LEN + 2) + 16 + 2 * 16 * 32;
__device__ __forceinline__ void attn_pv_body(const AttnPvArgs& a, const int bx, const int by, float* __restrict__ lds) {
  constexpr int KL = B_KV_LEN, AS = KL + 2, NT = 32;
  float* es = lds;               // exp(s - max), [16][386]
  float* inv = lds + 16 * AS;    // [16]
  float* red = inv + 16;         // [2][16][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 1, wn = wave & 1;
  const int hop = stepc::step(a.hop);  // (checked after the V fragment loads are issued)
  const int slot = a.tile_slot[bx];
  if (slot < 0) return;
  const int n0 = by * NT;
  // B fragments of this wave: segment grp (keys 0..255 or 256..383), column tile (n0 + wn*16)/16
  float4 bf[16];
  {
    const float4* vp = reinterpret_cast<const float4*>(a.v + (size_t)slot * KL * B_HID) +
                       ((size_t)((n0 + wn * 16) >> 4) * (KL >> 4) + grp * 16) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) bf[kb] = vp[(size_t)((grp == 0 || kb < 8) ? kb : 0) * 64];
  }
  if (hop < 0) return;
  const int H = a.scores.n, pos_s = ring_pos(a.scores, hop), pos_o = ring_pos(a.out, hop);
  // softmax statistics, 4 rows per wavefront (MODEL_SPEC 4.4.2); all 24 score loads of the wave are
  // issued before the first reduction so that the rows do not pay one memory latency each
  {
    float v[4][6];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int b = a.perm[bx * 16 + wave * 4 + rr];
      const float* srow = b >= 0 ? ring_frame(a.scores, b / H, pos_s, b % H) : nullptr;
#pragma unroll
      for (int i = 0; i < 6; ++i) v[rr][i] = b >= 0 ? srow[lane + 64 * i] : 0.0f;
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = wave * 4 + rr;
      float mx = -__builtin_huge_valf();
#pragma unroll
      for (int i = 0; i < 6; ++i) mx = fmaxf(mx, v[rr][i]);
      mx = bsp::wmax64(mx);
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < 6; i += 2) {
        const bsp::f32x2 e = bsp::exp2(bsp::f32x2{v[rr][i] - mx, v[rr][i + 1] - mx});
        s = s + e.x; s = s + e.y;
        es[r * AS + lane + 64 * i] = e.x; es[r * AS + lane + 64 * (i + 1)] = e.y;
      }
      const float tot = bsp::wsum64(s);
      if (lane == 0) inv[r] = 1.0f / tot;
    }
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  {
    const float* ap = es + (lane & 15) * AS + grp * 256 + (lane >> 4);
    if (grp == 0) {
#pragma unroll
      for (int ks = 0; ks < 64; ++ks) {
        const float4 f = bf[ks >> 2];
        const float bv = (ks & 3) == 0 ? f.x : ((ks & 3) == 1 ? f.y : ((ks & 3) == 2 ? f.z : f.w));
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * ks], bv, acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) {
        const float4 f = bf[ks >> 2];
        const float bv = (ks & 3) == 0 ? f.x : ((ks & 3) == 1 ? f.y : ((ks & 3) == 2 ? f.z : f.w));
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * ks], bv, acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[(grp * 16 + (lane >> 4) * 4 + e) * NT + wn * 16 + (lane & 15)] = acc[e];
  __syncthreads();
  for (int idx = tid; idx < 16 * NT; idx += 256) {
    const int r = idx / NT, n = n0 + idx % NT;
    const int b = a.perm[bx * 16 + r];
    if (b < 0) continue;
    const float v = red[idx] + red[16 * NT + idx];  // segment 0 + segment 1 (MODEL_SPEC 2.2)
    ring_frame(a.out, b / H, pos_o, b % H)[n] = v * inv[r];
  }
}
static __global__ __launch_bounds__(256) void attn_pv_kernel(const AttnPvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kAttnPvLdsFloats];
  attn_pv_body(a, blockIdx.x, blockIdx.y, lds);
}
struct AttnPvOp {
  using Args = AttnPvArgs;
  static constexpr int NTHR = 256;
  static constexpr int LDS_FLOATS = kAttnPvLdsFloats;
  __device__ static __forceinline__ void run(const Args& a, int bx, int by, float* lds) { attn_pv_body(a, bx, by, lds); }
};
