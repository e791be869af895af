// conv_gemm.hip.h -- the workhorse kernel: causal (dilated / strided / polyphase-transposed)
// Conv1d, 1x1 conv and linear layers as one LDS-staged FP32-MFMA GEMM over (stream, frame) rows.
//
//   A[(b,t)][(j,c)] = pre( in_ring[b][(t+1)*STRIDE-1-(KSZ-1-j)*DIL][c] )       M = B*T rows
//   W[(j,c)][n]                                                                 K = KSZ*CIN
//   out[(b,t)][n]   = res + act( epi(acc) )                                     N = NOUT
//
// Numerics (MODEL_SPEC 2.2): an output is the ordered sum of P = ceil(K/256) segment results, each
// segment ONE k-ascending float32 FMA chain starting at 0.  v_mfma_f32_16x16x4_f32 is exactly such
// a chain (4 k per instruction, in k order); each segment lives in its own accumulator and the
// segments are added ((s0+s1)+s2)+..., so the kernel reproduces the scalar definition bit for bit
// while the segments run concurrently (separate wave groups, or interleaved accumulators).  16x16x4 is used instead of 32x32x2 because the dependent-accumulator latency per
// unit of K is 4x shorter (40 cycles per 4 k vs 64 per 2 k), which is what bounds the small-M
// layers of a per-hop network.
//
// Tiling: a workgroup of LM x LN wavefronts, each wavefront owning WM x WN MFMA tiles of 16x16.
// A tile rows are gathered from the ring in 16-byte pieces (a row's KC channels are contiguous),
// W tiles are read as coalesced float4 rows; both go through padded LDS so that the MFMA operand
// reads (ds_read_b32, lane -> [row l&15][k l>>4]) are bank-conflict free (A stride KC+2 -> banks
// 2*i + k distinct over a 32-lane group).  W never touches LDS: see "PRE-PACKED" below.
#pragma once
#include "meas_env.h"
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "engine.h"
#include "ring.h"
#include "spec_math.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// GEMM weights are stored on the device PRE-PACKED in MFMA B-fragment order (done once on the host
// at Read*Parameters, csrc/common.hip pack_kn): for column tile nt (16 columns) and k-block kb
// (16 reduction indices = four 16x16x4 steps) one 1 KiB record of 64 lanes x float4, where lane
// (j = l&15, kq = l>>4) holds W[kb*16 + 4e + kq][nt*16 + j] for e = 0..3.  A wavefront therefore
// fetches the B operands of four MFMA steps with ONE fully coalesced 16-byte-per-lane load straight
// into registers: no LDS staging, no transposition, no bank conflicts for W.
__host__ __device__ inline size_t packed_w_offset(int K, int kk, int n) {
  return ((((size_t)(n >> 4) * (K >> 4) + (kk >> 4)) * 64 + ((kk & 3) << 4) + (n & 15)) << 2) + ((kk & 15) >> 2);
}

enum { PRE_NONE = 0, PRE_LRELU = 1 };
enum { ACT_NONE = 0, ACT_GELU = 1 };
enum { EPI_BIAS = 0, EPI_SCALE = 1, EPI_ROWSCALE = 2 };

struct ConvArgs {
  Ring in, out, res;
  const float* w;
  const float* bias;
  const int* hop;
  int B;
  float scale;            // EPI_SCALE
  const float* rowscale;  // EPI_ROWSCALE: one factor per stream
  // grouped mode (rows gathered by stream index, W chosen per M-tile)
  const int* perm;       // [n_tiles][MT] stream index or -1
  const int* tile_slot;  // [n_tiles] weight slot or -1 (tile unused)
  size_t w_slot_stride;  // floats between weight slots
  int rel_shift;         // added to every input frame offset (-1: read the previous hop's frame)
};

__device__ __forceinline__ void globalize(ConvArgs& a) {
  globalize(a.in); globalize(a.out); globalize(a.res);
  a.w = as_global(a.w); a.bias = as_global(a.bias); a.hop = as_global(a.hop); a.rowscale = as_global(a.rowscale);
  a.perm = as_global(a.perm); a.tile_slot = as_global(a.tile_slot);
}

template <int CIN_, int NOUT_, int KSZ_, int STRIDE_, int DIL_, int T_, int PRE_, int ACT_,
          int EPI_, bool RES_, bool GROUPED_ = false>
struct Layer {
  static constexpr int CIN = CIN_, NOUT = NOUT_, KSZ = KSZ_, STRIDE = STRIDE_, DIL = DIL_, T = T_;
  static constexpr int PRE = PRE_, ACT = ACT_, EPI = EPI_;
  static constexpr bool RES = RES_, GROUPED = GROUPED_;
  static constexpr int K = KSZ * CIN;
  static constexpr int SEG = 256;                      // MODEL_SPEC 2.2: reduction segment length
  static constexpr int P = (K + SEG - 1) / SEG;        // segments per output
  // staged k-chunk (never straddles a tap or a segment): 64 for the throughput tiling; the few-row
  // tiling takes 128 where the channel count allows (half the barriers on its latency-bound chain)
  static constexpr int KC = CIN >= 64 ? 64 : CIN;
  static constexpr int KC_FEW = CIN % 128 == 0 ? 128 : KC;
  static_assert(CIN % KC == 0 && KC % 4 == 0 && SEG % KC == 0 && K % KC == 0, "channel chunking");
  static_assert(CIN % KC_FEW == 0 && SEG % KC_FEW == 0 && K % KC_FEW == 0, "channel chunking (few-row tiling)");
};

// A workgroup is LK "k-groups" of LM x LN wavefronts; each wavefront owns WM x WN MFMA tiles.
// LK == 1: one group walks all segments into separate accumulators (throughput layers);
// LK == P: group g owns segment g, partial tiles are combined through LDS in segment order
//          (latency layers: the dependent MFMA chain is one segment, not K, long).
// KFEW_: the few-row tiling (MT == 16) stages 128-wide k-chunks where the layer allows (half the barriers on a
// latency-bound chain); false = 64-wide chunks, which needs ~50 fewer registers (tick launches want occupancy instead)
template <int WM_, int WN_, int LM_, int LN_, int LK_ = 1, bool KFEW_ = true>
struct TileCfg {
  static constexpr int WM = WM_, WN = WN_, LM = LM_, LN = LN_, LK = LK_;
  static constexpr bool KFEW = KFEW_;
  static constexpr int MT = 16 * WM * LM, NT = 16 * WN * LN;
  static constexpr int GTHR = 64 * LM * LN;  // threads per k-group
  static constexpr int NTHR = GTHR * LK;
};

// LDS the body needs (floats): the staged A chunk of every k-group, re-used for the segment combine
template <class L, class TC>
constexpr int conv_lds_floats() {
  constexpr int KC = (TC::MT == 16 && TC::KFEW) ? L::KC_FEW : L::KC;
  constexpr int stage = TC::MT * (KC + 2) * TC::LK, red = TC::LK > 1 ? L::P * TC::MT * TC::NT : 0;
  return stage > red ? stage : red;
}

// The kernel body takes its workgroup coordinates and its LDS as arguments so that independent layers can share
// one launch (fuse.hip.h); conv_gemm_kernel below is the plain one-layer launch.
template <class L, class TC>
__device__ __forceinline__ void conv_gemm_body(const ConvArgs& a, const int bx, const int by, float* __restrict__ lds) {
  constexpr int KC = (TC::MT == 16 && TC::KFEW) ? L::KC_FEW : L::KC;
  constexpr int NCHUNK = L::K / KC, CHUNKS_PER_SEG = L::SEG / KC;
  constexpr int AS = KC + 2, MT = TC::MT, NT = TC::NT, GTHR = TC::GTHR, LK = TC::LK;
  constexpr int P = L::P;
  static_assert(LK == 1 || LK == P, "k-groups: one group for all segments, or one group per segment");
  constexpr int PG = (P + LK - 1) / LK;  // segments (accumulator sets) per group
  constexpr int A_F4_PER_ROW = KC / 4;
  constexpr int A_SLOTS = (MT * A_F4_PER_ROW + GTHR - 1) / GTHR;
  constexpr int KB = KC / 16;               // packed k-blocks per chunk
  constexpr int STAGE_FLOATS = MT * AS;     // per k-group (only A goes through LDS)
  static_assert(KC % 16 == 0 && L::K % 16 == 0, "packed weights need K in blocks of 16");
  constexpr int RED_FLOATS = LK > 1 ? P * MT * NT : 0;
  constexpr int LDS_FLOATS = STAGE_FLOATS * LK > RED_FLOATS ? STAGE_FLOATS * LK : RED_FLOATS;
  static_assert(L::NOUT % NT == 0, "N tile must divide NOUT");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
  static_assert(LDS_FLOATS == conv_lds_floats<L, TC>(), "LDS size");

  const int tid = threadIdx.x;
  const int grp = tid / GTHR, gtid = tid % GTHR;
  const int lane = gtid & 63, wave = gtid >> 6;
  const int wave_m = (wave / TC::LN) * (16 * TC::WM), wave_n = (wave % TC::LN) * (16 * TC::WN);
  const int m0 = bx * MT, n0 = by * NT;
  const int M = a.B * L::T;
  float* As = lds + grp * STAGE_FLOATS;

  const float* wbase = a.w;
  if constexpr (L::GROUPED) {
    const int slot = a.tile_slot[bx];
    if (slot < 0) return;
    wbase += (size_t)slot * a.w_slot_stride;
  }

  const int hop = stepc::step(a.hop);   // (-1: this stage has no step this tick -- pipeline fill / drain of the batch's tick mode; checked
                            //  below, AFTER the first weight loads are issued, so that it costs the chain no extra latency)
  const int pos_in = ring_pos(a.in, hop);

  // per-thread A staging slots: which (stream, t) row and which 16-byte piece
  int a_b[A_SLOTS], a_t[A_SLOTS];
#pragma unroll
  for (int s = 0; s < A_SLOTS; ++s) {
    const int idx = gtid + s * GTHR;
    const int r = idx / A_F4_PER_ROW;
    int b = -1, t = 0;
    if (r < MT) {
      if constexpr (L::GROUPED) {  // grouped rows: perm holds row indices (stream * T + frame) or -1
        const int m = a.perm[bx * MT + r];
        if (m >= 0) { b = m / L::T; t = m % L::T; }
      } else {
        const int m = m0 + r;
        if (m < M) { b = m / L::T; t = m % L::T; }
      }
    }
    a_b[s] = b;
    a_t[s] = t;
  }

  f32x4 acc[PG][TC::WM][TC::WN];
#pragma unroll
  for (int g = 0; g < PG; ++g)
#pragma unroll
    for (int i = 0; i < TC::WM; ++i)
#pragma unroll
      for (int j = 0; j < TC::WN; ++j) acc[g][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 areg[A_SLOTS];
  float4 bnext[TC::WN][KB], bcur[TC::WN][KB];
  const float4* wfrag = reinterpret_cast<const float4*>(wbase) + (size_t)((n0 + wave_n) >> 4) * (L::K >> 4) * 64 + lane;
  // chunk `it` of this group: LK == 1 -> global chunk it; LK == P -> chunk it of segment grp
  // weights first: their addresses do not depend on the step counter, so the loads are in flight
  // while the scalar counter load that the ring addresses of the A operand wait for completes
  auto load_w = [&](int it) {
    const int ch = LK == 1 ? it : grp * CHUNKS_PER_SEG + it;
    const bool live = ch < NCHUNK;
    const int kk0 = ch * KC;
#pragma unroll
    for (int jn = 0; jn < TC::WN; ++jn)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int kbg = live ? (kk0 >> 4) + kb : 0;
        const float4 f = wfrag[((size_t)jn * (L::K >> 4) + kbg) * 64];
        bnext[jn][kb] = live ? f : make_float4(0.f, 0.f, 0.f, 0.f);
      }
  };
  auto load_a = [&](int it) {
    const int ch = LK == 1 ? it : grp * CHUNKS_PER_SEG + it;
    const bool live = ch < NCHUNK;
    const int kk0 = ch * KC;
    const int j = kk0 / L::CIN, c0 = kk0 % L::CIN;
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
      const int idx = gtid + s * GTHR;
      const int r = idx / A_F4_PER_ROW, q = idx % A_F4_PER_ROW;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && r < MT && a_b[s] >= 0) {
        const int rel = (a_t[s] + 1) * L::STRIDE - 1 - (L::KSZ - 1 - j) * L::DIL + a.rel_shift;
        v = *reinterpret_cast<const float4*>(ring_frame(a.in, a_b[s], pos_in, rel) + c0 + 4 * q);
      }
      areg[s] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int s = 0; s < A_SLOTS; ++s) {
      const int idx = gtid + s * GTHR;
      const int r = idx / A_F4_PER_ROW, q = idx % A_F4_PER_ROW;
      if (r < MT) {
        float4 v = areg[s];
        if constexpr (L::PRE == PRE_LRELU) {
          v.x = bsp::lrelu(v.x); v.y = bsp::lrelu(v.y); v.z = bsp::lrelu(v.z); v.w = bsp::lrelu(v.w);
        }
        float2* dst = reinterpret_cast<float2*>(&As[r * AS + 4 * q]);
        dst[0] = make_float2(v.x, v.y);
        dst[1] = make_float2(v.z, v.w);
      }
    }
#pragma unroll
    for (int jn = 0; jn < TC::WN; ++jn)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) bcur[jn][kb] = bnext[jn][kb];
  };

  auto load_chunk = [&](int it) { load_w(it); load_a(it); };
  constexpr int N_IT = LK == 1 ? NCHUNK : (CHUNKS_PER_SEG < NCHUNK ? CHUNKS_PER_SEG : NCHUNK);
  load_w(0);
  if (hop < 0) return;
  load_a(0);

  // ---- few-row tiling: the epilogue's operands (bias, residual, row scale) and output addresses are
  // fetched now, behind the first chunk's loads, instead of after the MFMA chain (one global-memory
  // latency less on the dependent path of a latency-bound launch)
  constexpr bool PF = TC::MT == 16;
  constexpr int E_SLOTS = LK == 1 ? 4 * TC::WM * TC::WN : (MT * NT + TC::NTHR - 1) / TC::NTHR;
  int pf_dst[PF ? E_SLOTS : 1];
  float pf_bias[PF ? E_SLOTS : 1], pf_res[PF ? E_SLOTS : 1], pf_rs[PF ? E_SLOTS : 1];
  auto elem = [&](int slot, int& r, int& n) -> bool {  // output element `slot` of this thread
    if constexpr (LK == 1) {
      const int e = slot & 3, jn = (slot >> 2) % TC::WN, i = (slot >> 2) / TC::WN;
      r = wave_m + i * 16 + (lane >> 4) * 4 + e;  // D layout of 16x16x4: row = (lane>>4)*4 + reg, col = lane&15
      n = n0 + wave_n + jn * 16 + (lane & 15);
      return true;
    } else {
      const int idx = tid + slot * TC::NTHR;
      r = idx / NT;
      n = n0 + idx % NT;
      return idx < MT * NT;
    }
  };
  if constexpr (PF) {
    // addresses first (scalar ring arithmetic, integer division included), then every load back to back: a load
    // issued in the middle of the address code made the compiler wait for ALL outstanding loads (s_waitcnt vmcnt(0))
    const int pos_out = ring_pos(a.out, hop), R_out = a.out.n * a.out.m;
    int pos_res = 0, R_res = 0;
    if constexpr (L::RES) { pos_res = ring_pos(a.res, hop); R_res = a.res.n * a.res.m; }
    int src_res[E_SLOTS], src_rs[E_SLOTS], src_n[E_SLOTS];
#pragma unroll
    for (int sl = 0; sl < E_SLOTS; ++sl) {
      int r, n, b = -1, t = 0;
      pf_dst[sl] = -1; src_res[sl] = 0; src_rs[sl] = 0; src_n[sl] = 0;
      if (elem(sl, r, n)) {
        if constexpr (L::GROUPED) {
          const int m = a.perm[bx * MT + r];
          if (m >= 0) { b = m / L::T; t = m % L::T; }
        } else {
          const int m = m0 + r;
          if (m < M) { b = m / L::T; t = m % L::T; }
        }
      }
      if (b >= 0) {
        pf_dst[sl] = (b * R_out + pos_out) * a.out.C + t * L::NOUT + n;
        src_n[sl] = n;
        src_rs[sl] = b * L::T + t;
        if constexpr (L::RES) src_res[sl] = (b * R_res + pos_res) * a.res.C + t * L::NOUT + n;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sl = 0; sl < E_SLOTS; ++sl) {
      const bool live = pf_dst[sl] >= 0;
      pf_bias[sl] = 0.f; pf_res[sl] = 0.f; pf_rs[sl] = 1.f;
      if constexpr (L::EPI == EPI_BIAS) pf_bias[sl] = live ? a.bias[src_n[sl]] : 0.f;
      if constexpr (L::EPI == EPI_ROWSCALE) pf_rs[sl] = live ? a.rowscale[src_rs[sl]] : 1.f;
      if constexpr (L::RES) pf_res[sl] = live ? a.res.base[src_res[sl]] : 0.f;
    }
  }
  auto finish_pf = [&](int sl, float v) {
    if (pf_dst[sl] < 0) return;
    if constexpr (L::EPI == EPI_BIAS) v = v + pf_bias[sl];
    if constexpr (L::EPI == EPI_SCALE) v = v * a.scale;
    if constexpr (L::EPI == EPI_ROWSCALE) v = v * pf_rs[sl];
    if constexpr (L::ACT == ACT_GELU) v = bsp::gelu(v);
    if constexpr (L::RES) v = pf_res[sl] + v;
    a.out.base[pf_dst[sl]] = v;
  };

#pragma unroll 1
  for (int it = 0; it < N_IT; ++it) {
    store_chunk();
    __syncthreads();
    if (it + 1 < N_IT) load_chunk(it + 1);  // global loads fly while the MFMAs below run
    const int g = LK == 1 ? (it / CHUNKS_PER_SEG) : 0;
    // accumulator set must be a compile-time index: dispatch over PG
#pragma unroll
    for (int gs = 0; gs < PG; ++gs) {
      if (gs != g) continue;
      if constexpr (TC::MT == 16) {
        // few-row tiling: one wavefront per SIMD and a dependent MFMA chain, so what counts is the chain's latency.
        // Left alone the compiler puts each ds_read next to the MFMA that uses it (ds_read, wait, 2 MFMAs, ...) and
        // the chain pays the LDS latency after every second MFMA; fetch the chunk's whole A operand first instead.
        float aop[KC / 4];
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) aop[ks] = As[(wave_m + (lane & 15)) * AS + ks * 4 + (lane >> 4)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
#pragma unroll
          for (int jn = 0; jn < TC::WN; ++jn) {
            const float4 f = bcur[jn][ks >> 2];
            const float bvv = (ks & 3) == 0 ? f.x : ((ks & 3) == 1 ? f.y : ((ks & 3) == 2 ? f.z : f.w));
            acc[gs][0][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[ks], bvv, acc[gs][0][jn], 0, 0, 0);
          }
        }
        continue;
      }
#pragma unroll
      for (int ks = 0; ks < KC / 4; ++ks) {
        float av[TC::WM], bv[TC::WN];
#pragma unroll
        for (int i = 0; i < TC::WM; ++i) av[i] = As[(wave_m + i * 16 + (lane & 15)) * AS + ks * 4 + (lane >> 4)];
#pragma unroll
        for (int jn = 0; jn < TC::WN; ++jn) {
          const float4 f = bcur[jn][ks >> 2];
          bv[jn] = (ks & 3) == 0 ? f.x : ((ks & 3) == 1 ? f.y : ((ks & 3) == 2 ? f.z : f.w));
        }
#pragma unroll
        for (int i = 0; i < TC::WM; ++i)
#pragma unroll
          for (int jn = 0; jn < TC::WN; ++jn)
            acc[gs][i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[jn], acc[gs][i][jn], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- combine segments in ascending order (MODEL_SPEC 2.2)
  const int pos_out = ring_pos(a.out, hop);
  const int R_out = a.out.n * a.out.m;
  int pos_res = 0, R_res = 0;
  if constexpr (L::RES) { pos_res = ring_pos(a.res, hop); R_res = a.res.n * a.res.m; }

  auto finish = [&](int r, int n, float v) {  // r: row in tile, n: absolute column
    int b = -1, t = 0;
    if constexpr (L::GROUPED) {
      const int m = a.perm[bx * MT + r];
      if (m >= 0) { b = m / L::T; t = m % L::T; }
    } else {
      const int m = m0 + r;
      if (m < M) { b = m / L::T; t = m % L::T; }
    }
    if (b < 0) return;
    if constexpr (L::EPI == EPI_BIAS) v = v + a.bias[n];
    if constexpr (L::EPI == EPI_SCALE) v = v * a.scale;
    if constexpr (L::EPI == EPI_ROWSCALE) v = v * a.rowscale[b * L::T + t];
    if constexpr (L::ACT == ACT_GELU) v = bsp::gelu(v);
    if constexpr (L::RES) v = a.res.base[((size_t)b * R_res + pos_res) * a.res.C + (size_t)t * L::NOUT + n] + v;
    a.out.base[((size_t)b * R_out + pos_out) * a.out.C + (size_t)t * L::NOUT + n] = v;
  };

  if constexpr (LK == 1) {
#pragma unroll
    for (int sl = 0; sl < 4 * TC::WM * TC::WN; ++sl) {
      const int e = sl & 3, jn = (sl >> 2) % TC::WN, i = (sl >> 2) / TC::WN;
      float v = acc[0][i][jn][e];
#pragma unroll
      for (int g = 1; g < PG; ++g) v = v + acc[g][i][jn][e];
      if constexpr (PF) {
        finish_pf(sl, v);
      } else {
        int r, n;
        elem(sl, r, n);
        finish(r, n, v);
      }
    }
  } else {
    float* red = lds;  // [P][MT][NT]; staging buffers are dead after the last barrier above
#pragma unroll
    for (int i = 0; i < TC::WM; ++i)
#pragma unroll
      for (int jn = 0; jn < TC::WN; ++jn)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          red[(grp * MT + wave_m + i * 16 + (lane >> 4) * 4 + e) * NT + wave_n + jn * 16 + (lane & 15)] = acc[0][i][jn][e];
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < E_SLOTS; ++sl) {
      const int idx = tid + sl * TC::NTHR;
      if (idx >= MT * NT) break;
      float v = red[idx];
#pragma unroll
      for (int g = 1; g < P; ++g) v = v + red[g * MT * NT + idx];
      if constexpr (PF) finish_pf(sl, v);
      else finish(idx / NT, n0 + idx % NT, v);
    }
  }
}

template <class L, class TC>
__global__ __launch_bounds__(TC::NTHR) void conv_gemm_kernel(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[conv_lds_floats<L, TC>()];
  conv_gemm_body<L, TC>(a, blockIdx.x, blockIdx.y, lds);
}

// launch geometry + algorithmic work of one layer, shared by the single and the paired launchers
template <class L, class TC>
struct ConvOp {
  using Args = ConvArgs;
  static constexpr int NTHR = TC::NTHR;
  static constexpr int LDS_FLOATS = conv_lds_floats<L, TC>();
  static inline dim3 grid(const ConvArgs& a, int n_group_tiles = 0) {
    dim3 g;
    g.x = L::GROUPED ? n_group_tiles : (a.B * L::T + TC::MT - 1) / TC::MT;
    g.y = L::NOUT / TC::NT;
    return g;
  }
  // 2*M*K*N flops; bytes = weights once + A rows once + output once
  static inline bhip::LaunchInfo info(const char* name, const ConvArgs& a) {
    const double M = (double)a.B * L::T, K = (double)L::KSZ * L::CIN, N = L::NOUT;
    const double in_rows = (double)a.B * (L::T * L::STRIDE + (L::KSZ - 1) * L::DIL - (L::STRIDE - 1));
    return bhip::LaunchInfo{name, 2.0 * M * K * N, 4.0 * (K * N + in_rows * L::CIN + M * N * (L::RES ? 2 : 1))};
  }
  // relative time of one workgroup: a fixed part (launch, operand latency) + its dependent MFMA chain (microseconds, roughly)
  static constexpr double wg_cost() { return 2.0 + (double)(L::K / 4) * TC::WM * TC::WN / TC::LK * (40.0 / 2400.0); }
  __device__ static __forceinline__ void run(const ConvArgs& a, int bx, int by, float* lds) { conv_gemm_body<L, TC>(a, bx, by, lds); }
};

template <class L, class TC>
static inline void launch_conv(const char* name, const ConvArgs& a, int n_group_tiles, hipStream_t stream) {
  const dim3 grid = ConvOp<L, TC>::grid(a, n_group_tiles);
  bhip::launch_site(ConvOp<L, TC>::info(name, a), stream,
                    [&] { hipLaunchKernelGGL((conv_gemm_kernel<L, TC>), grid, dim3(TC::NTHR), 0, stream, a); });
}

// ---- launch helpers shared by the modules ------------------------------------------------------
template <class L> using TLat = TileCfg<1, 1, 1, 2, L::P>;  // 16 x 32 tile, one 2-wave k-group per segment
#ifndef BEATRICE_TL
#define BEATRICE_TL 2, 2, 2, 2, 1
#endif
using TL = TileCfg<BEATRICE_TL>;                              // 64 x 64 tile, segments interleaved (BEATRICE_TL: A/B builds)

#include "gemv.hip.h"  // (needs everything above; defines gemv::launch)

// Layers with few rows are bound by the dependent MFMA chain and by how many CUs get a tile: use
// small tiles and one k-group per segment.  Layers with many rows are throughput-bound: 32 rows x the layer's width per
// workgroup with the A operand streamed by reduction segment (rc::conv_rows_body), or 64 x 64 tiles where that body does not apply.  A handful of rows
// (the 1-stream C-ABI): one FMA chain per lane instead of a mostly empty MFMA tile (gemv.hip.h).
// many-row layers (defined in rowchain.hip.h, which every user of launch_auto includes through chain_layers.hip.h)
template <class L>
static inline void launch_many_rows(const char* name, const ConvArgs& a, hipStream_t s);
template <class L>
static inline void launch_auto(const char* name, const ConvArgs& a, hipStream_t s) {
  // (a single-shot variant that issues every load up front -- tools/microbench/lat_gemm.hip.h -- measured
  //  slower on MI355X: 5.5 vs 4.9 us for a 256->256 linear at 256 rows, profiles/r01_notes.md; overlapping
  //  the chunked loads with the MFMA chain wins)
  // measured at one stream (us per launch, gemv vs few-row MFMA tiling): 256->256 linear 7.2 vs 5.0, k4 s2 convs 8.1-8.7 vs
  // 6.5-6.8 -- a single 256-long chain per lane pays the same serial memory round trips as the MFMA kernel and has no
  // second wavefront to hide them -- but layers with three or more reduction segments win (the segments run as
  // parallel lanes): k5 residual blocks 5.8 vs 7.4, k3 dilated convs 5.5-6.0 vs 6.6-6.9, the pitch estimator's first layer 6.7 vs 7.8
  static const bool no_gemv = bhip::meas_env("BEATRICE_HIP_NO_GEMV") != nullptr;  // A/B switch for measurements
  if constexpr (!L::GROUPED && L::P >= 3 && L::T == 1) {
    if (!no_gemv && a.B <= 2) { gemv::launch<L>(name, a, s); return; }
  }
  if (a.B * L::T <= 2048) launch_conv<L, TLat<L>>(name, a, 0, s);
  else launch_many_rows<L>(name, a, s);
}

static inline ConvArgs conv_args(const Ring& in, const Ring& out, const float* w, const float* b, const int* hop, int B) {
  ConvArgs a{};
  a.in = in; a.out = out; a.res = in;
  a.w = w; a.bias = b; a.hop = hop; a.B = B;
  a.scale = 1.0f; a.rowscale = nullptr; a.perm = nullptr; a.tile_slot = nullptr; a.w_slot_stride = 0;
  a.rel_shift = 0;
  return a;
}
