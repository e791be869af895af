// rowchain.hip.h -- row-local chains of layers as ONE workgroup per 16 rows (stream-stationary kernels).
//
// In the conditioned blocks of the waveform generator (and the tail of the content encoder) every layer has one
// frame per stream-hop, so a layer's output row depends only on the same row of its input: conv-k3 -> 1x1 + residual
// and q -> q.K^T -> softmax.V -> output linear + residual need no exchange between rows.  A workgroup that owns 16
// rows can therefore run the whole chain with the activations in LDS and only the weights streaming past
// (pre-packed MFMA B fragments, conv_gemm.hip.h), instead of one launch per layer with every layer's rows spread over
// the chip.  Per layer this trades parallelism (16 workgroups for 256 streams) for density (~4000-5000 back-to-back
// MFMAs per workgroup and no launch / fill / drain per layer): the right trade inside the tick pipeline (tick.hip.h),
// where ~40 stages of different steps share a launch and the chip is filled by stages, not by one layer's tiles.
// In the in-order chain the per-layer launches win; BEATRICE_HIP_ROWCHAIN=1 runs these kernels
// there for measurements and for the parity tests.
//
// Numerics are those of the per-layer kernels, operation for operation (MODEL_SPEC 2.2: every 256-long reduction
// segment one k-ascending MFMA chain, segments added in order, then bias / scale / activation / residual in the same
// order as conv_gemm's epilogue; softmax as attn_pv_kernel), so results are bit-identical.
#pragma once
#include "meas_env.h"
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "conv_gemm.hip.h"
#include "kernels_misc.hip.h"
#include "ring.h"
#include "spec_math.hip.h"

// Measurement aid (tools/debug/build_variant.sh ... -DABL_WEIGHTS_HOT): every weight-fragment prefetch re-reads the segment's
// first k-blocks, so the weight stream hits the CU's L1 -- results are WRONG; the build exists to time the launch without
// its L2 / Infinity Cache weight traffic.  Never defined in the product build.
#ifdef ABL_WEIGHTS_HOT
#define ABL_KB(x) ((x) & 1)
#else
#define ABL_KB(x) (x)
#endif

// per-phase cycle stamps of the attention bodies for tools/microbench/blockb_timing.hip (never defined in the product build)
#ifdef RC_TIMING
__device__ unsigned long long* g_rc_stamps;   // [workgroups][16]
#define RC_STAMP(i) do { if (threadIdx.x == 0) g_rc_stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define RC_STAMP(i) do { } while (0)
#endif

namespace rc {

constexpr int NTHR = 512, NWAVE = 8;
// LDS tiles of activations [16 rows][K] hold every 16-long k-block K-PERMUTED: logical k = 16 b + 4 j + q sits at position
// 16 b + 4 q + j (kpos), so that the four values a lane feeds to the four MFMAs of a k-block (k = q, 4 + q, 8 + q, 12 + q for the
// lane's q = l >> 4) are ONE ds_read_b128 instead of four ds_read_b32 (round 5; profiles/r05_notes.md).  Row stride = K + 8
// floats: stride / 4 = 2 (mod 16) makes the 16-byte slots of each of ds_read_b128's four lane groups ({0-3, 12-15, 20-27}, ...:
// rows 0-3 and 12-15 at one q, rows 4-11 at q + 1) all distinct -- conflict-free (with K + 4 rows 11 and 12 would collide).
#ifndef RC_KPERM
#define RC_KPERM 1   // A/B build switch: 0 = tiles in logical k order, four ds_read_b32 per k-block (rounds 2-4), row stride K + 2
#endif
constexpr bool KPERM = RC_KPERM != 0;
constexpr int AS = B_HID + (KPERM ? 8 : 2);        // LDS row stride of a 256-channel tile
constexpr int TILE = 16 * AS;        // floats of one [16][256] tile
constexpr int SS = B_KV_LEN + (KPERM ? 8 : 2);     // row stride of the score tile
constexpr int STILE = 16 * SS;
__host__ __device__ constexpr int kpos(int k) { return KPERM ? ((k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3)) : k; }
constexpr int KSTEP = KPERM ? 4 : 1;   // distance of columns n, n + 1 of a lane's four inside a tile row
// Which 16-byte piece (row r, floats 4 q .. 4 q + 3) of a [rows][256] tile a thread gathers: wave-load W (64 threads) and lane ln.
// Logical tiles: one row per wave-load.  k-permuted tiles: FOUR rows x 16 pieces per wave-load -- a piece is stored as four
// ds_write_b32 at p, p + 4, p + 8, p + 12, and 64 pieces of ONE row would hit 8 of the 32 banks (4-way, twice the time); rows
// are 8 banks apart, so four rows x 16 pieces are 2-way, which costs a ds_write_b32 nothing (MI355X_MICROARCH.md, LDS)
#ifndef RC_WMAP
#define RC_WMAP RC_KPERM
#endif
template <int ROWGROUPS /* rows / 4 */>
__device__ __forceinline__ void piece_of(const int W, const int ln, int* r, int* q) {
  if constexpr (RC_WMAP != 0) { *r = 4 * (W % ROWGROUPS) + (ln >> 4); *q = 16 * (W / ROWGROUPS) + (ln & 15); }
  else { *r = W; *q = ln; }
}
// a lane's operand base inside a tile row: its q = l >> 4
__device__ __forceinline__ int lane_koff(const int lane) { return KPERM ? 4 * (lane >> 4) : (lane >> 4); }
// the four values k = q, 4 + q, 8 + q, 12 + q of k-block kb for this lane (p = row base + lane_koff)
__device__ __forceinline__ float4 lds_kblock(const float* __restrict__ p, const int kb) {
  if constexpr (KPERM) return *reinterpret_cast<const float4*>(p + kb * 16);
  else { const float* q = p + kb * 16; return make_float4(q[0], q[4], q[8], q[12]); }
}
// Results come TRANSPOSED out of the matrix unit: the weights are the MFMA's A operand, the activations its B operand (bit for
// bit the same chains: tools/microbench/mfma_swap.hip), so a lane owns FOUR CONSECUTIVE COLUMNS n0 .. n0 + 3 of ONE row (row =
// l & 15, n0 = 16 tile + 4 (l >> 4)) -- one row lookup, one 16-byte bias / residual load and one 16-byte store per accumulator,
// where the other orientation (four rows of one column) paid all of that per element.  In a k-permuted LDS tile those four
// columns are the elements p, p + 4, p + 8, p + 12 with p = kpos(n0).
__device__ __forceinline__ void store_perm4(float* __restrict__ row, const int n0, const float v0, const float v1, const float v2, const float v3) {
  if constexpr (KPERM) {
    float* d = row + kpos(n0);
    d[0] = v0; d[4] = v1; d[8] = v2; d[12] = v3;
  } else {   // (row stride K + 2: 8-byte aligned)
    float2* d = reinterpret_cast<float2*>(row + n0);
    d[0] = make_float2(v0, v1); d[1] = make_float2(v2, v3);
  }
}

// One reduction segment (KB k-blocks of 16) for CG column tiles of this wavefront: acc[c] += (A[16 x 16 KB] . W)^T.
// `a` = this lane's activation base in the k-permuted LDS tile (row l & 15, position 4 (l >> 4) of the segment's first
// k-block: 16-byte aligned); `wf` = this lane's float4 of column tile 0 / k-block 0 of the segment; consecutive column tiles
// of the wavefront are `tile_stride` float4 apart.  acc[c][e] = row l & 15, column 16 tile_c + 4 (l >> 4) + e.
// B fragments are fetched four k-blocks ahead of their use (an L2 round trip is ~200-500 cycles, a k-block is
// 4 CG MFMAs = 128 CG cycles of the SIMD's matrix pipe; eight ahead measured 2 % better than four at two column tiles).
// Software pipeline, pinned: left alone the compiler SINKS the prefetch loads to just in front of their first use (to save
// registers), so that every k-block waits a full L2 round trip -- the ISA of round 2's build showed `global_load ... ;
// s_waitcnt vmcnt(0); v_mfma` throughout, and a wavefront's MFMA density was ~27 %.  A __builtin_amdgcn_sched_barrier(0)
// between "issue the loads of k-block i + D (weights) and i + 1 (A operand)" and "the MFMAs of k-block i" forbids that: loads
// may still float among the MFMAs of the PREVIOUS k-block, never behind the ones that were meant to hide them.
#ifndef RC_PIN_PIPELINE
#define RC_PIN_PIPELINE 1
#endif
__device__ __forceinline__ void pin_pipeline() {
#if RC_PIN_PIPELINE
  asm volatile("" ::: "memory");        // IR level: optimisation passes hoist invariant weight loads across a bare sched_barrier call
  __builtin_amdgcn_sched_barrier(0);    // machine scheduler: nothing crosses
#endif
}
#ifndef RC_PREFETCH
#define RC_PREFETCH 8
#endif
// prefetch depth of a segment in k-blocks (registers: D * CG float4)
template <int RT, int CG> __host__ __device__ constexpr int seg_depth() { return RT == 1 ? (CG <= 2 ? RC_PREFETCH : (CG <= 3 ? 4 : 2)) : (RT * CG <= 2 ? 8 : 4); }
// the first seg_depth() k-blocks of a segment's weight fragments, requested by the CALLER ahead of time (a body's first segment: the
// request goes out before the body gathers its rows, so that the two round trips overlap instead of following each other)
template <int RT, int CG> struct SegHead { float4 v[seg_depth<RT, CG>()][CG]; };
template <int RT, int CG>
__device__ __forceinline__ void seg_head_load(SegHead<RT, CG>& h, const float4* const (&wf)[CG]) {
#pragma unroll
  for (int d = 0; d < seg_depth<RT, CG>(); ++d)
#pragma unroll
    for (int c = 0; c < CG; ++c) h.v[d][c] = wf[c][(size_t)d * 64];
}
template <int CG, int KB, bool HEAD = false>
__device__ __forceinline__ void mma_segment_p(f32x4 (&acc)[CG], const float* __restrict__ a, const float4* const (&wf)[CG], const SegHead<1, CG>* head = nullptr) {
  constexpr int D = seg_depth<1, CG>();
  static_assert(KB % D == 0, "segment length");
  float4 bq[D][CG];
  pin_pipeline();   // (the segment's first loads stay behind what precedes it: hoisted over an epilogue they only add register pressure)
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int c = 0; c < CG; ++c) { if constexpr (HEAD) bq[d][c] = head->v[d][c]; else bq[d][c] = wf[c][(size_t)d * 64]; }
  float4 an = lds_kblock(a, 0);   // activations of the k-block to come (one k-block ahead: LDS latency behind 4 CG MFMAs), one ds_read_b128
#pragma unroll
  for (int kb = 0; kb < KB; kb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      float4 cur[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) cur[c] = bq[d][c];
      const float4 x = an;
      if (kb + d + D < KB) {
#pragma unroll
        for (int c = 0; c < CG; ++c) bq[d][c] = wf[c][(size_t)ABL_KB(kb + d + D) * 64];
      }
      if (kb + d + 1 < KB) an = lds_kblock(a, kb + d + 1);
      pin_pipeline();
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[c].x, x.x, acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[c].y, x.y, acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[c].z, x.z, acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[c].w, x.w, acc[c], 0, 0, 0);
    }
  }
}
// The same for RT row tiles of 16 that share every B fragment (a1 = a0 + one LDS tile): acc[t][c] += A_t . W_c.
// A fragment then feeds RT MFMAs: half the weight traffic per multiply-add at RT = 2.
template <int RT, int CG, int KB, bool HEAD = false>
__device__ __forceinline__ void mma_segment_rt(f32x4 (&acc)[RT][CG], const float* __restrict__ a, const int a_tile_stride,
                                               const float4* const (&wf)[CG], const SegHead<RT, CG>* head = nullptr) {
  constexpr int D = seg_depth<RT, CG>();
  static_assert(KB % D == 0, "segment length");
  float4 bq[D][CG];
  pin_pipeline();
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int c = 0; c < CG; ++c) { if constexpr (HEAD) bq[d][c] = head->v[d][c]; else bq[d][c] = wf[c][(size_t)d * 64]; }
  float4 an[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) an[t] = lds_kblock(a + t * a_tile_stride, 0);
#pragma unroll
  for (int kb = 0; kb < KB; kb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      float4 cur[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) cur[c] = bq[d][c];
      float4 av[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) av[t] = an[t];
      if (kb + d + D < KB) {
#pragma unroll
        for (int c = 0; c < CG; ++c) bq[d][c] = wf[c][(size_t)ABL_KB(kb + d + D) * 64];
      }
      if (kb + d + 1 < KB) {
#pragma unroll
        for (int t = 0; t < RT; ++t) an[t] = lds_kblock(a + t * a_tile_stride, kb + d + 1);
      }
      pin_pipeline();
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[c].x, av[t].x, acc[t][c], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[c].y, av[t].y, acc[t][c], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[c].z, av[t].z, acc[t][c], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[c].w, av[t].w, acc[t][c], 0, 0, 0);
    }
  }
}
template <int CG, int KB>
__device__ __forceinline__ void mma_segment(f32x4 (&acc)[CG], const float* __restrict__ a, const float4* __restrict__ wf,
                                            const size_t tile_stride) {
  const float4* wfc[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) wfc[c] = wf + c * tile_stride;
  mma_segment_p<CG, KB>(acc, a, wfc);
}

// A layer over 16 rows: K = NSEG segments of 256 (segment s read from LDS tile seg[s], row stride `as`), N = 16 * 8 * CG
// columns; wavefront w owns column tiles w, w + 8, ...  epi4(row, first column, values) receives the ordered sums of the segments
// for the four consecutive columns a lane owns of one row (pairs of them go through the packed scalar functions, spec_math.hip.h).
template <int NSEG, int CG, class Epi4>
__device__ __forceinline__ void layer256(const float* const (&seg)[NSEG], const int as, const float* __restrict__ w_packed,
                                         const int wave, const int lane, Epi4 epi4) {
  constexpr int K = NSEG * 256;
  const float4* wf = reinterpret_cast<const float4*>(w_packed) + (size_t)wave * (K / 16) * 64 + lane;
  const size_t tile_stride = (size_t)NWAVE * (K / 16) * 64;
  f32x4 tot[CG];
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    f32x4 acc[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_segment<CG, 16>(acc, seg[s] + (lane & 15) * as + lane_koff(lane), wf + (size_t)s * 16 * 64, tile_stride);
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      if (s == 0) tot[c] = acc[c];
      else { tot[c][0] = tot[c][0] + acc[c][0]; tot[c][1] = tot[c][1] + acc[c][1]; tot[c][2] = tot[c][2] + acc[c][2]; tot[c][3] = tot[c][3] + acc[c][3]; }
    }
  }
#pragma unroll
  for (int c = 0; c < CG; ++c) epi4(lane & 15, (wave + NWAVE * c) * 16 + (lane >> 4) * 4, tot[c]);   // columns n0 .. n0 + 3 of one row
}

// [16 rows][256 channels] from a ring into a k-permuted LDS tile; row r = frame `rel` of stream sid[r] (zeros when sid[r] < 0)
// (rowhop: LDS [16], the rows' own step counters in a ragged tick step -- sid[] is then already -1 for streams that sit it out --
//  or nullptr: every row at `pos`)
// HOPS = hops per step: sid[r] is then the ROW (stream * HOPS + hop in step) and the row's frame is `rel + hop in step`
template <int HOPS = 1>
__device__ __forceinline__ void load_tile(float* __restrict__ dst, const Ring& ring, const int* sid /* LDS, [16] */, const int pos, const int rel,
                                          const int tid, const int* rowhop = nullptr) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * NTHR;
    int r, q;
    piece_of<4>(idx >> 6, idx & 63, &r, &q);
    const int b = sid[r];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b >= 0) v = *reinterpret_cast<const float4*>(ring_frame(ring, b / HOPS, rowhop != nullptr ? ring_pos(ring, rowhop[r]) : pos, rel + b % HOPS) + 4 * q);
    store_perm4(dst + r * AS, 4 * q, v.x, v.y, v.z, v.w);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// First half of a conditioned block (MODEL_SPEC 4.4.2): h = gelu(Conv(256->256, k3, dil D)(x)); xa = x + Linear(h).
struct BlockAArgs {
  Ring x, xa;                      // block input (history 2 D frames) and the scratch ring the second half reads
  const float *c1_w, *c1_b, *c2_w, *c2_b;
  const int* hop;
  int B;
};
__device__ __forceinline__ void globalize(BlockAArgs& a) {
  globalize(a.x); globalize(a.xa);
  a.c1_w = as_global(a.c1_w); a.c1_b = as_global(a.c1_b); a.c2_w = as_global(a.c2_w); a.c2_b = as_global(a.c2_b); a.hop = as_global(a.hop);
}
constexpr int kBlockALds = 4 * TILE + 32;   // + sid[16], rowhop[16]
template <int D, bool RAG = false, int HOPS = 1>
__device__ __forceinline__ void block_a_body(const BlockAArgs& a, const int g, float* __restrict__ lds) {
  float* T[3] = {lds, lds + TILE, lds + 2 * TILE};  // taps t-2D, t-D, t
  float* Hh = lds + 3 * TILE;
  int* sid = reinterpret_cast<int*>(lds + 4 * TILE);
  int* rowhop_ = sid + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
  const int* rowhop = stepc::rag_t<RAG>() ? rowhop_ : nullptr;   // (ragged tick step: rows at their streams' own counters)
  if (tid < 16) {
    const int b = g * 16 + tid < a.B * HOPS ? g * 16 + tid : -1;   // (the row: stream * HOPS + hop in step)
    const int hr = b >= 0 ? stepc::of_t<RAG>(hop, b / HOPS) : -1;
    sid[tid] = hr >= 0 ? b : -1;
    rowhop_[tid] = hr >= 0 ? hr : 0;
  }
  __syncthreads();
  const int pos = ring_pos(a.x, hop);
#pragma unroll
  for (int j = 0; j < 3; ++j) load_tile<HOPS>(T[j], a.x, sid, pos, -(2 - j) * D, tid, rowhop);
  __syncthreads();
  {
    const float* const seg[3] = {T[0], T[1], T[2]};
    const float* __restrict__ bias = a.c1_b;
    layer256<3, 2>(seg, AS, a.c1_w, wave, lane, [&](int r, int n0, const f32x4& v) {
      const float4 bn = *reinterpret_cast<const float4*>(bias + n0);
      const bsp::f32x2 g0 = bsp::gelu2(bsp::f32x2{v[0] + bn.x, v[1] + bn.y}), g1 = bsp::gelu2(bsp::f32x2{v[2] + bn.z, v[3] + bn.w});
      store_perm4(Hh + r * AS, n0, g0.x, g0.y, g1.x, g1.y);
    });
  }
  __syncthreads();
  {
    const float* const seg[1] = {Hh};
    const float* __restrict__ bias = a.c2_b;
    const int pos_o = ring_pos(a.xa, hop);
    layer256<1, 2>(seg, AS, a.c2_w, wave, lane, [&](int r, int n0, const f32x4& v) {
      const int b = sid[r];
      if (b < 0) return;
      const float4 bn = *reinterpret_cast<const float4*>(bias + n0);
      const float* x = T[2] + r * AS + kpos(n0);   // the residual: this row's raw input, columns n0 .. n0 + 3
      *reinterpret_cast<float4*>(ring_frame(a.xa, b / HOPS, rowhop != nullptr ? ring_pos(a.xa, rowhop[r]) : pos_o, b % HOPS) + n0) =
          make_float4(x[0] + (v[0] + bn.x), x[KSTEP] + (v[1] + bn.y), x[2 * KSTEP] + (v[2] + bn.z), x[3 * KSTEP] + (v[3] + bn.w));
    });
  }
}
template <int D, int HOPS = 1>
struct BlockAOp {
  using Args = BlockAArgs;
  static constexpr int NTHR = rc::NTHR;
  static constexpr int LDS_FLOATS = kBlockALds;
  static inline dim3 grid(const Args& a) { return dim3((a.B * HOPS + 15) / 16, 1); }
  static inline bhip::LaunchInfo info(const Args& a) {
    return bhip::LaunchInfo{"wave.blk.a", 2.0 * a.B * HOPS * (768.0 + 256.0) * 256.0, 4.0 * ((768.0 + 256.0) * 256.0 + a.B * HOPS * 5.0 * 256.0)};
  }
  static constexpr double wg_cost() { return 20.0; }
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) { block_a_body<D, false, HOPS>(a, bx, lds); }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int, float* lds) { block_a_body<D, RAG, HOPS>(a, bx, lds); }
};
template <int D>
static __global__ __launch_bounds__(NTHR, 4) void block_a_kernel(const BlockAArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kBlockALds];
  block_a_body<D>(a, blockIdx.x, lds);
}

// ---------------------------------------------------------------------------------------------------------------------
// Second half: q = Linear(xa); s = (q . K^T) / 16; o = softmax(s) . V; x' = xa + Linear(o).  Rows are grouped by the
// key/value slot they attend to (the same tile lists as the per-layer attention kernels).
struct BlockBArgs {
  Ring xa, out;                    // first half's output; the next block's input ring
  const float *q_w, *q_b, *o_w, *o_b;
  const float *kt, *v;             // packed per-slot tables, slot stride 256 * 384 floats
  const int* perm;                 // [n_tiles][16] row (stream) indices or -1
  const int* tile_slot;            // [n_tiles]
  const int* hop;
};
__device__ __forceinline__ void globalize(BlockBArgs& a) {
  globalize(a.xa); globalize(a.out);
  a.q_w = as_global(a.q_w); a.q_b = as_global(a.q_b); a.o_w = as_global(a.o_w); a.o_b = as_global(a.o_b);
  a.kt = as_global(a.kt); a.v = as_global(a.v); a.perm = as_global(a.perm); a.tile_slot = as_global(a.tile_slot); a.hop = as_global(a.hop);
}
constexpr int kBlockBLds = 2 * TILE + STILE + 48;   // + inv[16], sid[16], rowhop[16]
template <bool RAG = false, int HOPS = 1>
__device__ __forceinline__ void block_b_body(const BlockBArgs& a, const int g, float* __restrict__ lds) {
  float* XA = lds;
  float* Q = lds + TILE;           // q, later o
  float* S = lds + 2 * TILE;       // scores, then exp(s - max)
  float* inv = S + STILE;
  int* sid = reinterpret_cast<int*>(inv + 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
  const int slot = a.tile_slot[g];
  if (slot < 0) return;
  RC_STAMP(0);
  int* rowhop_ = sid + 16;
  const int* rowhop = stepc::rag_t<RAG>() ? rowhop_ : nullptr;
  if (tid < 16) {
    const int b = a.perm[g * 16 + tid];   // (the row: stream * HOPS + hop in step)
    const int hr = b >= 0 ? stepc::of_t<RAG>(hop, b / HOPS) : -1;
    sid[tid] = hr >= 0 ? b : -1;
    rowhop_[tid] = hr >= 0 ? hr : 0;
  }
  __syncthreads();
  load_tile<HOPS>(XA, a.xa, sid, ring_pos(a.xa, hop), 0, tid, rowhop);
  __syncthreads();
  {
    const float* const seg[1] = {XA};
    const float* __restrict__ bias = a.q_b;
    RC_STAMP(1);
    layer256<1, 2>(seg, AS, a.q_w, wave, lane, [&](int r, int n0, const f32x4& v) {
      const float4 bn = *reinterpret_cast<const float4*>(bias + n0);
      store_perm4(Q + r * AS, n0, v[0] + bn.x, v[1] + bn.y, v[2] + bn.z, v[3] + bn.w);
    });
  }
  __syncthreads();
  RC_STAMP(2);
  {
    const float* const seg[1] = {Q};
    layer256<1, 3>(seg, AS, a.kt + (size_t)slot * B_HID * B_KV_LEN, wave, lane, [&](int r, int n0, const f32x4& v) {
      store_perm4(S + r * SS, n0, v[0] * 0.0625f, v[1] * 0.0625f, v[2] * 0.0625f, v[3] * 0.0625f);
    });
  }
  __syncthreads();
  RC_STAMP(3);
  // softmax statistics, two rows per wavefront (MODEL_SPEC 4.4.2; same operations as attn_pv_kernel); lane l owns the LOGICAL
  // keys l, l + 64, ... of a row (the order of the sums is the spec's), which sit at kpos(l) + 64 i in the k-permuted tile
  const int lp = kpos(lane);
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int r = wave * 2 + rr;
    float v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = S[r * SS + lp + 64 * i];
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int i = 0; i < 6; ++i) mx = fmaxf(mx, v[i]);
    mx = bsp::wmax64(mx);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; i += 2) {   // (pairs through the packed exp: the same bits, half the instructions; the sum in index order as before)
      const bsp::f32x2 e = bsp::exp2(bsp::f32x2{v[i] - mx, v[i + 1] - mx});
      s = s + e.x; s = s + e.y;
      S[r * SS + lp + 64 * i] = e.x; S[r * SS + lp + 64 * (i + 1)] = e.y;
    }
    const float tot = bsp::wsum64(s);
    if (lane == 0) inv[r] = 1.0f / tot;
  }
  __syncthreads();
  RC_STAMP(4);
  {  // o = (segment 0 + segment 1) * (1 / sum): K = 384 = 256 + 128, V packed [384][256]
    const float4* wf = reinterpret_cast<const float4*>(a.v + (size_t)slot * B_KV_LEN * B_HID) + (size_t)wave * (B_KV_LEN / 16) * 64 + lane;
    const size_t tile_stride = (size_t)NWAVE * (B_KV_LEN / 16) * 64;
    const float* ap = S + (lane & 15) * SS + lane_koff(lane);
    f32x4 acc0[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, acc1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    mma_segment<2, 16>(acc0, ap, wf, tile_stride);
    mma_segment<2, 8>(acc1, ap + 256, wf + (size_t)16 * 64, tile_stride);
    const int r = lane & 15;
    const float ir = inv[r];
#pragma unroll
    for (int c = 0; c < 2; ++c)
      store_perm4(Q + r * AS, (wave + NWAVE * c) * 16 + (lane >> 4) * 4, (acc0[c][0] + acc1[c][0]) * ir, (acc0[c][1] + acc1[c][1]) * ir,
                  (acc0[c][2] + acc1[c][2]) * ir, (acc0[c][3] + acc1[c][3]) * ir);
  }
  __syncthreads();
  RC_STAMP(5);
  {
    const float* const seg[1] = {Q};
    const float* __restrict__ bias = a.o_b;
    const int pos_o = ring_pos(a.out, hop);
    layer256<1, 2>(seg, AS, a.o_w, wave, lane, [&](int r, int n0, const f32x4& v) {
      const int b = sid[r];
      if (b < 0) return;
      const float4 bn = *reinterpret_cast<const float4*>(bias + n0);
      const float* x = XA + r * AS + kpos(n0);   // the residual: this row of xa, columns n0 .. n0 + 3
      *reinterpret_cast<float4*>(ring_frame(a.out, b / HOPS, rowhop != nullptr ? ring_pos(a.out, rowhop[r]) : pos_o, b % HOPS) + n0) =
          make_float4(x[0] + (v[0] + bn.x), x[KSTEP] + (v[1] + bn.y), x[2 * KSTEP] + (v[2] + bn.z), x[3 * KSTEP] + (v[3] + bn.w));
    });
  }
  RC_STAMP(6);
}
template <int HOPS = 1>
struct BlockBOpH {
  using Args = BlockBArgs;
  static constexpr int NTHR = rc::NTHR;
  static constexpr int LDS_FLOATS = kBlockBLds;
  static constexpr double wg_cost() { return 24.0; }
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) { block_b_body<false, HOPS>(a, bx, lds); }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int, float* lds) { block_b_body<RAG, HOPS>(a, bx, lds); }
};
using BlockBOp = BlockBOpH<1>;
static __global__ __launch_bounds__(NTHR, 4) void block_b_kernel(const BlockBArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kBlockBLds];
  block_b_body(a, blockIdx.x, lds);
}

// ---------------------------------------------------------------------------------------------------------------------
// The attention half for rows that do NOT share their K/V slot with 15 neighbours (many speakers per GPU, BASELINE.json
// configs[3]): one workgroup per PAIR OF QUADS, a quad = up to four rows of one slot; q and the output layer as above on
// the 8-row tile, the two products against the slots' own K / V on v_mfma_f32_4x4x1_16b_f32.
//
// The multi-block form runs 16 independent [4 x 1] . [1 x 4] products per instruction: A = lane 4 b + i -> row i of block b,
// B = lane 4 b + j -> column j of block b, D = register r of lane 4 b + j -> (row r, column j).  Here every block gets the
// SAME four rows (lane l supplies row l & 3) and its own four columns, so one instruction is a rank-1 update of
// [4 rows] x [64 columns] without padded rows (a 16 x 16 x 4 tile would spend 12 of its 16 rows on nothing -- round 2 ran
// 324 such tiles per tick at 64 speakers), and a lane's B operand is simply "its column at reduction index k" of a plain
// row-major matrix: NV adjacent columns per lane come from one load and feed NV instructions.  Arithmetic as everywhere:
// one k-ascending fma chain per output from 0, bit-identical to the 16 x 16 x 4 chain (tools/microbench/mfma_4x4.hip:
// layout and bits against fmaf; ~10.7 cycles per instruction).
// What bounds it is not the matrix pipe.  Nothing re-reads a K/V table within a tick, so the 786 KB per (quad, block) come
// from HBM / the Infinity Cache, and ONE compute unit pulls ~35 bytes per cycle through its L1 whatever the kernel does
// (tools/microbench/blockb_timing.hip; profiles/r03_notes.md section 4): the stream has to be spread over all CUs (four
// quads per workgroup took 125-150 us per workgroup) with enough bytes in flight per workgroup (one quad per workgroup on
// four wavefronts: 70 us inside the tick, latency-bound).
template <int NV> struct QuadVec;
template <> struct QuadVec<1> { using T = float; };
template <> struct QuadVec<2> { using T = float2; };
template <> struct QuadVec<4> { using T = float4; };
__device__ __forceinline__ float quad_elem(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
__device__ __forceinline__ float quad_elem(const float2& v, int c) { return c == 0 ? v.x : v.y; }
__device__ __forceinline__ float quad_elem(const float& v, int) { return v; }
// acc[c] += A[4 x KLEN] . B[KLEN x (this lane's column c)]: `a` = this lane's row in LDS (8-byte aligned, first k of the
// segment), `bp` = this lane's first column in row-major B (row stride LDB floats), first k of the segment.  Rows of B are
// fetched D ahead: 64 registers of rows in flight per lane (32 at NV = 1) are what fits under the launch's 128.
template <int NV, int KLEN, int LDB>
__device__ __forceinline__ void quad_segment(f32x4 (&acc)[NV], const float* __restrict__ a, const float* __restrict__ bp) {
  using V = typename QuadVec<NV>::T;
  constexpr int D = NV == 1 ? 32 : 64 / NV;  // k rows in flight
  constexpr int G = 8;                       // A operand: float2 reads, one group of G k ahead
  static_assert(KLEN % D == 0 && D % G == 0, "segment length");
  V bq[D];
  float2 an[G / 2];
  pin_pipeline();
#pragma unroll
  for (int d = 0; d < D; ++d) bq[d] = *reinterpret_cast<const V*>(bp + (size_t)d * LDB);
#pragma unroll
  for (int d = 0; d < G / 2; ++d) an[d] = *reinterpret_cast<const float2*>(a + 2 * d);
#pragma unroll 1
  for (int k = 0; k < KLEN; k += D) {
    const bool more = k + D < KLEN;
#pragma unroll
    for (int g = 0; g < D; g += G) {
      float2 ac[G / 2];
#pragma unroll
      for (int d = 0; d < G / 2; ++d) ac[d] = an[d];
      if (k + g + G < KLEN) {
#pragma unroll
        for (int d = 0; d < G / 2; ++d) an[d] = *reinterpret_cast<const float2*>(a + k + g + G + 2 * d);
      }
      V cur[G];
#pragma unroll
      for (int d = 0; d < G; ++d) cur[d] = bq[g + d];
      if (more) {
#pragma unroll
        for (int d = 0; d < G; ++d) bq[g + d] = *reinterpret_cast<const V*>(bp + (size_t)(k + g + d + D) * LDB);
      }
      pin_pipeline();
#pragma unroll
      for (int d = 0; d < G; ++d)
#pragma unroll
        for (int c = 0; c < NV; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32((d & 1) ? ac[d / 2].y : ac[d / 2].x, quad_elem(cur[d], c), acc[c], 0, 0, 0);
    }
  }
}
struct BlockBqArgs {
  Ring xa, out;
  const float *q_w, *q_b, *o_w, *o_b;    // as BlockBArgs (MFMA-fragment order)
  const float *ktp, *vp;                 // plain per-slot tables: K^T [256][384], V [384][256]
  const int* qperm;                      // [n_quads][4] row (stream) indices or -1
  const int* qslot;                      // [n_quads] slot or -1 (quad unused)
  const int* hop;
};
__device__ __forceinline__ void globalize(BlockBqArgs& a) {
  globalize(a.xa); globalize(a.out);
  a.q_w = as_global(a.q_w); a.q_b = as_global(a.q_b); a.o_w = as_global(a.o_w); a.o_b = as_global(a.o_b);
  a.ktp = as_global(a.ktp); a.vp = as_global(a.vp); a.qperm = as_global(a.qperm); a.qslot = as_global(a.qslot); a.hop = as_global(a.hop);
}
constexpr int kBlockBqLds = kBlockBLds;
template <bool RAG = false, int HOPS = 1>
__device__ __forceinline__ void block_bq_body(const BlockBqArgs& a, const int g, float* __restrict__ lds) {
  float* XA = lds;
  float* Q = lds + TILE;           // q, later o
  float* S = lds + 2 * TILE;       // scores, then exp(s - max)
  float* inv = S + STILE;
  int* sid = reinterpret_cast<int*>(inv + 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
  const int2 qs = *reinterpret_cast<const int2*>(a.qslot + 2 * g);
  if (qs.x < 0 && qs.y < 0) return;
  // These workgroups are the longest of the launch (two K/V tables through one CU's L1) and end it: issue priority over the
  // co-resident workgroup (zero-sum for the SIMD, profiles/r03_notes.md section 2, but it shortens the launch's tail).
  __builtin_amdgcn_s_setprio(3);
  RC_STAMP(0);
  int* rowhop_ = sid + 16;
  const int* rowhop = stepc::rag_t<RAG>() ? rowhop_ : nullptr;
  if (tid < 16) {   // rows 8..15 of the tile stay empty
    const int b = tid < 8 ? a.qperm[g * 8 + tid] : -1;
    const int hr = b >= 0 ? stepc::of_t<RAG>(hop, b / HOPS) : -1;
    sid[tid] = hr >= 0 ? b : -1;
    rowhop_[tid] = hr >= 0 ? hr : 0;
  }
  __syncthreads();
  load_tile<HOPS>(XA, a.xa, sid, ring_pos(a.xa, hop), 0, tid, rowhop);
  __syncthreads();
  {
    const float* const seg[1] = {XA};
    const float* __restrict__ bias = a.q_b;
    RC_STAMP(1);
    // (q in LOGICAL column order: the quad products below read a row's k in sequence; XA above and o below are k-permuted tiles)
    layer256<1, 2>(seg, AS, a.q_w, wave, lane, [&](int r, int n0, const f32x4& v) {
      const float4 bn = *reinterpret_cast<const float4*>(bias + n0);
      float2* d = reinterpret_cast<float2*>(Q + r * AS + n0);
      d[0] = make_float2(v[0] + bn.x, v[1] + bn.y); d[1] = make_float2(v[2] + bn.z, v[3] + bn.w);
    });
  }
  __syncthreads();
  RC_STAMP(2);
  const int quad = wave_u >> 2, part = wave_u & 3;   // four wavefronts per quad
  const int my_slot = quad == 0 ? qs.x : qs.y;
  const int arow = quad * 4 + (lane & 3);            // the row this lane supplies as the A operand
  if (part < 3) {  // s = (q . K^T) / 16: three wavefronts x 128 keys, two per lane
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int j = part * 128 + 2 * lane;
    if (my_slot >= 0) quad_segment<2, B_HID, B_KV_LEN>(acc, Q + arow * AS, a.ktp + (size_t)my_slot * B_HID * B_KV_LEN + j);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) S[(quad * 4 + r) * SS + j + c] = acc[c][r] * 0.0625f;
  }
  __syncthreads();
  RC_STAMP(3);
  {  // softmax statistics, one row per wavefront (MODEL_SPEC 4.4.2; same operations as block_b_body)
    const int r = wave;
    float v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = S[r * SS + lane + 64 * i];
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int i = 0; i < 6; ++i) mx = fmaxf(mx, v[i]);
    mx = bsp::wmax64(mx);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; i += 2) {   // (pairs through the packed exp: the same bits, half the instructions; the sum in index order as before)
      const bsp::f32x2 e = bsp::exp2(bsp::f32x2{v[i] - mx, v[i + 1] - mx});
      s = s + e.x; s = s + e.y;
      S[r * SS + lane + 64 * i] = e.x; S[r * SS + lane + 64 * (i + 1)] = e.y;
    }
    const float tot = bsp::wsum64(s);
    if (lane == 0) inv[r] = 1.0f / tot;
  }
  __syncthreads();
  RC_STAMP(4);
  {  // o = (segment 0 + segment 1) * (1 / sum): four wavefronts x 64 channels
    const int n = part * 64 + lane;
    f32x4 acc0[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, acc1[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    if (my_slot >= 0) {
      const float* vp = a.vp + (size_t)my_slot * B_KV_LEN * B_HID + n;
      quad_segment<1, 256, B_HID>(acc0, S + arow * SS, vp);
      quad_segment<1, 128, B_HID>(acc1, S + arow * SS + 256, vp + (size_t)256 * B_HID);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = acc0[0][r] + acc1[0][r];
      Q[(quad * 4 + r) * AS + kpos(n)] = v * inv[quad * 4 + r];   // (o: the output layer's k-permuted operand tile)
    }
  }
  __syncthreads();
  RC_STAMP(5);
  {
    const float* const seg[1] = {Q};
    const float* __restrict__ bias = a.o_b;
    const int pos_o = ring_pos(a.out, hop);
    layer256<1, 2>(seg, AS, a.o_w, wave, lane, [&](int r, int n0, const f32x4& v) {
      const int b = sid[r];
      if (b < 0) return;
      const float4 bn = *reinterpret_cast<const float4*>(bias + n0);
      const float* x = XA + r * AS + kpos(n0);   // the residual: this row of xa, columns n0 .. n0 + 3
      *reinterpret_cast<float4*>(ring_frame(a.out, b / HOPS, rowhop != nullptr ? ring_pos(a.out, rowhop[r]) : pos_o, b % HOPS) + n0) =
          make_float4(x[0] + (v[0] + bn.x), x[KSTEP] + (v[1] + bn.y), x[2 * KSTEP] + (v[2] + bn.z), x[3 * KSTEP] + (v[3] + bn.w));
    });
  }
  __builtin_amdgcn_s_setprio(0);
  RC_STAMP(6);
}
template <int HOPS = 1>
struct BlockBqOpH {
  using Args = BlockBqArgs;
  static constexpr int NTHR = rc::NTHR;
  static constexpr int LDS_FLOATS = kBlockBqLds;
  static constexpr double wg_cost() { return 30.0; }
  __device__ static __forceinline__ void run(const Args& a, int bx, int, float* lds) { block_bq_body<false, HOPS>(a, bx, lds); }
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args& a, int bx, int, float* lds) { block_bq_body<RAG, HOPS>(a, bx, lds); }
};
using BlockBqOp = BlockBqOpH<1>;
// ---------------------------------------------------------------------------------------------------------------------
// Any conv_gemm Layer for ONE tile of 16 rows and ALL its output columns: the A operand streams through two LDS tiles
// one 256-long reduction segment at a time (gathered from the input ring exactly as conv_gemm does, next segment's
// loads in flight during the current segment's MFMAs, one barrier per segment), the eight wavefronts own the
// column tiles w, w + 8, ...  Same results as conv_gemm_kernel<L, *> (same segments, same order, same epilogue).  Inside
// a tick this replaces conv_gemm's 64-wide k-chunks (two barriers and an exposed load latency per 16 MFMA steps):
// phone.rb (K = 1280): 65 -> 46 us per workgroup with two workgroups per CU.
constexpr int kConvRowsLds = 2 * TILE + 32;
template <int RT> constexpr int conv_rows_lds() { return 2 * RT * TILE + 48 * RT; }   // + per row: stream, frame, the stream's step counter
// COLS = output columns per workgroup (0 = all): a wide layer can be cut into column slabs, one workgroup each (grid y).
// RT = row tiles of 16 per workgroup: at 2 every weight fragment feeds two MFMAs (half the weight traffic per row).
template <class L, int COLS = 0, int RT = 1, bool RAG = false>
__device__ __forceinline__ void conv_rows_body(const ConvArgs& a, const int bx, const int by, float* __restrict__ lds) {
  constexpr int NCOL = COLS > 0 ? COLS : L::NOUT;
  static_assert(L::NOUT % NCOL == 0 && NCOL % 16 == 0, "column slabs");
  constexpr int K = L::K, P = L::P, NTL = NCOL / 16, CG = (NTL + NWAVE - 1) / NWAVE;
  constexpr int ROWS = 16 * RT;
  const int nt_base = by * NTL;  // first column tile of this workgroup's slab
  constexpr int LAST = K - 256 * (P - 1);  // length of the last segment
  static_assert(K % 16 == 0 && LAST % 64 == 0 && L::NOUT % 16 == 0 && !L::GROUPED, "layer shape");
  float* slot[2] = {lds, lds + RT * TILE};   // a slot = RT tiles of [16][256], one after the other
  int* rb_ = reinterpret_cast<int*>(lds + 2 * RT * TILE);  // [ROWS] stream, [ROWS] frame of each row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
  // the first thing a workgroup does: request the first weight fragments of its first segment -- that round trip then runs beside the
  // row table's barrier and the row gather instead of behind them (a short body is mostly such round trips, profiles/r05_notes.md)
  const float4* wfc[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    const int nt = wave + NWAVE * c < NTL ? wave + NWAVE * c : NTL - 1;  // surplus tiles of a ragged layer recompute the last one
    wfc[c] = reinterpret_cast<const float4*>(a.w) + (size_t)(nt_base + nt) * (K / 16) * 64 + lane;
  }
#ifndef RC_SEG_HEAD
#define RC_SEG_HEAD 0   // A/B switch, OFF: 1 = the first segment's first weight fragments requested before the row gather -- measured 256 streams x 2 hops 118.1 -> 120.5 us (slower), x 4 hops 237.4 -> 236.4, x 1 hop unchanged (profiles/r05_notes.md)
#endif
  SegHead<RT, CG> head;
  if constexpr (RC_SEG_HEAD != 0) seg_head_load<RT, CG>(head, wfc);
  const int M = a.B * L::T;
  const bool rag = stepc::rag_t<RAG>();   // (ragged tick step: every row at its stream's own counter; absent streams' rows drop out)
  if (tid < ROWS) {
    const int m = bx * ROWS + tid;
    const int b = m < M ? m / L::T : -1;
    const int hr = b >= 0 ? stepc::of_t<RAG>(hop, b) : -1;
    rb_[tid] = hr >= 0 ? b : -1;
    rb_[ROWS + tid] = m < M ? m % L::T : 0;
    rb_[2 * ROWS + tid] = hr >= 0 ? hr : 0;
  }
  __syncthreads();
  const int pos_in = ring_pos(a.in, hop);
  // this thread's 2 RT 16-byte pieces of a segment: (row pr[i], piece pq[i]), wave-load wave + 8 i (piece_of)
  int pr[2 * RT], pq[2 * RT], pb[2 * RT], pt[2 * RT];
#pragma unroll
  for (int i = 0; i < 2 * RT; ++i) {
    piece_of<4 * RT>(wave + NWAVE * i, lane, &pr[i], &pq[i]);
    pb[i] = rb_[pr[i]]; pt[i] = rb_[ROWS + pr[i]];
  }
  float4 nx[2 * RT];
  auto load_seg = [&](int s) {
#pragma unroll
    for (int i = 0; i < 2 * RT; ++i) {
      const int kk = s * 256 + 4 * pq[i];
      const bool live = kk < K;
      const int j = live ? kk / L::CIN : 0, c = live ? kk % L::CIN : 0;
      nx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && pb[i] >= 0)
        nx[i] = *reinterpret_cast<const float4*>(ring_frame(a.in, pb[i], rag ? ring_pos(a.in, rb_[2 * ROWS + pr[i]]) : pos_in,   // (ragged: worked out per load, no register held for it in the common case)
                                                            (pt[i] + 1) * L::STRIDE - 1 - (L::KSZ - 1 - j) * L::DIL + a.rel_shift) + c);
    }
  };
  auto store_seg = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < 2 * RT; ++i) {
      float4 v = nx[i];
      if constexpr (L::PRE == PRE_LRELU) { v.x = bsp::lrelu(v.x); v.y = bsp::lrelu(v.y); v.z = bsp::lrelu(v.z); v.w = bsp::lrelu(v.w); }
      store_perm4(dst + pr[i] * AS, 4 * pq[i], v.x, v.y, v.z, v.w);   // (row 16 t + r of the slot = row r of its tile t: tiles are contiguous)
    }
  };
  load_seg(0);
  store_seg(slot[0]);
  __syncthreads();
  f32x4 tot[RT][CG];
#pragma unroll
  for (int s = 0; s < P; ++s) {
    if (s + 1 < P) load_seg(s + 1);
    f32x4 acc[RT][CG];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float4* wfs[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) wfs[c] = wfc[c] + (size_t)s * 16 * 64;
    const float* ap = slot[s & 1] + (lane & 15) * AS + lane_koff(lane);
    if (RC_SEG_HEAD != 0 && s == 0) {   // (s is a compile-time index after unrolling)
      if constexpr (RT == 1) {
        if constexpr (P > 1) mma_segment_p<CG, 16, true>(acc[0], ap, wfs, &head);
        else mma_segment_p<CG, LAST / 16, true>(acc[0], ap, wfs, &head);
      } else {
        if constexpr (P > 1) mma_segment_rt<RT, CG, 16, true>(acc, ap, TILE, wfs, &head);
        else mma_segment_rt<RT, CG, LAST / 16, true>(acc, ap, TILE, wfs, &head);
      }
    } else if constexpr (RT == 1) {
      if (s + 1 < P) mma_segment_p<CG, 16>(acc[0], ap, wfs);
      else mma_segment_p<CG, LAST / 16>(acc[0], ap, wfs);
    } else {
      if (s + 1 < P) mma_segment_rt<RT, CG, 16>(acc, ap, TILE, wfs);
      else mma_segment_rt<RT, CG, LAST / 16>(acc, ap, TILE, wfs);
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        if (s == 0) tot[t][c] = acc[t][c];
        else { tot[t][c][0] = tot[t][c][0] + acc[t][c][0]; tot[t][c][1] = tot[t][c][1] + acc[t][c][1]; tot[t][c][2] = tot[t][c][2] + acc[t][c][2]; tot[t][c][3] = tot[t][c][3] + acc[t][c][3]; }
      }
    if (s + 1 < P) {
      store_seg(slot[(s + 1) & 1]);
      __syncthreads();
    }
  }
  // epilogue: conv_gemm's, operation for operation; a lane owns columns n0 .. n0 + 3 of ONE row per accumulator (the transposed
  // form, top of this file): one row lookup, 16-byte bias / residual loads and one 16-byte store
  const int pos_out = ring_pos(a.out, hop), R_out = a.out.n * a.out.m;
  int pos_res = 0, R_res = 0;
  if constexpr (L::RES) { pos_res = ring_pos(a.res, hop); R_res = a.res.n * a.res.m; }
  float4 bias_n[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    const int nt = wave + NWAVE * c < NTL ? wave + NWAVE * c : NTL - 1;
    bias_n[c] = L::EPI == EPI_BIAS ? *reinterpret_cast<const float4*>(a.bias + (nt_base + nt) * 16 + (lane >> 4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int r = 16 * t + (lane & 15);
    const int rb = rb_[r], rf = rb_[ROWS + r];
    int po = pos_out, pr = pos_res;
    if (rag) {
      const int hr = rb_[2 * ROWS + r];
      po = ring_pos(a.out, hr);
      if constexpr (L::RES) pr = ring_pos(a.res, hr);
    }
    float rs = 1.0f;
    if constexpr (L::EPI == EPI_ROWSCALE) rs = a.rowscale[(rb < 0 ? 0 : rb) * L::T + rf];
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      if (wave + NWAVE * c >= NTL) continue;
      const int n0 = (nt_base + wave + NWAVE * c) * 16 + (lane >> 4) * 4;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tot[t][c][e];
      if constexpr (L::EPI == EPI_BIAS) { v[0] = v[0] + bias_n[c].x; v[1] = v[1] + bias_n[c].y; v[2] = v[2] + bias_n[c].z; v[3] = v[3] + bias_n[c].w; }
      if constexpr (L::EPI == EPI_SCALE) { for (int e = 0; e < 4; ++e) v[e] = v[e] * a.scale; }
      if constexpr (L::EPI == EPI_ROWSCALE) { for (int e = 0; e < 4; ++e) v[e] = v[e] * rs; }
      if constexpr (L::ACT == ACT_GELU) {   // pairs through the packed scalar functions (spec_math.hip.h): the same bits
        const bsp::f32x2 g0 = bsp::gelu2(bsp::f32x2{v[0], v[1]}), g1 = bsp::gelu2(bsp::f32x2{v[2], v[3]});
        v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
      }
      if (rb < 0) continue;
      // (32-bit index arithmetic: a ring holds fewer than 2^32 floats, RingArena::build)
      if constexpr (L::RES) {
        const float4 x = *reinterpret_cast<const float4*>(a.res.base + ((unsigned)(rb * R_res + pr) * (unsigned)a.res.C + (unsigned)(rf * L::NOUT + n0)));
        v[0] = x.x + v[0]; v[1] = x.y + v[1]; v[2] = x.z + v[2]; v[3] = x.w + v[3];
      }
      *reinterpret_cast<float4*>(a.out.base + ((unsigned)(rb * R_out + po) * (unsigned)a.out.C + (unsigned)(rf * L::NOUT + n0))) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}
template <class L, int COLS = 0, int RT = 1>
struct ConvRowsOp {
  using Args = ConvArgs;
  static constexpr int NTHR = rc::NTHR;
  static constexpr int LDS_FLOATS = conv_rows_lds<RT>();
  static inline dim3 grid(const ConvArgs& a) { return dim3((a.B * L::T + 16 * RT - 1) / (16 * RT), COLS > 0 ? L::NOUT / COLS : 1); }
  static inline bhip::LaunchInfo info(const char* name, const ConvArgs& a) { return ConvOp<L, TileCfg<1, 1, 1, 2, 1>>::info(name, a); }
  __device__ static __forceinline__ void run(const ConvArgs& a, int bx, int by, float* lds) { conv_rows_body<L, COLS, RT>(a, bx, by, lds); }
  template <bool RAG> __device__ static __forceinline__ void run_t(const ConvArgs& a, int bx, int by, float* lds) { conv_rows_body<L, COLS, RT, RAG>(a, bx, by, lds); }
};

// a body type that is never added to a table (keeps the type lists of the tick launch the same length at every hops-per-step)
struct NopOp {
  struct Args { int unused; };
  static constexpr int NTHR = 64;
  static constexpr int LDS_FLOATS = 0;
  __device__ static __forceinline__ void run(const Args&, int, int, float*) {}
  template <bool RAG> __device__ static __forceinline__ void run_t(const Args&, int, int, float*) {}
};
__device__ __forceinline__ void globalize(NopOp::Args&) {}

// the same body as a launch of its own (the in-order chain at large batches, wave.hip)
template <class L, int COLS, int RT>
static __global__ __launch_bounds__(NTHR, 4) void conv_rows_kernel(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[conv_rows_lds<RT>()];
  conv_rows_body<L, COLS, RT>(a, blockIdx.x, blockIdx.y, lds);
}
template <class L, int COLS = 0, int RT = 2>
static inline void launch_conv_rows(const char* name, const ConvArgs& a, hipStream_t stream) {
  using Op = ConvRowsOp<L, COLS, RT>;
  bhip::launch_site(Op::info(name, a), stream, [&] { hipLaunchKernelGGL((conv_rows_kernel<L, COLS, RT>), Op::grid(a), dim3(NTHR), 0, stream, a); });
}

}  // namespace rc

// launch_auto's many-row branch (conv_gemm.hip.h): 32 rows x the layer's width per workgroup where conv_rows_body applies
// -- at 8192 streams 8-13 % faster per launch than the 64 x 64 tiling (wave.up2 74 -> 64 us, up1 63 -> 59, res1b 58 -> 54):
// a weight fragment feeds two MFMAs and there is one barrier per 256-long segment instead of two per 64 columns of K.
// BEATRICE_HIP_NO_CONVROWS=1: the 64 x 64 tiling (A/B measurements).
template <class L>
static inline void launch_many_rows(const char* name, const ConvArgs& a, hipStream_t s) {
  constexpr int LAST = L::K - 256 * (L::P - 1);
  constexpr bool ok = !L::GROUPED && L::K % 16 == 0 && LAST % 64 == 0 && L::NOUT % 16 == 0;
  if constexpr (ok) {
    static const bool off = bhip::meas_env("BEATRICE_HIP_NO_CONVROWS") != nullptr;
    if (!off) {
      constexpr int COLS = (L::NOUT % 128 == 0 && (L::NOUT > 256 || (L::NOUT == 256 && L::K >= 768))) ? 128 : 0;
      rc::launch_conv_rows<L, COLS, 2>(name, a, s);
      return;
    }
  }
  launch_conv<L, TL>(name, a, 0, s);
}
