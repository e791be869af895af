// lat_gemm.hip.h -- experimental single-shot variant of conv_gemm_kernel (not part of the product
// library; kept for tools/microbench/lat_ablate.hip, which decomposes the cost of a latency-bound launch).
#pragma once
#include "conv_gemm.hip.h"

// ------------------------------------------------------------------------------------------------
// lat_gemm_kernel -- the same layer for FEW rows (M <= ~2048: every T = 1 layer of the per-hop
// chain at B <= 2048).  These launches are latency-bound, not throughput-bound: a 16-row tile and a
// 256-long reduction segment is 64 dependent MFMAs (~1 us), and everything else in the kernel used
// to be serialised memory latency.  So this variant issues ALL of its global loads up front:
//   * the W fragment of the wave's 16 columns for the whole segment goes straight into 64 VGPRs in
//     MFMA B-operand layout (lane (j, kq) holds k = 4*ks + kq): no LDS round trip, and the loads do
//     not depend on the hop counter, so they overlap the scalar hop load and the address math;
//   * the 16 x 256 A segment is fetched as 16-byte pieces and written once to LDS (stride 258);
//   * bias and residual of the thread's output elements are fetched before the MFMA chain.
// One workgroup = P k-groups (one per segment) x LN waves; one barrier before and one after the
// chain; segment results are combined through LDS in ascending order (MODEL_SPEC 2.2).
// ABL: ablation switches for tools/microbench (bit 0: no W loads, bit 1: no A loads, bit 2: no MFMA
// chain, bit 3: no bias/residual prefetch, bit 4: no hop load).  0 in the product.
template <class L, int LN, int ABL = 0>
__global__ __launch_bounds__(64 * LN * L::P) void lat_gemm_kernel(const ConvArgs a) {
  constexpr int P = L::P, SEG = L::SEG, NT = 16 * LN, MT = 16, GTHR = 64 * LN, NTHR = GTHR * P;
  constexpr int AS = SEG + 2;
  constexpr int A_F4 = MT * SEG / 4;                    // float4 pieces per segment tile
  constexpr int A_SLOTS = (A_F4 + GTHR - 1) / GTHR;
  constexpr int E_SLOTS = (MT * NT + NTHR - 1) / NTHR;  // output elements per thread
  constexpr int RED_FLOATS = P > 1 ? P * MT * NT : 0;
  constexpr int LDS_FLOATS = P * MT * AS > RED_FLOATS ? P * MT * AS : RED_FLOATS;
  static_assert(L::NOUT % NT == 0, "N tile must divide NOUT");
  static_assert(L::CIN % 4 == 0 && L::K % 4 == 0, "16-byte pieces must not straddle a tap");
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x;
  const int grp = tid / GTHR, gtid = tid % GTHR;
  const int lane = gtid & 63, wave = gtid >> 6;
  const int m0 = blockIdx.x * MT, n0 = blockIdx.y * NT;
  const int M = a.B * L::T;
  const int kk0 = grp * SEG;  // first reduction index of this group's segment

  const float* wbase = a.w;
  if constexpr (L::GROUPED) {
    const int slot = a.tile_slot[blockIdx.x];
    if (slot < 0) return;
    wbase += (size_t)slot * a.w_slot_stride;
  }

  // ---- W fragment -> registers (independent of the hop counter): 16 coalesced float4 loads
  float breg[SEG / 4];
  {
    const float4* wp = reinterpret_cast<const float4*>(wbase) + ((size_t)((n0 + wave * 16) >> 4) * (L::K >> 4) + (kk0 >> 4)) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < SEG / 16; ++kb) {
      const bool live = kk0 + 16 * kb < L::K && !(ABL & 1);
      const float4 f = wp[(size_t)(live ? kb : 0) * 64];
      breg[4 * kb + 0] = live ? f.x : 0.f; breg[4 * kb + 1] = live ? f.y : 0.f;
      breg[4 * kb + 2] = live ? f.z : 0.f; breg[4 * kb + 3] = live ? f.w : 0.f;
    }
  }

  const int hop = (ABL & 16) ? 0 : *a.hop;
  const int pos_in = ring_pos(a.in, hop);

  // ---- A segment -> registers -> LDS
  float4 areg[A_SLOTS];
#pragma unroll
  for (int s = 0; s < A_SLOTS; ++s) {
    const int idx = gtid + s * GTHR;
    const int r = idx / (SEG / 4), q = idx % (SEG / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int kk = kk0 + 4 * q;
    if (idx < A_F4 && kk < L::K && !(ABL & 2)) {
      int b = -1, t = 0;
      if constexpr (L::GROUPED) {
        const int m = a.perm[blockIdx.x * MT + r];
        if (m >= 0) { b = m / L::T; t = m % L::T; }
      } else {
        const int m = m0 + r;
        if (m < M) { b = m / L::T; t = m % L::T; }
      }
      if (b >= 0) {
        const int j = kk / L::CIN, c = kk % L::CIN;
        const int rel = (t + 1) * L::STRIDE - 1 - (L::KSZ - 1 - j) * L::DIL + a.rel_shift;
        v = *reinterpret_cast<const float4*>(ring_frame(a.in, b, pos_in, rel) + c);
      }
    }
    areg[s] = v;
  }

  // ---- epilogue operands of this thread's output elements (bias, residual, addresses)
  const int pos_out = ring_pos(a.out, hop);
  const int R_out = a.out.n * a.out.m;
  float e_bias[E_SLOTS], e_res[E_SLOTS], e_rs[E_SLOTS];
  float* e_dst[E_SLOTS];
#pragma unroll
  for (int s = 0; s < E_SLOTS; ++s) {
    const int idx = tid + s * NTHR;
    e_dst[s] = nullptr;
    e_bias[s] = 0.f; e_res[s] = 0.f; e_rs[s] = 1.f;
    if (idx < MT * NT) {
      const int r = idx / NT, n = n0 + idx % NT;
      int b = -1, t = 0;
      if constexpr (L::GROUPED) {
        const int m = a.perm[blockIdx.x * MT + r];
        if (m >= 0) { b = m / L::T; t = m % L::T; }
      } else {
        const int m = m0 + r;
        if (m < M) { b = m / L::T; t = m % L::T; }
      }
      if (b >= 0) {
        e_dst[s] = a.out.base + ((size_t)b * R_out + pos_out) * a.out.C + (size_t)t * L::NOUT + n;
        if constexpr (L::EPI == EPI_BIAS && !(ABL & 8)) e_bias[s] = a.bias[n];
        if constexpr (L::EPI == EPI_ROWSCALE) e_rs[s] = a.rowscale[b * L::T + t];
        if constexpr (L::RES && !(ABL & 8)) {
          const int R_res = a.res.n * a.res.m;
          e_res[s] = a.res.base[((size_t)b * R_res + ring_pos(a.res, hop)) * a.res.C + (size_t)t * L::NOUT + n];
        }
      }
    }
  }

  float* As = lds + grp * MT * AS;
#pragma unroll
  for (int s = 0; s < A_SLOTS; ++s) {
    const int idx = gtid + s * GTHR;
    if (idx < A_F4 && !(ABL & 64)) {
      const int r = idx / (SEG / 4), q = idx % (SEG / 4);
      float4 v = areg[s];
      if constexpr (L::PRE == PRE_LRELU) {
        v.x = bsp::lrelu(v.x); v.y = bsp::lrelu(v.y); v.z = bsp::lrelu(v.z); v.w = bsp::lrelu(v.w);
      }
      float2* dst = reinterpret_cast<float2*>(&As[r * AS + 4 * q]);
      dst[0] = make_float2(v.x, v.y);
      dst[1] = make_float2(v.z, v.w);
    }
  }
  __syncthreads();

  // ---- one dependent MFMA chain over the segment.  Every group but the last has a full segment;
  // the last one has K - 256*(P-1) reduction indices: both counts are compile-time, so the chain is
  // straight-line code (a per-step bound check would put a branch between dependent MFMAs).
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  {
    // A fragment LDS -> registers first (64 independent ds_reads), then a pure register MFMA chain:
    // a ds_read inside the chain would add its ~100-cycle latency to every 44-cycle dependent MFMA.
    const float* ap = As + (lane & 15) * AS + (lane >> 4);
    constexpr int LAST_STEPS = (ABL & 4) ? 1 : (L::K - SEG * (P - 1)) / 4;
    float afr[SEG / 4];
#pragma unroll
    for (int ks = 0; ks < SEG / 4; ++ks) afr[ks] = ap[4 * ks];
    if (P == 1 || grp == P - 1) {
#pragma unroll
      for (int ks = 0; ks < LAST_STEPS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[ks], breg[ks], acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < SEG / 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[ks], breg[ks], acc, 0, 0, 0);
    }
  }

  // ---- combine segments in ascending order, epilogue
  float v_out[E_SLOTS];
  if constexpr (P > 1 && !(ABL & 32)) {
    __syncthreads();  // every group is done reading its A tile; reuse LDS
    float* red = lds;
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(grp * MT + (lane >> 4) * 4 + e) * NT + wave * 16 + (lane & 15)] = acc[e];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < E_SLOTS; ++s) {
      const int idx = tid + s * NTHR;
      float v = 0.f;
      if (idx < MT * NT) {
        v = red[idx];
#pragma unroll
        for (int g = 1; g < P; ++g) v = v + red[g * MT * NT + idx];
      }
      v_out[s] = v;
    }
  } else {
    // single group: route the accumulator through LDS so the store is row-major coalesced
    __syncthreads();
    float* red = lds;
#pragma unroll
    for (int e = 0; e < 4; ++e) red[((lane >> 4) * 4 + e) * NT + wave * 16 + (lane & 15)] = acc[e];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < E_SLOTS; ++s) {
      const int idx = tid + s * NTHR;
      v_out[s] = idx < MT * NT ? red[idx] : 0.f;
    }
  }
#pragma unroll
  for (int s = 0; s < E_SLOTS; ++s) {
    if (e_dst[s] == nullptr) continue;
    float v = v_out[s];
    if constexpr (L::EPI == EPI_BIAS) v = v + e_bias[s];
    if constexpr (L::EPI == EPI_SCALE) v = v * a.scale;
    if constexpr (L::EPI == EPI_ROWSCALE) v = v * e_rs[s];
    if constexpr (L::ACT == ACT_GELU) v = bsp::gelu(v);
    if constexpr (L::RES) v = e_res[s] + v;
    *e_dst[s] = v;
  }
}

template <class L, int LN>
static inline void launch_lat(const char* name, const ConvArgs& a, int n_group_tiles, hipStream_t stream) {
  constexpr int NT = 16 * LN;
  dim3 grid;
  if (L::GROUPED) grid.x = n_group_tiles;
  else grid.x = (a.B * L::T + 15) / 16;
  grid.y = L::NOUT / NT;
  const double M = (double)a.B * L::T, K = (double)L::KSZ * L::CIN, N = L::NOUT;
  const double in_rows = (double)a.B * (L::T * L::STRIDE + (L::KSZ - 1) * L::DIL - (L::STRIDE - 1));
  const bhip::LaunchInfo info{name, 2.0 * M * K * N, 4.0 * (K * N + in_rows * L::CIN + M * N * (L::RES ? 2 : 1))};
  bhip::launch_site(info, stream, [&] { hipLaunchKernelGGL((lat_gemm_kernel<L, LN>), grid, dim3(64 * LN * L::P), 0, stream, a); });
}

