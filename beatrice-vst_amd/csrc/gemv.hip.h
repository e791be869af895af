// gemv.hip.h -- the same layers as conv_gemm.hip.h for a HANDFUL of rows (the 1-stream C-ABI: one stream, one hop).
//
// With one stream a layer has 1-8 rows; an MFMA tile would be 15/16 empty.  Here every output element's reduction
// segment is ONE LANE's k-ascending FMA chain -- v_fma_f32 rounds exactly like the f32 MFMA
// (one rounding per multiply-add), so the bits are those of MODEL_SPEC 2.2 -- with the weights streamed straight from
// the packed blob (the four float4 of a k-block per lane, coalesced across the 16 columns of a tile), the input rows
// broadcast from LDS, and the segments of an output added in order at the end.  A workgroup owns one column tile of 16
// and all (row, segment) pairs of it; workgroup b of every launch lands on XCD b % 8, so a layer's column tiles -- and
// their weights -- live in the same XCD's L2 from hop to hop.  Used where it measures faster than the few-row MFMA tiling:
// layers with three or more reduction segments (launch_auto, conv_gemm.hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include "conv_gemm.hip.h"

namespace gemv {

constexpr int NTHR = 256, PAIRS_PER_PASS = NTHR / 16;

template <class L>
__device__ __forceinline__ void gemv_body(const ConvArgs& a, const int nt, float* __restrict__ lds) {
  constexpr int K = L::K, P = L::P, T = L::T;
  constexpr int LAST = K - 256 * (P - 1);
  static_assert(!L::GROUPED && K % 16 == 0 && LAST % 16 == 0, "layer shape");
  const int tid = threadIdx.x, j = tid & 15, pr = tid >> 4;
  const int M = a.B * T;                       // rows of this launch (small)
  const int pairs = M * P;
  float* xs = lds;                             // [M][K] input rows (pre-activation applied)
  float* part = lds + M * K;                   // [pairs][16] segment results
  const int hop = stepc::step(a.hop);
  if (hop < 0) return;
  const int pos_in = ring_pos(a.in, hop);
  for (int e = tid; e < M * (K / 4); e += NTHR) {
    const int m = e / (K / 4), kk = (e % (K / 4)) * 4;
    const int b = m / T, t = m % T;
    const int tap = kk / L::CIN, c = kk % L::CIN;
    float4 v = *reinterpret_cast<const float4*>(ring_frame(a.in, b, pos_in, (t + 1) * L::STRIDE - 1 - (L::KSZ - 1 - tap) * L::DIL + a.rel_shift) + c);
    if constexpr (L::PRE == PRE_LRELU) { v.x = bsp::lrelu(v.x); v.y = bsp::lrelu(v.y); v.z = bsp::lrelu(v.z); v.w = bsp::lrelu(v.w); }
    *reinterpret_cast<float4*>(xs + m * K + kk) = v;
  }
  __syncthreads();
  const float4* wt = reinterpret_cast<const float4*>(a.w) + (size_t)nt * (K / 16) * 64 + j;  // + kb * 64 + kq * 16
  for (int p0 = 0; p0 < pairs; p0 += PAIRS_PER_PASS) {
    const int p = p0 + pr;
    if (p < pairs) {
      const int m = p / P, s = p % P;
      const int kbs = (s + 1 < P ? 256 : LAST) / 16;
      const float4* w = wt + (size_t)s * 16 * 64;
      const float* x = xs + m * K + s * 256;
      float acc = 0.0f;
      // the segment's weights in two halves of up to eight k-blocks, every load of a half issued before its first FMA
      // (the lane is alone with its chain: what counts is how few memory round trips sit on it, not registers)
      for (int kb0 = 0; kb0 < kbs; kb0 += 8) {
        float4 f[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) f[u][q] = kb0 + u < kbs ? w[(size_t)(kb0 + u) * 64 + q * 16] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (kb0 + u < kbs) {
            const float* xk = x + (kb0 + u) * 16;
            // k = kb*16 + 4e + q: element e of the float4 of k-quad q
            acc = bsp::fma(xk[0], f[u][0].x, acc); acc = bsp::fma(xk[1], f[u][1].x, acc); acc = bsp::fma(xk[2], f[u][2].x, acc); acc = bsp::fma(xk[3], f[u][3].x, acc);
            acc = bsp::fma(xk[4], f[u][0].y, acc); acc = bsp::fma(xk[5], f[u][1].y, acc); acc = bsp::fma(xk[6], f[u][2].y, acc); acc = bsp::fma(xk[7], f[u][3].y, acc);
            acc = bsp::fma(xk[8], f[u][0].z, acc); acc = bsp::fma(xk[9], f[u][1].z, acc); acc = bsp::fma(xk[10], f[u][2].z, acc); acc = bsp::fma(xk[11], f[u][3].z, acc);
            acc = bsp::fma(xk[12], f[u][0].w, acc); acc = bsp::fma(xk[13], f[u][1].w, acc); acc = bsp::fma(xk[14], f[u][2].w, acc); acc = bsp::fma(xk[15], f[u][3].w, acc);
          }
        }
      }
      part[p * 16 + j] = acc;
    }
  }
  __syncthreads();
  // epilogue: segments in order, then conv_gemm's epilogue operation for operation
  const int pos_out = ring_pos(a.out, hop), R_out = a.out.n * a.out.m;
  int pos_res = 0, R_res = 0;
  if constexpr (L::RES) { pos_res = ring_pos(a.res, hop); R_res = a.res.n * a.res.m; }
  for (int e = tid; e < M * 16; e += NTHR) {
    const int m = e >> 4, jj = e & 15, n = nt * 16 + jj;
    const int b = m / T, t = m % T;
    float v = part[(m * P) * 16 + jj];
#pragma unroll
    for (int s = 1; s < P; ++s) v = v + part[(m * P + s) * 16 + jj];
    if constexpr (L::EPI == EPI_BIAS) v = v + a.bias[n];
    if constexpr (L::EPI == EPI_SCALE) v = v * a.scale;
    if constexpr (L::EPI == EPI_ROWSCALE) v = v * a.rowscale[b * T + t];
    if constexpr (L::ACT == ACT_GELU) v = bsp::gelu(v);
    if constexpr (L::RES) v = a.res.base[((size_t)b * R_res + pos_res) * a.res.C + (size_t)t * L::NOUT + n] + v;
    a.out.base[((size_t)b * R_out + pos_out) * a.out.C + (size_t)t * L::NOUT + n] = v;
  }
}

template <class L>
static __global__ __launch_bounds__(NTHR) void gemv_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  gemv_body<L>(a, blockIdx.x, lds);
}

// rows a launch may have: its (row, segment) pairs are walked 16 at a time, the input rows sit in LDS
template <class L>
constexpr int max_rows() { return (48 * 1024 / 4 - 64) / (L::K + L::P * 16); }

template <class L>
static inline void launch(const char* name, const ConvArgs& a, hipStream_t stream) {
  const int M = a.B * L::T;
  const size_t lds = sizeof(float) * ((size_t)M * L::K + (size_t)M * L::P * 16);
  bhip::launch_site(ConvOp<L, TileCfg<1, 1, 1, 2, 1>>::info(name, a), stream,
                    [&] { hipLaunchKernelGGL((gemv_kernel<L>), dim3(L::NOUT / 16), dim3(NTHR), lds, stream, a); });
}

}  // namespace gemv
