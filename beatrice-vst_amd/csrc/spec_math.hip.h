// spec_math.hip.h -- scalar float32 primitives of MODEL_SPEC.md section 2, for gfx950 device code.
//
// Each function is a fixed sequence of IEEE-754 single operations (add/mul/div correctly rounded,
// explicit fused multiply-add, integer bit edits), so results do not depend on compiler
// contraction or fast-math choices; build with -ffp-contract=off.  v_fma_f32, v_rndne_f32 and the
// correctly-rounded f32 divide expansion are what these lower to on CDNA4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsp {

__device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float exp(float x) {
  x = x < -86.0f ? -86.0f : (x > 88.0f ? 88.0f : x);
  const float n = __builtin_rintf(x * 1.44269504088896341f);
  float r = fma(n, -0.693359375f, x);
  r = fma(n, 2.12194440e-4f, r);
  float p = 1.3888889225e-3f;
  p = fma(p, r, 8.3333337680e-3f);
  p = fma(p, r, 4.1666667908e-2f);
  p = fma(p, r, 1.6666667163e-1f);
  p = fma(p, r, 0.5f);
  p = fma(p, r, 1.0f);
  p = fma(p, r, 1.0f);
  return __uint_as_float(__float_as_uint(p) + ((uint32_t)(int32_t)n << 23));
}

__device__ __forceinline__ float sigmoid(float x) { return 1.0f / (1.0f + exp(-x)); }

__device__ __forceinline__ float tanh(float x) {
  const float ax = __builtin_fabsf(x);
  const float e = exp(2.0f * ax);
  const float t = 1.0f - 2.0f / (e + 1.0f);
  return __builtin_copysignf(t, x);
}

__device__ __forceinline__ float gelu(float x) {
  const float x3 = (x * x) * x;
  const float inner = 0.7978845608f * fma(0.044715f, x3, x);
  return (0.5f * x) * (1.0f + tanh(inner));
}

__device__ __forceinline__ float lrelu(float x) { return x > 0.0f ? x : 0.1f * x; }

__device__ __forceinline__ float log(float x) {
  const uint32_t ix = __float_as_uint(x);
  int e = (int)((ix >> 23) & 255u) - 127;
  float m = __uint_as_float((ix & 0x007fffffu) | 0x3f800000u);
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
  const float s = (m - 1.0f) / (m + 1.0f);
  const float z = s * s;
  float p = fma(z, 0.11111111f, 0.14285715f);
  p = fma(p, z, 0.2f);
  p = fma(p, z, 0.33333334f);
  p = fma(p, z, 1.0f);
  return fma((float)e, 0.69314718f, (2.0f * s) * p);
}


// ---- the same functions on PAIRS (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two float32 results per instruction).
// The tick launch is bound by instruction ISSUE next to the MFMAs (profiles/r03_notes.md section 1): gelu() above compiles to ~45
// VALU instructions per element, gelu2() to ~18.  Results are bit-identical to the scalar functions for every finite input
// (BeatriceHip_MathSelfTest sweeps all 2^32 bit patterns on the device; tests/test_gpu_spec_math.py); where they save work:
//   * n = rint(x * log2e) as (m + 1.5 * 2^23) - 1.5 * 2^23: the same round-half-even of the same product m for |m| < 2^22, and
//     the low bits of the intermediate ARE the integer n (its bits << 23 are added to the polynomial's bits: no conversion);
//   * the clamp as v_med3_f32 (a NaN argument has no defined result in MODEL_SPEC -- its conversion to an integer is
//     undefined in the scalar definition as well);
//   * tanh's quotient 2 / (e + 1): e + 1 lies in [2, 2^64) after the clamp below, so the scaling and fix-up steps of the
//     correctly rounded division (v_div_scale, v_div_fmas' scale, v_div_fixup) are identities and the remaining
//     reciprocal + Newton/residual chain -- the very instructions the compiler emits for `2.0f / d` -- runs on pairs;
//     the argument is clamped at 44 instead of 88: from 2|x| >= 18 on, 2 / (e + 1) < 2^-25 and 1 - that IS 1.0f, so every
//     clamp >= 18 gives the same bits.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// exp of a pair whose elements already lie in [-86, 88]
__device__ __forceinline__ f32x2 exp2_clamped(f32x2 x) {
  const f32x2 magic = splat2(12582912.0f);                   // 1.5 * 2^23
  const f32x2 m = x * splat2(1.44269504088896341f);
  const f32x2 t = m + magic;                                 // round-half-even to an integer (one ulp = 1 here)
  const f32x2 n = t - magic;
  f32x2 r = fma2(n, splat2(-0.693359375f), x);
  r = fma2(n, splat2(2.12194440e-4f), r);
  f32x2 p = splat2(1.3888889225e-3f);
  p = fma2(p, r, splat2(8.3333337680e-3f));
  p = fma2(p, r, splat2(4.1666667908e-2f));
  p = fma2(p, r, splat2(1.6666667163e-1f));
  p = fma2(p, r, splat2(0.5f));
  p = fma2(p, r, splat2(1.0f));
  p = fma2(p, r, splat2(1.0f));
  f32x2 o;   // bits(t) = 0x4B400000 + n: shifted left by 23 that is n << 23 (mod 2^32)
  o.x = __uint_as_float(__float_as_uint(p.x) + (__float_as_uint(t.x) << 23));
  o.y = __uint_as_float(__float_as_uint(p.y) + (__float_as_uint(t.y) << 23));
  return o;
}
__device__ __forceinline__ f32x2 exp2(f32x2 x) {
  x.x = __builtin_amdgcn_fmed3f(x.x, -86.0f, 88.0f);
  x.y = __builtin_amdgcn_fmed3f(x.y, -86.0f, 88.0f);
  return exp2_clamped(x);
}
// 2 / d for d in [2, 2^64): the correctly rounded quotient (see above)
__device__ __forceinline__ f32x2 two_over(f32x2 d) {
  f32x2 y;
  y.x = __builtin_amdgcn_rcpf(d.x);
  y.y = __builtin_amdgcn_rcpf(d.y);
  const f32x2 one = splat2(1.0f), two = splat2(2.0f);
  const f32x2 e = fma2(-d, y, one);
  y = fma2(e, y, y);
  f32x2 q = two * y;
  f32x2 r = fma2(-d, q, two);
  q = fma2(r, y, q);
  r = fma2(-d, q, two);
  return fma2(r, y, q);
}
__device__ __forceinline__ f32x2 tanh2(f32x2 x) {
  const f32x2 w = x + x;    // 2 x: exact, so |2 x| = 2 |x|
  f32x2 a;
  a.x = __builtin_amdgcn_fmed3f(__builtin_fabsf(w.x), 0.0f, 44.0f);
  a.y = __builtin_amdgcn_fmed3f(__builtin_fabsf(w.y), 0.0f, 44.0f);
  const f32x2 e = exp2_clamped(a);
  const f32x2 t = splat2(1.0f) - two_over(e + splat2(1.0f));
  f32x2 o;
  o.x = __builtin_copysignf(t.x, x.x);
  o.y = __builtin_copysignf(t.y, x.y);
  return o;
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
  const f32x2 x3 = (x * x) * x;
  const f32x2 inner = splat2(0.7978845608f) * fma2(splat2(0.044715f), x3, x);
  return (splat2(0.5f) * x) * (splat2(1.0f) + tanh2(inner));
}
__device__ __forceinline__ f32x2 sigmoid2(f32x2 x) {   // (the quotient may be denormal here: the full division, per element)
  const f32x2 e = exp2(-x);
  f32x2 o;
  o.x = 1.0f / (1.0f + e.x);
  o.y = 1.0f / (1.0f + e.y);
  return o;
}

// MODEL_SPEC 2.3 "wave sum": xor butterfly over the 64 lanes of one wavefront, offsets 32..1.
// Every lane ends with the identical total (float add is commutative).
__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float wmax64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

}  // namespace bsp
