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
