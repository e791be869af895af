/*
 * oracle/spec_math.h -- TEST INFRASTRUCTURE (CPU oracle).  Never include from product code.
 *
 * Scalar float32 primitives of MODEL_SPEC.md section 2.  The closed reference library uses its
 * own approximations ("fmath", reference LICENSES_BUNDLED.txt:13-27) which are not available;
 * the spec therefore fixes its own, written only with IEEE-754 single-precision add / mul / div
 * (correctly rounded), fused multiply-add and integer bit manipulation, so that a CPU and a GPU
 * implementation that follow the same operation order agree bit for bit.
 * Compile with -ffp-contract=off: every fusion in the spec is an explicit sp_fma().
 */
#ifndef ORACLE_SPEC_MATH_H_
#define ORACLE_SPEC_MATH_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float sp_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline float sp_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t sp_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp: clamp, n = rint(x*log2e), two-step Cody-Waite reduction, degree-6 Horner, scale by 2^n. */
static inline float sp_exp(float x) {
  x = x < -86.0f ? -86.0f : (x > 88.0f ? 88.0f : x);
  const float n = rintf(x * 1.44269504088896341f);
  float r = sp_fma(n, -0.693359375f, x);
  r = sp_fma(n, 2.12194440e-4f, r);
  float p = 1.3888889225e-3f;
  p = sp_fma(p, r, 8.3333337680e-3f);
  p = sp_fma(p, r, 4.1666667908e-2f);
  p = sp_fma(p, r, 1.6666667163e-1f);
  p = sp_fma(p, r, 0.5f);
  p = sp_fma(p, r, 1.0f);
  p = sp_fma(p, r, 1.0f);
  return sp_from_bits(sp_bits(p) + ((uint32_t)(int32_t)n << 23));
}

static inline float sp_sigmoid(float x) { return 1.0f / (1.0f + sp_exp(-x)); }

static inline float sp_tanh(float x) {
  const float ax = fabsf(x);
  const float e = sp_exp(2.0f * ax);
  const float t = 1.0f - 2.0f / (e + 1.0f);
  return copysignf(t, x);
}

/* tanh-form GELU */
static inline float sp_gelu(float x) {
  const float x3 = (x * x) * x;
  const float inner = 0.7978845608f * sp_fma(0.044715f, x3, x);
  return (0.5f * x) * (1.0f + sp_tanh(inner));
}

static inline float sp_lrelu(float x) { return x > 0.0f ? x : 0.1f * x; }

/* log for normal positive x: split exponent, m in (sqrt(.5), sqrt(2)], 2*atanh series. */
static inline float sp_log(float x) {
  const uint32_t ix = sp_bits(x);
  int e = (int)((ix >> 23) & 255u) - 127;
  float m = sp_from_bits((ix & 0x007fffffu) | 0x3f800000u);
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
  const float s = (m - 1.0f) / (m + 1.0f);
  const float z = s * s;
  float p = sp_fma(z, 0.11111111f, 0.14285715f);
  p = sp_fma(p, z, 0.2f);
  p = sp_fma(p, z, 0.33333334f);
  p = sp_fma(p, z, 1.0f);
  return sp_fma((float)e, 0.69314718f, (2.0f * s) * p);
}

/* "wave sum": 64 partials combined by an xor butterfly (offsets 32,16,...,1); every position
 * ends with the same value because float addition is commutative.  MODEL_SPEC section 2.3. */
static inline float sp_wsum64(const float* part) {
  float a[64], b[64];
  memcpy(a, part, sizeof(a));
  for (int off = 32; off >= 1; off >>= 1) {
    for (int l = 0; l < 64; ++l) b[l] = a[l] + a[l ^ off];
    memcpy(a, b, sizeof(a));
  }
  return a[0];
}

#endif /* ORACLE_SPEC_MATH_H_ */
