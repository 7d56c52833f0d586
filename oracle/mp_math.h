/*
 * oracle/mp_math.h -- the oracle's OWN statement of the two deterministic
 * functions the mixed-precision definition rests on.  TEST INFRASTRUCTURE ONLY.
 *
 * The mixed-precision likelihood takes log(sigma_D - (sigma_S - eps)), the log
 * of a difference of one float ulp of sigma_S, so the float sigmoid is part of
 * the DEFINITION of the quantity (two different <= 1 ulp expf give values that
 * differ by O(1) per row), and so are the per-block uniforms.  The definition,
 * in words:
 *
 *   expf_det(x): NaN -> NaN; x > 88.72283 -> +inf; x < -87.33654 -> 0.
 *     n = trunc(x*log2(e) +- 0.5) (half away from zero), as float arithmetic;
 *     r = fma(n, 2.12194440e-4, fma(n, -0.693359375, x))     (Cody-Waite, ln 2 in two pieces);
 *     p = sum_{k=0..7} r^k / k!  evaluated by Horner's rule from k = 7 down, one fmaf per step, with the float
 *         constants 1.9841270e-4, 1.3888889e-3, 8.3333333e-3, 4.1666667e-2, 1.6666667e-1, 0.5, 1, 1;
 *     result = (p * 2^(n/2)) * 2^(n - n/2)   (n/2 truncated toward zero; both factors normal floats).
 *   sigmoidf_det(x) = 1 / (1 + expf_det(x)) in float.
 *   mp_uniform(seed, call, block) = (h >> 11) * 2^-53 with
 *     h = mix(mix(mix(seed ^ 0xA0761D6478BD642F) + call) + block),  mix = splitmix64's output function
 *     (add 0x9E3779B97F4A7C15; xor-shift 30, multiply 0xBF58476D1CE4E5B9; xor-shift 27, multiply
 *     0x94D049BB133111EB; xor-shift 31).
 *
 * The product has its own statement of the same definition (mcmc_amd/csrc/mhlik_detmath.h, mhlik_synth.h) for host
 * and device; this file shares no code with it.  tests/test_detmath.py compares the two bit for bit over the float
 * range and over a grid of (seed, call, block), and this one against libm's expf (<= 1 ulp).
 */
#ifndef ORACLE_MP_MATH_H
#define ORACLE_MP_MATH_H

#include <math.h>
#include <stdint.h>

static inline float oracle_pow2f(int e) /* 2^e for -126 <= e <= 127, through the exponent field */
{
  union { uint32_t u; float f; } v;
  v.u = (uint32_t)(e + 127) << 23;
  return v.f;
}

static inline float oracle_expf_det(float x)
{
  static const float coef[8] = {1.0f, 1.0f, 0.5f, 1.6666667e-1f, 4.1666667e-2f, 8.3333333e-3f, 1.3888889e-3f, 1.9841270e-4f};
  if (isnan(x)) return x;
  if (x > 88.72283f) return INFINITY;
  if (x < -87.33654f) return 0.0f;
  const float scaled = x * 1.44269504088896341f;
  const float half = scaled < 0.0f ? -0.5f : 0.5f;
  const int n = (int)(scaled + half);
  const float nf = (float)n;
  const float r = fmaf(nf, 2.12194440e-4f, fmaf(nf, -0.693359375f, x));
  float poly = coef[7];
  for (int k = 6; k >= 0; k--) poly = fmaf(poly, r, coef[k]);
  const int lo = n / 2;
  return (poly * oracle_pow2f(lo)) * oracle_pow2f(n - lo);
}

static inline float oracle_sigmoidf_det(float x) { return 1.0f / (1.0f + oracle_expf_det(x)); }

static inline uint64_t oracle_splitmix_out(uint64_t s)
{
  uint64_t z = s + UINT64_C(0x9E3779B97F4A7C15);
  z ^= z >> 30;
  z *= UINT64_C(0xBF58476D1CE4E5B9);
  z ^= z >> 27;
  z *= UINT64_C(0x94D049BB133111EB);
  z ^= z >> 31;
  return z;
}

static inline double oracle_mp_uniform(uint64_t seed, uint64_t call, uint64_t block)
{
  const uint64_t keyed = oracle_splitmix_out(seed ^ UINT64_C(0xA0761D6478BD642F));
  const uint64_t per_call = oracle_splitmix_out(keyed + call);
  const uint64_t h = oracle_splitmix_out(per_call + block);
  return ldexp((double)(h >> 11), -53);
}

#endif
