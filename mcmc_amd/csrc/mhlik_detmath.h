/*
 * mhlik_detmath.h -- a float exp / sigmoid built only from IEEE-exact
 * operations (multiply, add, fma, int<->float conversion, bit casts, one
 * division), so host C and gfx950 device code return the same bits.
 *
 * Why: the mixed-precision path takes log(sigma_D - (sigma_S - eps)) with
 * eps ~ 1e-7, i.e. the log of a difference at the level of ONE float ulp of
 * sigma_S.  Two libm expf implementations that are each within 1 ulp (glibc,
 * ROCm ocml, CUDA) give results that differ by O(1) per affected row, so the
 * float sigmoid has to be the same function on both sides for the CPU
 * restatement to check the kernel at all.  Accuracy: <= 1 ulp against libm
 * expf (tests/test_detmath.py).
 */
#ifndef MHLIK_DETMATH_H
#define MHLIK_DETMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MHLIK_DM __host__ __device__ static inline
#else
#define MHLIK_DM static inline
#endif

MHLIK_DM float mhlik_bits_to_float(uint32_t u)
{
  float f;
  memcpy(&f, &u, sizeof f);
  return f;
}

/* e^x, x float.  Cody-Waite reduction x = n ln2 + r, |r| <= ln2/2, degree-7
 * Horner polynomial in fma form, scaling by 2^n through the exponent bits. */
MHLIK_DM float mhlik_expf_det(float x)
{
  if (x != x) return x;
  if (x > 88.72283f) return mhlik_bits_to_float(0x7f800000u); /* +inf */
  if (x < -87.33654f) return 0.0f;                             /* below the normal range: flushed */
  const float t = x * 1.44269504088896341f;
  const int n = (int)(t + (t < 0.0f ? -0.5f : 0.5f));
  const float fn = (float)n;
  float r = fmaf(fn, -0.693359375f, x);           /* ln2 high part, exact product */
  r = fmaf(fn, 2.12194440e-4f, r);                /* minus the low part */
  float p = 1.9841270e-4f;                        /* 1/5040 */
  p = fmaf(p, r, 1.3888889e-3f);                  /* 1/720 */
  p = fmaf(p, r, 8.3333333e-3f);                  /* 1/120 */
  p = fmaf(p, r, 4.1666667e-2f);                  /* 1/24  */
  p = fmaf(p, r, 1.6666667e-1f);                  /* 1/6   */
  p = fmaf(p, r, 0.5f);
  p = fmaf(p, r, 1.0f);
  p = fmaf(p, r, 1.0f);
  /* 2^n: n in [-126, 128]; split so each factor is a normal float */
  const int n1 = n / 2, n2 = n - n1;
  const float s1 = mhlik_bits_to_float((uint32_t)(n1 + 127) << 23);
  const float s2 = mhlik_bits_to_float((uint32_t)(n2 + 127) << 23);
  return p * s1 * s2;
}

/* 1 / (1 + e^x) in float */
MHLIK_DM float mhlik_sigmoidf_det(float x) { return 1.0f / (1.0f + mhlik_expf_det(x)); }

#endif
