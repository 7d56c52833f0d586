/*
 * mhlik_softplus.h -- -log(1 + e^x) in double with about half the vector
 * instructions of exp() followed by log().
 *
 * Where it is used: the epilogue of the K-chain double kernel
 * (mhlik_batched.hpp).  There every (row, chain) element costs one exp and one
 * log on the same FP64 lanes the f64 MFMA runs on; with ocml's exp and log the
 * epilogue took 44 % of the kernel (tools/batched_lab.hip, "no transcendental
 * epilogue").  The single-theta kernel keeps -log(1 + exp(x)) as the
 * reference writes it: it is HBM-bound and its per-row terms match the
 * oracle's libm terms to the last place or two.
 *
 *   s(x) = log(1 + e^x) = max(x, 0) + log1p(y),   y = e^{-|x|} in (0, 1]
 *   y    : x = n ln2 + r, |r| <= ln2/2 (two-part ln2), degree-12 Taylor
 *          polynomial, scaled by 2^n with ldexp
 *   log1p: w = 1 + y in (1, 2]; halve w above sqrt(2) (adds ln2);
 *          log w = 2 atanh(z), z = (w-1)/(w+1) formed from y without
 *          cancellation (y/(y+2) or (y-1)/(y+3)), |z| <= 0.1716, nine terms
 * Error: < 6e-16 relative to s(x) for x >= -708, absolute below where the
 * result is a denormal (tests/test_softplus.py, tools/check_softplus.hip),
 * far inside the 1e-10 the summed log-likelihood has to meet.  Like the
 * reference's formula the result is -inf above x = 709.78 (exp overflows
 * there, src/mcmc_cpu/mcmc_cpu.c:440).
 */
#ifndef MHLIK_SOFTPLUS_H
#define MHLIK_SOFTPLUS_H

#include <math.h>

#if defined(__HIPCC__)
#define MHLIK_SP __host__ __device__ static inline
#else
#define MHLIK_SP static inline
#endif

/* e^-u for u >= 0: x = n ln2 + r, |r| <= ln2/2 (two-part ln2), degree-12 Taylor polynomial, scaled by 2^n with
 * ldexp; < 3e-16 relative; u > 745 underflows to 0. */
MHLIK_SP double mhlik_exp_neg(double u)
{
  const double t = -u;
  const double fn = rint(t * 1.4426950408889634074);
  double r = fma(fn, -6.93147180369123816490e-01, t); /* ln2 high part: fn * hi is exact */
  r = fma(fn, -1.90821492927058770002e-10, r);        /* ln2 low part */
  double p = 2.08767569878680989792e-09;              /* 1/12! */
  p = fma(p, r, 2.50521083854417187751e-08);          /* 1/11! */
  p = fma(p, r, 2.75573192239858906526e-07);          /* 1/10! */
  p = fma(p, r, 2.75573192239858906526e-06);          /* 1/9!  */
  p = fma(p, r, 2.48015873015873015873e-05);          /* 1/8!  */
  p = fma(p, r, 1.98412698412698412698e-04);          /* 1/7!  */
  p = fma(p, r, 1.38888888888888888889e-03);          /* 1/6!  */
  p = fma(p, r, 8.33333333333333333333e-03);          /* 1/5!  */
  p = fma(p, r, 4.16666666666666666667e-02);          /* 1/4!  */
  p = fma(p, r, 1.66666666666666666667e-01);          /* 1/3!  */
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)fn); /* t < -745: underflows to 0 */
}

MHLIK_SP double mhlik_neg_softplus(double x)
{
  const double y = mhlik_exp_neg(fabs(x));

  /* log1p(y), y in [0, 1] */
  const int big = y > 0.41421356237309504880; /* 1 + y > sqrt(2) */
  const double num = big ? y - 1.0 : y;
  const double den = big ? y + 3.0 : y + 2.0;
#if defined(__HIP_DEVICE_COMPILE__)
  /* den in [2, 4]: reciprocal by two Newton steps, quotient corrected once */
  double rc = __builtin_amdgcn_rcp(den);
  rc = fma(fma(-den, rc, 1.0), rc, rc);
  rc = fma(fma(-den, rc, 1.0), rc, rc);
  double z = num * rc;
  z = fma(fma(-den, z, num), rc, z);
#else
  const double z = num / den;
#endif
  const double z2 = z * z;
  double q = 1.0 / 19.0;
  q = fma(q, z2, 1.0 / 17.0);
  q = fma(q, z2, 1.0 / 15.0);
  q = fma(q, z2, 1.0 / 13.0);
  q = fma(q, z2, 1.0 / 11.0);
  q = fma(q, z2, 1.0 / 9.0);
  q = fma(q, z2, 1.0 / 7.0);
  q = fma(q, z2, 1.0 / 5.0);
  q = fma(q, z2, 1.0 / 3.0);
  q = q * z2;                           /* atanh(z)/z - 1 */
  double l = fma(z + z, q, z + z);      /* 2 atanh(z) */
  l += big ? 6.93147180559945309417e-01 : 0.0;
  const double s = (x > 0.0 ? x : 0.0) + l;
  return x > 709.782712893383973096 ? -HUGE_VAL : -s;
}

/* ---------------------------------------------------------------------------
 * Product form: the default epilogue of the K-chain double kernel
 * (mhlik_set_batched_f64_mode(FAST)).
 *
 * What the kernel needs is not s(x_i) for every element but the SUM over the
 * rows a lane holds, and  sum_i log(1 + y_i) = log prod_i (1 + y_i):
 *
 *   sum_i s(x_i) = sum_i max(x_i, 0) + log( prod_i (1 + e^-|x_i|) )
 *
 * Every factor lies in (1, 2], a lane multiplies 32 of them per (tile, chain
 * block) -- the product stays below 2^32, no rescaling -- and takes ONE log at
 * the end.  Per element that leaves e^-|x| (18 FP64 instructions), one add, one
 * multiply and the max/add of the linear part: 22 instructions instead of 45,
 * with no table and no data-dependent cost (a table-driven log1p was tried: its
 * LDS reads cost more than they save once the dots of a wave spread over many
 * table entries).  Error: each factor and each multiplication is rounded to
 * nearest (relative 1.1e-16, unbiased), so a product of 32 factors is within
 * ~1e-15 relative and its log within ~1e-15 ABSOLUTE on a value of order 20;
 * the exp is < 3e-16.  The rounding of each factor 1 + y is the rounding the
 * reference's own log(1 + exp(x)) makes: 1.1e-16 ABSOLUTE per element (a term
 * with e^-|x| < 2^-53 is dropped, there as here), so a lane's sum is within
 * 2e-15 |sum| + 4e-15 of the exact one (tools/check_softplus.hip) -- relative
 * accuracy for sums of order one, absolute for sums of tiny terms, where the
 * EXACT mode (which keeps tiny terms to relative accuracy) is the better of
 * the two and also differs from the reference's formula.
 * mhlik_softplus_group_sum is the host statement of what a lane does (used by
 * tests/test_softplus.py to measure accuracy and bias against 80-bit sums).
 * -------------------------------------------------------------------------*/
typedef struct {
  double lin;  /* sum of max(x, 0); +inf once an x above 709.78 was seen (the reference's formula overflows there) */
  double prod; /* product of (1 + e^-|x|) */
} mhlik_softplus_acc;

MHLIK_SP void mhlik_softplus_acc_init(mhlik_softplus_acc *a)
{
  a->lin = 0.0;
  a->prod = 1.0;
}

MHLIK_SP void mhlik_softplus_acc_add(mhlik_softplus_acc *a, double x)
{
  a->lin += x > 709.782712893383973096 ? HUGE_VAL : (x > 0.0 ? x : 0.0);
  a->prod *= 1.0 + mhlik_exp_neg(fabs(x));
}

/* - sum of log(1 + e^x) over everything added (at most ~1000 elements: the product must stay finite) */
MHLIK_SP double mhlik_softplus_acc_neg_sum(const mhlik_softplus_acc *a) { return -(a->lin + log(a->prod)); }

#endif
