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

#if defined(__HIP_DEVICE_COMPILE__)
/* A coefficient as an opaque scalar-register value: v_fma_f64 takes it as its one scalar operand.  Left to
 * itself the compiler keeps the coefficients in vector registers and forms every Horner step as a copy of the
 * coefficient plus v_fmac_f64 -- a third more instructions for the same bits. */
#define MHLIK_SP_COEF(name, value) \
  double name = value;             \
  __asm__("" : "+s"(name))
#else
#define MHLIK_SP_COEF(name, value) const double name = value
#endif

/* e^-u for u >= 0: x = n ln2 + r, |r| <= ln2/2 (two-part ln2), degree-12 Taylor polynomial, scaled by 2^n with
 * ldexp; < 3e-16 relative; u > 745 underflows to 0. */
MHLIK_SP double mhlik_exp_neg(double u)
{
  MHLIK_SP_COEF(log2e, 1.4426950408889634074);
  MHLIK_SP_COEF(ln2hi, -6.93147180369123816490e-01); /* ln2 high part: fn * hi is exact */
  MHLIK_SP_COEF(ln2lo, -1.90821492927058770002e-10); /* ln2 low part */
  MHLIK_SP_COEF(c12, 2.08767569878680989792e-09);    /* 1/12! */
  MHLIK_SP_COEF(c11, 2.50521083854417187751e-08);
  MHLIK_SP_COEF(c10, 2.75573192239858906526e-07);
  MHLIK_SP_COEF(c9, 2.75573192239858906526e-06);
  MHLIK_SP_COEF(c8, 2.48015873015873015873e-05);
  MHLIK_SP_COEF(c7, 1.98412698412698412698e-04);
  MHLIK_SP_COEF(c6, 1.38888888888888888889e-03);
  MHLIK_SP_COEF(c5, 8.33333333333333333333e-03);
  MHLIK_SP_COEF(c4, 4.16666666666666666667e-02);
  MHLIK_SP_COEF(c3, 1.66666666666666666667e-01);     /* 1/3! */
  const double t = -u;
  const double fn = rint(t * log2e);
  double r = fma(fn, ln2hi, t);
  r = fma(fn, ln2lo, r);
  double p = fma(c12, r, c11);
  p = fma(p, r, c10);
  p = fma(p, r, c9);
  p = fma(p, r, c8);
  p = fma(p, r, c7);
  p = fma(p, r, c6);
  p = fma(p, r, c5);
  p = fma(p, r, c4);
  p = fma(p, r, c3);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)fn); /* t < -745: underflows to 0 */
}

MHLIK_SP double mhlik_neg_softplus(double x)
{
  const double ax = fabs(x);
  const double y = mhlik_exp_neg(!(ax > 800.0) ? ax : 800.0); /* beyond ~1e25 the reduction would lose r; NaN stays NaN */

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
  MHLIK_SP_COEF(a19, 1.0 / 19.0);
  MHLIK_SP_COEF(a17, 1.0 / 17.0);
  MHLIK_SP_COEF(a15, 1.0 / 15.0);
  MHLIK_SP_COEF(a13, 1.0 / 13.0);
  MHLIK_SP_COEF(a11, 1.0 / 11.0);
  MHLIK_SP_COEF(a9, 1.0 / 9.0);
  MHLIK_SP_COEF(a7, 1.0 / 7.0);
  MHLIK_SP_COEF(a5, 1.0 / 5.0);
  MHLIK_SP_COEF(a3, 1.0 / 3.0);
  double q = fma(a19, z2, a17);
  q = fma(q, z2, a15);
  q = fma(q, z2, a13);
  q = fma(q, z2, a11);
  q = fma(q, z2, a9);
  q = fma(q, z2, a7);
  q = fma(q, z2, a5);
  q = fma(q, z2, a3);
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
 * rows of a tile, and  sum_i log(1 + y_i) = log prod_i (1 + y_i):
 *
 *   sum_i s(x_i) = sum_i max(x_i, 0) + log( prod_i (1 + e^-|x_i|) )
 *
 * Every factor lies in (1, 2]: a lane multiplies the 32 of its rows, the 16
 * lane products of a chain are multiplied when the tile is finished (512
 * factors: below 2^512, no rescaling) and ONE log is taken per (tile, chain).
 * Per element that leaves e^-|x| (16 FP64 instructions: mhlik_exp_neg_fast),
 * one fma for  prod += prod * y  and the max/add of the linear part: 19
 * instructions instead of the 45 of mhlik_neg_softplus, with no table and no
 * data-dependent cost (a table-driven log1p was tried: its LDS reads cost more
 * than they save once the dots of a wave spread over many table entries).
 * Error: prod + prod*y is ONE rounding to nearest of prod*(1 + y) (relative
 * 1.1e-16, unbiased), so a product of 32 factors is within ~1e-15 relative
 * and its log within ~1e-15 ABSOLUTE on a value of order 20; the exp is
 * < 5e-16.  That rounding is no worse than the one the reference's own
 * log(1 + exp(x)) makes when it forms 1 + exp(x): 1.1e-16 ABSOLUTE per
 * element (a term with e^-|x| < 2^-53 is dropped, there as here), so a lane's
 * sum is within 2e-15 |sum| + 4e-15 of the exact one (tools/check_softplus.hip)
 * -- relative accuracy for sums of order one, absolute for sums of tiny terms,
 * where the EXACT mode (which keeps tiny terms to relative accuracy) is the
 * better of the two and also differs from the reference's formula.
 * Overflow: the reference's formula gives -inf above x = 709.78.  The adds do
 * not test for it, nor for arguments so large that the range reduction breaks
 * down; a group whose linear part exceeds that value or whose product has
 * left its range is redone with mhlik_softplus_acc_add_safe (the kernel does
 * so while the elements are still in registers).
 * tests/test_softplus.py states a lane's work on the host and measures
 * accuracy and bias against 80-bit sums.
 * -------------------------------------------------------------------------*/
#define MHLIK_SOFTPLUS_OVERFLOW 709.782712893383973096

/* e^t for any t: as mhlik_exp_neg with a degree-10 minimax polynomial (c0 = c1 = 1; relative error 2.9e-16
 * equi-rippled over |r| <= ln2/2, from a Remez exchange in 60-digit arithmetic) instead of the degree-12 Taylor
 * one; < 5e-16 relative with the roundings; underflows to 0 below -745, +inf above 709.8, NaN stays NaN. */
MHLIK_SP double mhlik_exp_fast(double t)
{
  MHLIK_SP_COEF(log2e, 1.4426950408889634074);
  MHLIK_SP_COEF(ln2hi, -6.93147180369123816490e-01);
  MHLIK_SP_COEF(ln2lo, -1.90821492927058770002e-10);
  MHLIK_SP_COEF(c10, 0x1.2707a75412ea7p-22);
  MHLIK_SP_COEF(c9, 0x1.72e91b12ab2d6p-19);
  MHLIK_SP_COEF(c8, 0x1.a01b7c4e4defdp-16);
  MHLIK_SP_COEF(c7, 0x1.a0198d583e339p-13);
  MHLIK_SP_COEF(c6, 0x1.6c16c0c8318e3p-10);
  MHLIK_SP_COEF(c5, 0x1.11111125b3ec8p-7);
  MHLIK_SP_COEF(c4, 0x1.55555555890c1p-5);
  MHLIK_SP_COEF(c3, 0x1.55555555507c5p-3);
  MHLIK_SP_COEF(c2, 0x1.ffffffffffed2p-2);
  const double fn = rint(t * log2e);
  double r = fma(fn, ln2hi, t);
  r = fma(fn, ln2lo, r);
  double p = fma(c10, r, c9);
  p = fma(p, r, c8);
  p = fma(p, r, c7);
  p = fma(p, r, c6);
  p = fma(p, r, c5);
  p = fma(p, r, c4);
  p = fma(p, r, c3);
  p = fma(p, r, c2);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
#if defined(__HIP_DEVICE_COMPILE__)
  return ldexp(p, (int)fn); /* v_cvt_i32_f64 saturates (NaN -> 0): p 2^n is 0, +inf or NaN as it should be */
#else
  return ldexp(p, fn != fn ? 0 : fn > 4000.0 ? 4000 : fn < -4000.0 ? -4000 : (int)fn);
#endif
}

/* e^-u for u >= 0, the product form's factor */
MHLIK_SP double mhlik_exp_neg_fast(double u) { return mhlik_exp_fast(-u); }

typedef struct {
  double lin;  /* sum of max(x, 0); +inf once an x above 709.78 was flagged (the reference's formula overflows there) */
  double prod; /* product of (1 + e^-|x|) */
} mhlik_softplus_acc;

MHLIK_SP void mhlik_softplus_acc_init(mhlik_softplus_acc *a)
{
  a->lin = 0.0;
  a->prod = 1.0;
}

MHLIK_SP void mhlik_softplus_acc_add(mhlik_softplus_acc *a, double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  double pos; /* max(x, 0) in one instruction: fmax() adds a canonicalising v_max_f64 x, x in front; NaN -> 0 either way */
  __asm__("v_max_f64 %0, %1, 0" : "=v"(pos) : "v"(x));
  a->lin += pos;
#else
  a->lin += x > 0.0 ? x : 0.0;
#endif
  a->prod = fma(a->prod, mhlik_exp_neg_fast(fabs(x)), a->prod);
}

/* The same, with the tests the fast add leaves out: +inf in lin above the reference's overflow point, and no
 * trip through the range reduction for |x| beyond the underflow point (from ~1e25 on its r is garbage and the
 * polynomial overflows).  A group is redone with this from its saved state when mhlik_softplus_acc_suspect()
 * says so -- the kernel does that per pass, while the elements are still in the accumulators. */
MHLIK_SP void mhlik_softplus_acc_add_safe(mhlik_softplus_acc *a, double x)
{
  a->lin += x > MHLIK_SOFTPLUS_OVERFLOW ? HUGE_VAL : (x > 0.0 ? x : 0.0);
  const double ax = fabs(x);
  const double y = ax <= 745.2 ? mhlik_exp_neg_fast(ax) : (ax != ax ? ax : 0.0);
  a->prod = fma(a->prod, y, a->prod);
}

/* after at most 32 fast adds to a state that was not suspect: lin may have passed an x above the overflow
 * point, or prod has left [1, 2^40] (an infinite or NaN factor) */
MHLIK_SP int mhlik_softplus_acc_suspect(const mhlik_softplus_acc *a)
{
  return !(a->lin <= MHLIK_SOFTPLUS_OVERFLOW) || !(a->prod <= 0x1p40);
}

/* a += b: the elements of both groups (at most ~1000 elements altogether: the product must stay finite) */
MHLIK_SP void mhlik_softplus_acc_merge(mhlik_softplus_acc *a, const mhlik_softplus_acc *b)
{
  a->lin += b->lin;
  a->prod *= b->prod;
}

/* - sum of log(1 + e^x) over everything added */
MHLIK_SP double mhlik_softplus_acc_neg_sum(const mhlik_softplus_acc *a) { return -(a->lin + log(a->prod)); }

#endif
