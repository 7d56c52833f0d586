/*
 * oracle/mp_path.c -- CPU restatement of the mixed-precision likelihood.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  "Parity unpinned".
 *
 * The reference's mixed-precision path (src/mcmc_mp/mcmc_mp.cu:28-200 kernel,
 * :475-552 host) is work in progress: its re-examined-dark branch takes the
 * log of a log (:63-65), and the per-block uniforms come from cuRAND in the
 * order blocks win an atomic counter (:86-87, :180-181).  What is restated
 * here is the algorithm that code is evidently after, with the same per-row
 * arithmetic it spells out:
 *   dotS  float row dot (cublasSgemv :515)        dotD  double row dot (:55,:137)
 *   B_i = (1/(1+expf(dotS))) - e   [float sigmoid, double subtraction :63,:146]
 *   L_i = 1/(1+exp(dotD))                                               [:147]
 *   bright block : B_i stored as float (sLDblock); value  sum log(L_i - B_i) [:149]
 *                  resample exponent sum( logf(B_i) - log(L_i) )            [:148]
 *   dark block   : value sum (float)log(B_i), float accumulator        [:104]
 *   re-examined dark block (b % ResampFactor == ResampChoose): value as dark,
 *                  exponent sum( log(B_i) - log(L_i) )
 *   z' = (1 - exp(exponent) <= u_b) ? dark : bright                   [:185-193]
 *   result = sum over blocks bright at entry (double)
 *          + sum over blocks dark at entry (float)                     [:546]
 * u_b = oracle_mp_uniform(seed, call, b) and the float sigmoid is
 * oracle_sigmoidf_det (oracle/mp_math.h): part of the DEFINITION of the
 * quantity, because log(L_i - B_i) is the log of a one-float-ulp difference
 * and two different <=1-ulp expf implementations move it by O(1) per row.
 * oracle/mp_math.h is the oracle's own statement of that definition; the
 * product's (mhlik_detmath.h, mhlik_synth.h) shares no code with it, and
 * tests/test_detmath.py compares the two bit for bit and this one with libm.
 */
#include "oracle.h"

#include <math.h>
#include <stdint.h>

#include "mp_math.h" /* the oracle's own float sigmoid and per-block uniform; no product header in this file */

double oracle_mp_likelihood(const double *Xd, const float *Xf, long N, int d, long lda, const double *theta,
                            int block_rows, int blocks, int resamp_factor, int resamp_choose, uint64_t u_seed,
                            uint64_t u_call, double eps, int8_t *z, int *bright_out, int *dark_out)
{
  double red_d = 0.0;
  float red_f = 0.0f;
  int nb = 0;
  for (long b = 0; b < blocks; b++) {
    const long r0 = b * (long)block_rows;
    if (r0 >= N) { /* block holds no rows: keeps its indicator, contributes nothing */
      nb += z[b] != 0;
      continue;
    }
    const long r1 = (r0 + block_rows < N) ? r0 + block_rows : N;
    const int was_bright = z[b] != 0;
    const int re = !was_bright && (b % resamp_factor == resamp_choose);
    float ld = 0.0f;
    double lb = 0.0, rs = 0.0;
    for (long i = r0; i < r1; i++) {
      float ds = 0.0f;
      for (int j = 0; j < d; j++) ds = fmaf((float)theta[j], Xf[(long)j * lda + i], ds);
      const float sig_s = oracle_sigmoidf_det(ds);
      const double Bd = (double)sig_s - eps;
      if (was_bright || re) {
        double dd = 0.0;
        for (int j = 0; j < d; j++) dd += theta[j] * Xd[(long)j * lda + i]; /* -ffp-contract=off */
        const double L = 1.0 / (1.0 + exp(dd));
        if (was_bright) {
          const float Bf = (float)Bd;
          rs += (double)logf(Bf) - log(L);
          lb += log(L - (double)Bf);
        } else {
          ld += (float)log(Bd);
          rs += log(Bd) - log(L);
        }
      } else {
        ld += (float)log(Bd);
      }
    }
    if (was_bright)
      red_d += lb;
    else
      red_f += ld;
    if (was_bright || re) {
      const double u = oracle_mp_uniform(u_seed, u_call, (uint64_t)b);
      z[b] = ((1.0 - exp(rs)) <= u) ? 0 : 1;
    }
    nb += z[b] != 0;
  }
  if (bright_out) *bright_out = nb;
  if (dark_out) *dark_out = blocks - nb;
  return red_d + red_f;
}

/* likelihood_bound / call_boundKernel (mcmc_mp.cu:746-821, kernels :202-280):
 * sum of the double sigmoids; *epsilon = min_i (sigma_D - sigma_S).  (The
 * reference's pairwise "keep the smaller if I am negative" rule, :229-230, is
 * order dependent; the plain minimum is its intent.) */
double oracle_mp_bound(const double *Xd, const float *Xf, long N, int d, long lda, const double *theta,
                       double *epsilon)
{
  double s = 0.0, mn = INFINITY;
  for (long i = 0; i < N; i++) {
    float ds = 0.0f;
    double dd = 0.0;
    for (int j = 0; j < d; j++) {
      ds = fmaf((float)theta[j], Xf[(long)j * lda + i], ds);
      dd += theta[j] * Xd[(long)j * lda + i];
    }
    const double L = 1.0 / (1.0 + exp(dd));
    const double S = (double)oracle_sigmoidf_det(ds);
    s += L;
    if (L - S < mn) mn = L - S;
  }
  *epsilon = N > 0 ? mn : 0.0;
  return s;
}
