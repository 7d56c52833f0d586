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

/* ---- the callers: mp_sampler's chain, get_lowerbound, get_blockSz ----------
 *
 * Restated from src/mcmc_mp/mcmc_mp.cu: burn_in_metropolis_mp :420-473 and
 * metropolis_mp :368-418 (chain starts at theta = 1 whatever burn[0] holds,
 * :430-435; prior N(0, 5^2), :16; accept iff u <= exp(log ratio), :402,:635;
 * per iteration: d ziggurat draws, gsl_rng_uniform_int for the re-examination
 * phase inside the likelihood (:504), then one uniform), get_lowerbound
 * :652-744 (starts at 0, 50000 iterations, keeps the largest per-theta
 * epsilon), get_blockSz :824-943 with short_run_mp :945-995 (block sizes
 * 64..1024, ResampFactor = blocks/10, /100, ... while >= 10; 500-iteration
 * runs from theta = 1, all blocks bright at the start of each).  The
 * per-block uniforms are oracle_mp_uniform(1234, call, block) with `call`
 * counting the likelihood evaluations of the current chain from 0.
 */
#include <gsl/gsl_randist.h>
#include <gsl/gsl_rng.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  const double *Xd;
  const float *Xf;
  long N, lda;
  int d;
  int block_rows, blocks, resamp_factor;
  double eps;
  int8_t *z;
  uint64_t calls;
  int bright, dark;
  double last_epsilon; /* lower-bound search */
  gsl_rng *r;
} mp_chain;

static double mp_prior5(const double *th, int d)
{
  double s = 0.0;
  for (int j = 0; j < d; j++) s += log(gsl_ran_gaussian_pdf(th[j], 5.0));
  return s;
}

static double mp_chain_lik(mp_chain *c, const double *th)
{
  const int phase = (int)gsl_rng_uniform_int(c->r, (unsigned long)c->resamp_factor);
  return oracle_mp_likelihood(c->Xd, c->Xf, c->N, c->d, c->lda, th, c->block_rows, c->blocks, c->resamp_factor, phase,
                              1234u, c->calls++, c->eps, c->z, &c->bright, &c->dark);
}

static double mp_chain_bound(mp_chain *c, const double *th)
{
  return oracle_mp_bound(c->Xd, c->Xf, c->N, c->d, c->lda, th, &c->last_epsilon);
}

/* one iteration; `bound` selects the lower-bound search's "likelihood" */
static int mp_iteration(mp_chain *c, double sd, double *cur, double *prop, double *cpost, int bound,
                        oracle_trace_rec *tr, long cap, long *ntr)
{
  for (int j = 0; j < c->d; j++) prop[j] = cur[j] + gsl_ran_gaussian_ziggurat(c->r, sd);
  const double pl = bound ? mp_chain_bound(c, prop) : mp_chain_lik(c, prop);
  const double ppost = mp_prior5(prop, c->d) + pl;
  const double a = exp(ppost - *cpost);
  const double u = gsl_rng_uniform(c->r);
  const int acc = u <= a;
  if (tr && *ntr < cap) {
    tr[*ntr].lik = pl;
    tr[*ntr].ratio = a;
    tr[*ntr].logu = u;
    tr[*ntr].accept = acc;
  }
  if (ntr) ++*ntr;
  if (acc) {
    memcpy(cur, prop, sizeof(double) * (size_t)c->d);
    *cpost = ppost;
  }
  return acc;
}

/* get_lowerbound (mcmc_mp.cu:652-744); `reps` is LOWERBOUND_REPS (50000, :23) */
double oracle_get_lowerbound(const double *Xd, const float *Xf, long N, int d, long lda, gsl_rng *r, double rwsd,
                             int reps, oracle_trace_rec *tr, long cap, long *ntr_out)
{
  mp_chain c = {Xd, Xf, N, lda, d, 0, 0, 1, 0.0, NULL, 0, 0, 0, 0.0, r};
  double *cur = calloc((size_t)d, sizeof(double)), *prop = calloc((size_t)d, sizeof(double));
  long ntr = 0;
  double cpost = mp_prior5(cur, d) + mp_chain_bound(&c, cur); /* theta = 0 */
  double best = c.last_epsilon;
  for (int i = 1; i < reps; i++) {
    mp_iteration(&c, rwsd, cur, prop, &cpost, 1, tr, cap, &ntr);
    if (best < c.last_epsilon) best = c.last_epsilon;
  }
  free(cur);
  free(prop);
  if (ntr_out) *ntr_out = ntr;
  return best;
}

/* The chains get_blockSz times (mcmc_mp.cu:824-943, short_run_mp :945-995), one after the other on one generator.
 * Which configuration wins is a matter of wall time and not restated; what is: the list of configurations tried
 * (cfg_out: block size, blocks, ResampFactor triples) and every iteration of every short run (trace). */
int oracle_get_blockSz_runs(const double *Xd, const float *Xf, long N, int d, long lda, gsl_rng *r, double rwsd,
                            double bound_eps, int *cfg_out, int cfg_cap, oracle_trace_rec *tr, long cap, long *ntr_out)
{
  int ncfg = 0;
  long ntr = 0;
  double *cur = calloc((size_t)d, sizeof(double)), *prop = calloc((size_t)d, sizeof(double));
  for (int bs = 64; bs <= 1024; bs *= 2) {
    const int blocks = (int)(N / bs);
    if (blocks < 1) break;
    int8_t *z = malloc((size_t)blocks);
    for (int f = blocks / 10; f >= 10; f /= 10) {
      memset(z, 1, (size_t)blocks);
      mp_chain c = {Xd, Xf, N, lda, d, bs, blocks, f, bound_eps, z, 0, blocks, 0, 0.0, r};
      for (int j = 0; j < d; j++) cur[j] = 1.0;
      double cpost = mp_chain_lik(&c, cur);
      cpost += mp_prior5(cur, d);
      for (int i = 1; i < 500; i++) mp_iteration(&c, rwsd, cur, prop, &cpost, 0, tr, cap, &ntr);
      if (ncfg < cfg_cap) {
        cfg_out[3 * ncfg] = bs;
        cfg_out[3 * ncfg + 1] = blocks;
        cfg_out[3 * ncfg + 2] = f;
      }
      ncfg++;
    }
    free(z);
  }
  free(cur);
  free(prop);
  if (ntr_out) *ntr_out = ntr;
  return ncfg;
}

/* mp_sampler (mcmc_mp.cu:283-366): burn-in rows 1..burnin-1 then Ns samples. */
void oracle_mp_sampler(const double *Xd, const float *Xf, long N, int d, long lda, gsl_rng *r, double rwsd,
                       int block_rows, int blocks, int resamp_factor, double bound_eps, int burnin, int Ns,
                       double *burn, double *samples, int *accepted_out, int *bright_out, oracle_trace_rec *tr,
                       long cap, long *ntr_out)
{
  int8_t *z = malloc((size_t)(blocks > 0 ? blocks : 1));
  memset(z, 1, (size_t)(blocks > 0 ? blocks : 1));
  mp_chain c = {Xd, Xf, N, lda, d, block_rows, blocks, resamp_factor, bound_eps, z, 0, blocks, 0, 0.0, r};
  double *cur = calloc((size_t)d, sizeof(double)), *prop = calloc((size_t)d, sizeof(double));
  double cpost = 0.0;
  long ntr = 0;
  int accepted = 0;
  if (burnin != 0) {
    for (int j = 0; j < d; j++) cur[j] = 1.0;
    cpost = mp_chain_lik(&c, cur);
    cpost += mp_prior5(cur, d);
    for (int i = 1; i < burnin; i++) {
      mp_iteration(&c, rwsd, cur, prop, &cpost, 0, tr, cap, &ntr);
      memcpy(burn + (long)i * d, cur, sizeof(double) * (size_t)d);
    }
  }
  for (int i = 0; i < Ns; i++) {
    accepted += mp_iteration(&c, rwsd, cur, prop, &cpost, 0, tr, cap, &ntr);
    memcpy(samples + (long)i * d, cur, sizeof(double) * (size_t)d);
  }
  if (accepted_out) *accepted_out = accepted;
  if (bright_out) *bright_out = c.bright;
  if (ntr_out) *ntr_out = ntr;
  free(cur);
  free(prop);
  free(z);
}
