/*
 * multichain.c -- K independent MH chains advanced in lock step, one batched
 * likelihood pass (X.Theta on the matrix cores) per iteration.
 *
 * No counterpart in the reference, which runs one chain with one theta per
 * likelihood call (src/mcmc_gpu/mcmc_gpu.cu:101-107); this is the host loop
 * that feeds BASELINE.json's batched mode.  Each chain k is exactly the
 * reference's DP chain (same burn-in / sampling loops, prior N(0,10^2),
 * accept test log(u) <= ratio, draw order d Gaussians then one uniform) driven
 * by its own generator r[k], so chain k reproduces what gpu_sampler would
 * give with that generator, up to the rounding of the batched row dots.
 */
#include "mh_core.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const int PRIOR_SD = 10;

typedef struct {
  mhlik_ctx *lik;
  int K, d, f32;
  double *prop, *cur; /* [K][d] */
  float *propf;       /* [K][d] when f32 */
  double *cpost, *L;  /* [K] */
} mc_state;

static void mc_eval(mc_state *s, const double *Theta)
{
  if (s->f32) {
    for (long i = 0; i < (long)s->K * s->d; i++) s->propf[i] = (float)Theta[i];
    MH_CHECK(mhlik_loglik_batched_f32(s->lik, s->propf, s->K, s->L), s->lik);
  } else {
    MH_CHECK(mhlik_loglik_batched_f64(s->lik, Theta, s->K, s->L), s->lik);
  }
}

/* one lock-step iteration; returns nothing, bumps accepted[k] when asked */
static void mc_step(mc_state *s, gsl_rng **r, double sd, int *accepted)
{
  const int K = s->K, d = s->d;
  for (int k = 0; k < K; k++)
    for (int j = 0; j < d; j++) s->prop[(long)k * d + j] = s->cur[(long)k * d + j] + gsl_ran_gaussian_ziggurat(r[k], sd);
  mc_eval(s, s->prop);
  for (int k = 0; k < K; k++) {
    const double ppost = mh_log_prior_d(s->prop + (long)k * d, d, PRIOR_SD) + s->L[k];
    const double ratio = ppost - s->cpost[k];
    const double u = gsl_rng_uniform(r[k]);
    if (log(u) <= ratio) {
      memcpy(s->cur + (long)k * d, s->prop + (long)k * d, sizeof(double) * d);
      s->cpost[k] = ppost;
      if (accepted) accepted[k] += 1;
    }
  }
}

/* samples: [K][Ns][d], burn: [K][burnin][d] (row 0 of each chain = its start),
 * accepted: [K].  use_f32 != 0 evaluates on the float matrix. */
void mcmc_amd_multichain_sampler(mhlik_ctx *lik, int K, gsl_rng **r, mcmc_str mcin, mcmc_tune_str mct,
                                 double *samples, double *burn, int *accepted, int use_f32)
{
  const int d = mcin.ddata;
  if (K < 1 || K > MHLIK_MAX_BATCH) {
    fprintf(stderr, "ERROR: multichain sampler needs 1 <= K <= %d.\n", MHLIK_MAX_BATCH);
    exit(MHLIK_EINVAL);
  }
  mc_state s;
  s.lik = lik;
  s.K = K;
  s.d = d;
  s.f32 = use_f32;
  s.prop = (double *)calloc((size_t)K * d, sizeof(double));
  s.cur = (double *)calloc((size_t)K * d, sizeof(double));
  s.propf = (float *)calloc((size_t)K * d, sizeof(float));
  s.cpost = (double *)calloc((size_t)K, sizeof(double));
  s.L = (double *)calloc((size_t)K, sizeof(double));

  const long bl = mcin.burnin > 0 ? mcin.burnin : 1;
  for (int k = 0; k < K; k++) {
    accepted[k] = 0;
    if (mcin.burnin > 0) memcpy(s.cur + (long)k * d, burn + (long)k * bl * d, sizeof(double) * d);
  }
  if (mcin.burnin > 0) {
    mc_eval(&s, s.cur);
    for (int k = 0; k < K; k++) s.cpost[k] = mh_log_prior_d(s.cur + (long)k * d, d, PRIOR_SD) + s.L[k];
    for (int i = 1; i < mcin.burnin; i++) {
      mc_step(&s, r, mct.rwsd, NULL);
      for (int k = 0; k < K; k++) memcpy(burn + ((long)k * bl + i) * d, s.cur + (long)k * d, sizeof(double) * d);
    }
  }
  for (int i = 0; i < mcin.Ns; i++) {
    mc_step(&s, r, mct.rwsd, accepted);
    for (int k = 0; k < K; k++)
      memcpy(samples + ((long)k * mcin.Ns + i) * d, s.cur + (long)k * d, sizeof(double) * d);
  }
  free(s.prop);
  free(s.cur);
  free(s.propf);
  free(s.cpost);
  free(s.L);
}
