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

/* ------------------------------------------------------------------------- *
 * K mixed-precision chains in lock step (BASELINE.json config 5).
 *
 * Chain k is mp_sampler (src/mcmc_mp/mcmc_mp.cu:283-473; here mcmc_mp.c) run
 * with generator r[k]: start at theta = 1 (:433), prior N(0,5^2) (:16), per
 * iteration d Gaussians, then gsl_rng_uniform_int(r[k], ResampFactor) inside
 * the likelihood (:504), then one uniform, accept iff u <= exp(log ratio)
 * (:402).  Every chain uses the per-block uniform seed 1234 (:329) and the
 * shared call counter, like K separate runs of the reference would.  One
 * mhlik_mp_loglik_batched call per iteration evaluates all K proposals.
 *
 * shared_phase != 0: every chain still draws its phase from its own generator
 * (the streams stay those of mp_sampler) but all chains re-examine the phase
 * chain 0 drew.  Which dark blocks a chain looks at again in a given iteration
 * may be any rule that does not depend on the chain's state, so each chain
 * stays a valid sampler; what changes is cost: with one phase only 1/F of the
 * double-precision matrix is read per iteration instead of nearly all of it.
 * ------------------------------------------------------------------------- */
static const int MP_PRIOR_SD = 5;
static const uint64_t MP_U_SEED = 1234ULL;

typedef struct {
  mhlik_ctx *lik;
  int K, d;
  mp_str *mp;
  double bound;
  uint64_t calls;
  int shared_phase;
  int32_t *choose, *nb, *nd;
  uint64_t *seeds;
  double *L;
} mpk_env;

static void mpk_eval(mpk_env *e, gsl_rng **r, const double *Theta)
{
  for (int k = 0; k < e->K; k++) e->choose[k] = (int32_t)gsl_rng_uniform_int(r[k], (unsigned long)e->mp->ResampFactor);
  if (e->shared_phase)
    for (int k = 1; k < e->K; k++) e->choose[k] = e->choose[0];
  mhlik_mp_batch st;
  st.block_rows = e->mp->dataBlockSz;
  st.blocks = e->mp->dataBlocks;
  st.resamp_factor = e->mp->ResampFactor;
  st.K = e->K;
  st.resamp_choose = e->choose;
  st.u_seed = e->seeds;
  st.u_call = e->calls++;
  st.bound_eps = e->bound;
  MH_CHECK(mhlik_mp_loglik_batched(e->lik, &st, Theta, e->L, e->nb, e->nd), e->lik);
}

/* samples [K][Ns][d]; burn [K][burnin][d] (row 0 of a chain is not written,
 * like mp_sampler); accepted [K]; bright_blocks [K] (may be NULL): bright
 * blocks of each chain after the last iteration.  The context must hold both
 * matrices with segments = 1.  theta0 (d values, may be NULL): common start of
 * the chains instead of the reference's theta = 1, which saturates the
 * sigmoids (and makes log(sigma_S - eps) NaN) once d is large. */
void mcmc_amd_mp_multichain_sampler(mhlik_ctx *lik, int K, gsl_rng **r, mp_str *mpVar, mcmc_str mcin,
                                    mcmc_tune_str mct, double *samples, double *burn, int *accepted,
                                    double lhoodBound, int shared_phase, int *bright_blocks, const double *theta0)
{
  const int d = mcin.ddata;
  if (K < 1 || K > MHLIK_MAX_BATCH) {
    fprintf(stderr, "ERROR: multichain sampler needs 1 <= K <= %d.\n", MHLIK_MAX_BATCH);
    exit(MHLIK_EINVAL);
  }
  MH_CHECK(mhlik_mp_reset_batched(lik, mpVar->dataBlockSz, mpVar->dataBlocks, K), lik);
  mpVar->brightBlocks = mpVar->dataBlocks; /* mcmc_mp.cu:333-334 */
  mpVar->darkBlocks = 0;

  mpk_env e;
  e.lik = lik;
  e.K = K;
  e.d = d;
  e.mp = mpVar;
  e.bound = lhoodBound;
  e.calls = 0;
  e.shared_phase = shared_phase;
  e.choose = (int32_t *)calloc((size_t)K, sizeof(int32_t));
  e.nb = (int32_t *)calloc((size_t)K, sizeof(int32_t));
  e.nd = (int32_t *)calloc((size_t)K, sizeof(int32_t));
  e.seeds = (uint64_t *)calloc((size_t)K, sizeof(uint64_t));
  e.L = (double *)calloc((size_t)K, sizeof(double));
  double *prop = (double *)calloc((size_t)K * d, sizeof(double));
  double *cur = (double *)calloc((size_t)K * d, sizeof(double));
  double *cpost = (double *)calloc((size_t)K, sizeof(double));
  for (int k = 0; k < K; k++) {
    e.seeds[k] = MP_U_SEED;
    accepted[k] = 0;
  }

  const long bl = mcin.burnin > 0 ? mcin.burnin : 1;
  const long total = (mcin.burnin > 0 ? mcin.burnin - 1 : 0) + mcin.Ns;
  if (mcin.burnin != 0) { /* start of burn_in_metropolis_mp, :430-440 */
    for (long i = 0; i < (long)K * d; i++) cur[i] = theta0 ? theta0[i % d] : 1; /* :433 starts every chain at 1 */
    mpk_eval(&e, r, cur);
    for (int k = 0; k < K; k++) cpost[k] = mh_log_prior_d(cur + (long)k * d, d, MP_PRIOR_SD) + e.L[k];
  }
  for (long it = 0; it < total; it++) {
    const int sampling = mcin.burnin > 0 ? it >= mcin.burnin - 1 : 1;
    for (int k = 0; k < K; k++)
      for (int j = 0; j < d; j++) prop[(long)k * d + j] = cur[(long)k * d + j] + gsl_ran_gaussian_ziggurat(r[k], mct.rwsd);
    mpk_eval(&e, r, prop);
    for (int k = 0; k < K; k++) {
      const double ppost = mh_log_prior_d(prop + (long)k * d, d, MP_PRIOR_SD) + e.L[k];
      const double a = exp(ppost - cpost[k]); /* acceptance_ratio_mp, :635 */
      const double u = gsl_rng_uniform(r[k]);
      if (u <= a) { /* :402 */
        memcpy(cur + (long)k * d, prop + (long)k * d, sizeof(double) * d);
        cpost[k] = ppost;
        if (sampling) accepted[k] += 1;
      }
      if (sampling) {
        const long i = mcin.burnin > 0 ? it - (mcin.burnin - 1) : it;
        memcpy(samples + ((long)k * mcin.Ns + i) * d, cur + (long)k * d, sizeof(double) * d);
      } else {
        memcpy(burn + ((long)k * bl + it + 1) * d, cur + (long)k * d, sizeof(double) * d);
      }
    }
  }
  long nb_sum = 0;
  for (int k = 0; k < K; k++) {
    if (bright_blocks) bright_blocks[k] = e.nb[k];
    nb_sum += e.nb[k];
  }
  mpVar->brightBlocks = (int)(nb_sum / K); /* mean over the chains */
  mpVar->darkBlocks = mpVar->dataBlocks - mpVar->brightBlocks;
  mpVar->ratio = ((float)mpVar->brightBlocks) / mpVar->dataBlocks;

  free(e.choose);
  free(e.nb);
  free(e.nd);
  free(e.seeds);
  free(e.L);
  free(prop);
  free(cur);
  free(cpost);
}
