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
  double *inc, *u;    /* [K][d], [K]: the random numbers of the NEXT iteration, drawn while the device runs this one */
} mc_state;

static void mc_begin(mc_state *s, const double *Theta)
{
  if (s->f32) {
    for (long i = 0; i < (long)s->K * s->d; i++) s->propf[i] = (float)Theta[i];
    MH_CHECK(mhlik_loglik_batched_begin_f32(s->lik, s->propf, s->K), s->lik);
  } else {
    MH_CHECK(mhlik_loglik_batched_begin_f64(s->lik, Theta, s->K), s->lik);
  }
}

static void mc_end(mc_state *s) { MH_CHECK(mhlik_loglik_batched_end(s->lik, s->L), s->lik); }

/* What one iteration takes from chain k's generator, in the reference's order: d Gaussians, then the uniform of
 * the accept test (mcmc_gpu.cu:105,111).  Neither depends on the chain's state and nothing else draws from r[k]
 * in between, so they can be drawn ahead of the evaluation they belong to without changing the stream. */
static void mc_draw(mc_state *s, gsl_rng **r, double sd)
{
  const int K = s->K, d = s->d;
  for (int k = 0; k < K; k++) {
    for (int j = 0; j < d; j++) s->inc[(long)k * d + j] = gsl_ran_gaussian_ziggurat(r[k], sd);
    s->u[k] = gsl_rng_uniform(r[k]);
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
  s.inc = (double *)calloc((size_t)K * d, sizeof(double));
  s.u = (double *)calloc((size_t)K, sizeof(double));
  double *u_now = (double *)calloc((size_t)K, sizeof(double));
  double *pprior = (double *)calloc((size_t)K, sizeof(double));

  const long bl = mcin.burnin > 0 ? mcin.burnin : 1;
  const long nburn = mcin.burnin > 0 ? mcin.burnin - 1 : 0; /* burn-in iterations after the start */
  const long total = nburn + mcin.Ns;
  for (int k = 0; k < K; k++) {
    accepted[k] = 0;
    if (mcin.burnin > 0) memcpy(s.cur + (long)k * d, burn + (long)k * bl * d, sizeof(double) * d);
  }
  if (mcin.burnin > 0) { /* the chain start, burn_in_metropolis_gpu (mcmc_gpu.cu:139-146) */
    mc_begin(&s, s.cur);
    if (total > 0) mc_draw(&s, r, mct.rwsd);
    mc_end(&s);
    for (int k = 0; k < K; k++) s.cpost[k] = mh_log_prior_d(s.cur + (long)k * d, d, PRIOR_SD) + s.L[k];
  } else if (total > 0) {
    mc_draw(&s, r, mct.rwsd);
  }
  for (long it = 0; it < total; it++) {
    /* proposal from the numbers drawn ahead; this iteration's uniforms move aside before the next draw */
    for (long i = 0; i < (long)K * d; i++) s.prop[i] = s.cur[i] + s.inc[i];
    memcpy(u_now, s.u, sizeof(double) * K);
    mc_begin(&s, s.prop);
    /* host work that does not need the likelihood runs under the device pass: the next iteration's random
     * numbers (never past the last iteration) and the proposals' log prior */
    if (it + 1 < total) mc_draw(&s, r, mct.rwsd);
    for (int k = 0; k < K; k++) pprior[k] = mh_log_prior_d(s.prop + (long)k * d, d, PRIOR_SD);
    mc_end(&s);
    const int sampling = it >= nburn;
    for (int k = 0; k < K; k++) {
      const double ppost = pprior[k] + s.L[k];
      const double ratio = ppost - s.cpost[k];
      if (log(u_now[k]) <= ratio) {
        memcpy(s.cur + (long)k * d, s.prop + (long)k * d, sizeof(double) * d);
        s.cpost[k] = ppost;
        if (sampling) accepted[k] += 1;
      }
      if (sampling)
        memcpy(samples + ((long)k * mcin.Ns + (it - nburn)) * d, s.cur + (long)k * d, sizeof(double) * d);
      else
        memcpy(burn + ((long)k * bl + it + 1) * d, s.cur + (long)k * d, sizeof(double) * d);
    }
  }
  free(s.prop);
  free(s.cur);
  free(s.propf);
  free(s.cpost);
  free(s.L);
  free(s.inc);
  free(s.u);
  free(u_now);
  free(pprior);
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

/* the phases of one evaluation, from each chain's own generator (mcmc_mp.cu:504) */
static void mpk_draw_phase(mpk_env *e, gsl_rng **r, int32_t *choose)
{
  for (int k = 0; k < e->K; k++) choose[k] = (int32_t)gsl_rng_uniform_int(r[k], (unsigned long)e->mp->ResampFactor);
  if (e->shared_phase)
    for (int k = 1; k < e->K; k++) choose[k] = choose[0];
}

/* queue one evaluation with the phases in e->choose; mpk_end collects it */
static void mpk_begin(mpk_env *e, const double *Theta)
{
  mhlik_mp_batch st;
  st.block_rows = e->mp->dataBlockSz;
  st.blocks = e->mp->dataBlocks;
  st.resamp_factor = e->mp->ResampFactor;
  st.K = e->K;
  st.resamp_choose = e->choose;
  st.u_seed = e->seeds;
  st.u_call = e->calls++;
  st.bound_eps = e->bound;
  MH_CHECK(mhlik_mp_loglik_batched_begin(e->lik, &st, Theta), e->lik);
}

static void mpk_end(mpk_env *e) { MH_CHECK(mhlik_mp_loglik_batched_end(e->lik, e->L, e->nb, e->nd), e->lik); }

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
  /* what the NEXT iteration takes from the generators, drawn while the device runs this one: per chain, in the
   * reference's order, d Gaussians (:389), the phase (:504), the uniform of the accept test (:400).  None of them
   * depends on the chain's state and nothing else draws from r[k] in between. */
  double *inc = (double *)calloc((size_t)K * d, sizeof(double));
  int32_t *choose_next = (int32_t *)calloc((size_t)K, sizeof(int32_t));
  double *u_next = (double *)calloc((size_t)K, sizeof(double));
  double *u_now = (double *)calloc((size_t)K, sizeof(double));
  double *pprior = (double *)calloc((size_t)K, sizeof(double));
  for (int k = 0; k < K; k++) {
    e.seeds[k] = MP_U_SEED;
    accepted[k] = 0;
  }

  const long bl = mcin.burnin > 0 ? mcin.burnin : 1;
  const long total = (mcin.burnin > 0 ? mcin.burnin - 1 : 0) + mcin.Ns;
#define MPK_DRAW_AHEAD()                                                                              \
  do {                                                                                                \
    for (int k = 0; k < K; k++)                                                                       \
      for (int j = 0; j < d; j++) inc[(long)k * d + j] = gsl_ran_gaussian_ziggurat(r[k], mct.rwsd);   \
    mpk_draw_phase(&e, r, choose_next);                                                               \
    for (int k = 0; k < K; k++) u_next[k] = gsl_rng_uniform(r[k]);                                    \
  } while (0)
  if (mcin.burnin != 0) { /* start of burn_in_metropolis_mp, :430-440 */
    for (long i = 0; i < (long)K * d; i++) cur[i] = theta0 ? theta0[i % d] : 1; /* :433 starts every chain at 1 */
    mpk_draw_phase(&e, r, e.choose);
    mpk_begin(&e, cur);
    if (total > 0) MPK_DRAW_AHEAD();
    mpk_end(&e);
    for (int k = 0; k < K; k++) cpost[k] = mh_log_prior_d(cur + (long)k * d, d, MP_PRIOR_SD) + e.L[k];
  } else if (total > 0) {
    MPK_DRAW_AHEAD();
  }
  for (long it = 0; it < total; it++) {
    const int sampling = mcin.burnin > 0 ? it >= mcin.burnin - 1 : 1;
    for (long i = 0; i < (long)K * d; i++) prop[i] = cur[i] + inc[i];
    memcpy(e.choose, choose_next, sizeof(int32_t) * K);
    memcpy(u_now, u_next, sizeof(double) * K);
    mpk_begin(&e, prop);
    /* under the device pass: the next iteration's random numbers (never past the last iteration), the log prior */
    if (it + 1 < total) MPK_DRAW_AHEAD();
    for (int k = 0; k < K; k++) pprior[k] = mh_log_prior_d(prop + (long)k * d, d, MP_PRIOR_SD);
    mpk_end(&e);
    for (int k = 0; k < K; k++) {
      const double ppost = pprior[k] + e.L[k];
      const double a = exp(ppost - cpost[k]); /* acceptance_ratio_mp, :635 */
      if (u_now[k] <= a) { /* :402 */
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
#undef MPK_DRAW_AHEAD
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
  free(inc);
  free(choose_next);
  free(u_next);
  free(u_now);
  free(pprior);
}
