/*
 * oracle/sp_path.c -- CPU restatement of the HOST arithmetic of the reference's
 * single-precision sampler (src/mcmc_gpu_sp/mcmc_gpu_sp.cu:20-177 sp_sampler,
 * metropolis_sp, burn_in_metropolis_sp; :561-580 acceptance_ratio_sp,
 * log_prior_sp).  TEST INFRASTRUCTURE ONLY (see oracle.h).  "Parity unpinned":
 * the reference ships no golden vectors and cannot be built here.
 *
 * "Teacher forcing": in float the N-term likelihood sum legitimately depends on
 * the summation order (the reference's own value depends on its block size,
 * mcmc_gpu_sp.cu:179-214), so this oracle does not evaluate the likelihood.  The
 * caller feeds it the float likelihood of every evaluation (lik[0] = chain
 * start, then one per iteration, burn-in first), and the oracle restates
 * everything AROUND it, operation for operation and type for type:
 *   - the state, the proposal and the posteriors are float (mcmc_int.cposteriorf,
 *     resources.h:118-121);
 *   - proposedf[j] = currentf[j] + (float)gsl_ran_gaussian_ziggurat(r, (double)rwsdf)   (:105,:154)
 *   - log_prior_sp: a FLOAT accumulator += DOUBLE log(gsl_ran_gaussian_pdf(x, 10))     (:570-580)
 *   - pposteriorf = prior + plhood; ratio = pposteriorf - cposteriorf, all float        (:561-568)
 *   - uf = gsl_rng_uniform(r) rounded to float; accept iff log(uf) <= ratio              (:112-114)
 *     The file is CUDA C++: with a float argument `log` resolves to the float
 *     overload, so the comparison is float against float (logf here).
 *   - burn-in writes rows 1..burnin-1 of burnf (row 0 is the start), sampling
 *     writes every row of samplesf                                                       (:117-127,:163-173)
 * The feed is checked by the caller against the DP oracle (1e-4 relative).
 */
#include "oracle.h"

#include <math.h>

static const int SP_PRIOR_SD = 10; /* mcmc_gpu_sp.cu:9 region: PRIOR_SD 10 */

static float sp_log_prior(const float *sample, int d)
{
  float log_prob = 0;
  for (int j = 0; j < d; j++) log_prob += log(gsl_ran_gaussian_pdf(sample[j], SP_PRIOR_SD));
  return log_prob;
}

/* what follows the likelihood call in one iteration of :101-128 / :150-175; returns the accept bit */
static int sp_step(gsl_rng *r, int d, float *current, const float *proposed, float *cposterior, float plhood,
                   oracle_trace_rec *rec)
{
  const float pposterior = sp_log_prior(proposed, d) + plhood;
  const float ratio = pposterior - *cposterior;
  const float uf = gsl_rng_uniform(r);
  const float logu = logf(uf);
  const int accept = logu <= ratio;
  if (accept) {
    for (int j = 0; j < d; j++) current[j] = proposed[j];
    *cposterior = pposterior;
  }
  if (rec) {
    rec->lik = plhood;
    rec->ratio = ratio;
    rec->logu = logu;
    rec->accept = accept;
  }
  return accept;
}

/* Proposal of the NEXT evaluation: the caller needs it to produce lik[] (this is
 * what makes the forcing honest: the proposals come from this oracle's own state
 * and RNG stream, so a product whose state or stream differs feeds the wrong
 * likelihoods and the traces part). */
static void sp_propose(gsl_rng *r, int d, float rwsdf, const float *current, float *proposed)
{
  for (int j = 0; j < d; j++) proposed[j] = current[j] + (float)gsl_ran_gaussian_ziggurat(r, (double)rwsdf);
}

/*
 * lik[burnin + Ns] (burnin >= 1): lik[0] = L(theta0), lik[i] the likelihood of the i-th proposal.
 * proposals_out (may be NULL): [burnin - 1 + Ns][d], the proposals this oracle made -- the caller can evaluate
 * them independently and compare with the feed.  Returns the number of trace records written (burnin - 1 + Ns).
 */
long oracle_sp_sampler(gsl_rng *r, int d, float rwsdf, int burnin, int Ns, const float *lik, float *burnf,
                       float *samplesf, float *proposals_out, int *accepted_out, float *cposterior_out,
                       oracle_trace_rec *tr, long cap)
{
  float current[4096], proposed[4096];
  if (d > 4096 || burnin < 1) return -1;
  long n = 0, ev = 0;
  int accepted = 0;
  for (int j = 0; j < d; j++) current[j] = burnf[j]; /* :142-143 */
  float cposterior = sp_log_prior(current, d) + lik[ev++]; /* :145-147 */
  for (int i = 1; i < burnin; i++) {                      /* :150-175 */
    sp_propose(r, d, rwsdf, current, proposed);
    if (proposals_out)
      for (int j = 0; j < d; j++) proposals_out[n * d + j] = proposed[j];
    sp_step(r, d, current, proposed, &cposterior, lik[ev++], n < cap ? &tr[n] : NULL);
    n++;
    for (int j = 0; j < d; j++) burnf[(long)i * d + j] = current[j];
  }
  for (int i = 0; i < Ns; i++) { /* :101-128 */
    sp_propose(r, d, rwsdf, current, proposed);
    if (proposals_out)
      for (int j = 0; j < d; j++) proposals_out[n * d + j] = proposed[j];
    accepted += sp_step(r, d, current, proposed, &cposterior, lik[ev++], n < cap ? &tr[n] : NULL);
    n++;
    for (int j = 0; j < d; j++) samplesf[(long)i * d + j] = current[j];
  }
  if (accepted_out) *accepted_out = accepted;
  if (cposterior_out) *cposterior_out = cposterior;
  return n;
}
