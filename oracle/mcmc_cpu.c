/*
 * oracle/mcmc_cpu.c -- CPU restatement of the reference's double-precision MH
 * sampler, tuners and log-likelihood (src/mcmc_cpu/mcmc_cpu.c:12-445).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  "Parity unpinned": the reference
 * ships no golden vectors and cannot be built here (GSL absent).
 *
 * Everything that decides a result is kept as the reference has it: the order
 * of RNG draws (d Gaussians, then one uniform, per iteration), the prior
 * (iid N(0, 10^2) through gsl_ran_gaussian_pdf), the accept test
 * log(u) <= log-ratio, which rows of burn[] / samples[] are written, and the
 * likelihood arithmetic (column-axpy dgemv, then a serial libm loop).  The
 * eight copies of the MH loop body in the reference are one function here.
 */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static const int PRIOR_SD = 10; /* mcmc_cpu.c:10 */

/* ------------------------------------------------------------------ hooks */

static oracle_trace_rec *g_trace = NULL;
static long g_trace_cap = 0, g_trace_n = 0;
static int g_verbose = 0;

void oracle_trace_begin(oracle_trace_rec *buf, long capacity)
{
  g_trace = buf;
  g_trace_cap = capacity;
  g_trace_n = 0;
}

long oracle_trace_end(void)
{
  long n = g_trace_n;
  g_trace = NULL;
  g_trace_cap = g_trace_n = 0;
  return n;
}

void oracle_set_verbose(int on) { g_verbose = on; }

#define SAY(...)                                  \
  do {                                            \
    if (g_verbose) fprintf(stdout, __VA_ARGS__);  \
  } while (0)

static void alloc_pair(mcmc_int_v *v, int d)
{
  /* malloc_mcmc_vectors, src/util/alloc_util.c:126-138 */
  v->proposed = (double *)malloc(sizeof(double) * d);
  v->current = (double *)malloc(sizeof(double) * d);
  v->proposedf = NULL;
  v->currentf = NULL;
  if (!v->proposed || !v->current) fprintf(stderr, "ERROR: MH work vectors did not allocate.\n");
}

static void free_pair(mcmc_int_v v)
{
  free(v.proposed);
  free(v.current);
}

/* ------------------------------------------- likelihood, prior, acceptance */

/* mcmc_cpu.c:419-445 */
double log_likelihood(double *sample, data_str data, mcmc_str mcin, out_str *res)
{
  double log_lik = 0;
  const long m = mcin.Nd, lda = mcin.Nd;
  const int n = mcin.ddata;
  clock_t t0 = clock();

  oracle_dgemv_colmajor_n(m, n, 1.0, data.data, lda, sample, 0.0, data.mvout); /* :436 */

  for (long i = 0; i < m; i++) log_lik -= log(1 + exp(data.mvout[i])); /* :439-441 */

  if (res && mcin.Ns > 0) {
    /* :443 -- integer division, as in the reference: almost always adds 0 */
    clock_t dt = clock() - t0;
    res->cpuTime += (dt * 1000 / CLOCKS_PER_SEC) / mcin.Ns;
  }
  return log_lik;
}

/* mcmc_cpu.c:407-417 */
double log_prior(double *sample, mcmc_str mcin)
{
  double log_prob = 0;
  for (int k = 0; k < mcin.ddata; k++) log_prob += log(gsl_ran_gaussian_pdf(sample[k], PRIOR_SD));
  return log_prob;
}

/* mcmc_cpu.c:397-405 */
double acceptance_ratio(mcmc_int_v mclocv, mcmc_int *mcloc, mcmc_str mcin, double lhood)
{
  mcloc->pposterior = log_prior(mclocv.proposed, mcin) + lhood;
  return mcloc->pposterior - mcloc->cposterior;
}

/* --------------------------------------------------------- one MH iteration */

/* The body shared by mcmc_cpu.c:49-74, :97-120, :340-359 and :370-394.
 * Returns 1 when the proposal was accepted (current/cposterior updated). */
static int mh_step(data_str data, gsl_rng *r, mcmc_str mcin, double sd, mcmc_int_v v, mcmc_int *loc,
                   out_str *res)
{
  const int d = mcin.ddata;
  for (int k = 0; k < d; k++) v.proposed[k] = v.current[k] + gsl_ran_gaussian_ziggurat(r, sd);

  double plhood = log_likelihood(v.proposed, data, mcin, res);
  loc->acceptance = acceptance_ratio(v, loc, mcin, plhood);
  loc->u = gsl_rng_uniform(r);

  const double logu = log(loc->u);
  const int accept = (logu <= loc->acceptance);

  if (g_trace) {
    if (g_trace_n < g_trace_cap) {
      oracle_trace_rec *t = &g_trace[g_trace_n];
      t->lik = plhood;
      t->ratio = loc->acceptance;
      t->logu = logu;
      t->accept = accept;
    }
    g_trace_n++;
  }

  if (accept) {
    for (int k = 0; k < d; k++) v.current[k] = v.proposed[k];
    loc->cposterior = loc->pposterior;
  }
  return accept;
}

static void start_chain(data_str data, mcmc_str mcin, mcmc_int_v v, mcmc_int *loc, const double *theta0,
                        out_str *res)
{
  for (int k = 0; k < mcin.ddata; k++) v.current[k] = theta0[k];
  double clhood = log_likelihood(v.current, data, mcin, res);
  loc->cposterior = log_prior(v.current, mcin) + clhood;
}

/* ---------------------------------------------------------------- sampler */

/* mcmc_cpu.c:12-38 */
void cpu_sampler(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata, out_str *res)
{
  mcmc_int_v mclocv;
  mcmc_int mcloc;
  alloc_pair(&mclocv, mcin.ddata);

  clock_t t0 = clock();
  burn_in_metropolis_cpu(data, r, mcin, mct, mcdata, mclocv, &mcloc);
  clock_t burn = clock() - t0;

  int accepted = 0;
  t0 = clock();
  metropolis_cpu(data, r, mcin, mct, mcdata, mclocv, &mcloc, &accepted, res);
  clock_t run = clock() - t0;

  res->burnTime = burn * 1000 / CLOCKS_PER_SEC;
  res->mcmcTime = run * 1000 / CLOCKS_PER_SEC;
  res->acceptance = (double)accepted / mcin.Ns;
  free_pair(mclocv);
}

/* mcmc_cpu.c:40-76 */
void metropolis_cpu(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata,
                    mcmc_int_v mclocv, mcmc_int *mcloc, int *accepted_samples, out_str *res)
{
  const int d = mcin.ddata;
  res->cpuTime = 0;
  SAY("Starting metropolis algorithm. Selected rwsd = %f\n", mct.rwsd);
  for (int i = 0; i < mcin.Ns; i++) {
    if (mh_step(data, r, mcin, mct.rwsd, mclocv, mcloc, res)) *accepted_samples += 1;
    /* after the step `current` is the accepted proposal or the old state */
    for (int k = 0; k < d; k++) mcdata.samples[(long)i * d + k] = mclocv.current[k];
  }
  SAY("Metropolis algorithm finished. Accepted Samples = %d\n\n", *accepted_samples);
}

/* mcmc_cpu.c:78-122 */
void burn_in_metropolis_cpu(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata,
                            mcmc_int_v mclocv, mcmc_int *mcloc)
{
  const int d = mcin.ddata;
  out_str scratch;
  scratch.cpuTime = 0;
  SAY("Starting burn in process. Selected rwsd = %f\n", mct.rwsd);

  start_chain(data, mcin, mclocv, mcloc, mcdata.burn, &scratch); /* :88-94, theta0 = burn row 0 */

  for (int i = 1; i < mcin.burnin; i++) {
    mh_step(data, r, mcin, mct.rwsd, mclocv, mcloc, &scratch);
    for (int k = 0; k < d; k++) mcdata.burn[(long)i * d + k] = mclocv.current[k];
  }
  SAY("Burn in process finished.\n\n");
}

/* ------------------------------------------------------------- short runs */

/* mcmc_cpu.c:326-360 */
void short_run_burn_in(data_str data, gsl_rng *r, mcmc_int_v mclocv, mcmc_str mcin, double sd, mcmc_int *mcloc,
                       double *initCond)
{
  out_str scratch;
  scratch.cpuTime = 0;
  start_chain(data, mcin, mclocv, mcloc, initCond, &scratch);
  for (int i = 1; i < mcin.burnin; i++) mh_step(data, r, mcin, sd, mclocv, mcloc, &scratch);
}

/* mcmc_cpu.c:362-395 */
void short_run_metropolis(data_str data, gsl_rng *r, mcmc_int_v mclocv, mcmc_str mcin, int chain_length,
                          double sd, mcmc_int *mcloc, double *samples, int *accepted_samples)
{
  const int d = mcin.ddata;
  out_str scratch;
  scratch.cpuTime = 0;
  for (int i = 0; i < chain_length; i++) {
    if (mh_step(data, r, mcin, sd, mclocv, mcloc, &scratch)) *accepted_samples += 1;
    for (int k = 0; k < d; k++) samples[(long)i * d + k] = mclocv.current[k];
  }
}

/* ----------------------------------------------------------------- tuners */

/* One tuning round: fresh short burn-in from initCond, then a chain of
 * `chain_length`; returns the acceptance rate. */
static double tuning_round(data_str data, gsl_rng *r, mcmc_int_v v, mcmc_str mcin, int chain_length, double sd,
                           mcmc_int *loc, double *samples, double *initCond)
{
  int accepted = 0;
  short_run_burn_in(data, r, v, mcin, sd, loc, initCond);
  short_run_metropolis(data, r, v, mcin, chain_length, sd, loc, samples, &accepted);
  return accepted / (double)chain_length;
}

/* The inner search of mcmc_cpu.c:237-264 / :160-187: multiply sd by
 * acc/target until |acc - target| < 0.01 or `runs` rounds are spent.
 * *sd is left as the reference leaves it (the NEXT trial value unless the
 * loop broke); the best sd seen is returned through *best_sd. */
static void search_sd(data_str data, gsl_rng *r, mcmc_int_v v, mcmc_str mcin, int chain_length, int runs,
                      double target, mcmc_int *loc, double *samples, double *initCond, double *sd,
                      double *best_sd, const char *indent)
{
  const double error_tolerance = 0.01;
  double min_error = 9999999;
  for (int run = 0; run < runs; run++) {
    SAY("%sStarting Run %2d. Current rwsd = %5.3f\n", indent, run, *sd);
    double acc = tuning_round(data, r, v, mcin, chain_length, *sd, loc, samples, initCond);
    double err = fabs(acc - target);
    if (err < min_error) {
      *best_sd = *sd;
      min_error = err;
      SAY("%s\tAccepted: rwsd = %5.3f, acceptance = %4.3f, error = %4.3f\n", indent, *best_sd, acc, min_error);
    } else {
      SAY("%s\trwsd = %5.3f, acceptance = %4.3f, error = %4.3f\n", indent, *sd, acc, err);
    }
    if (min_error < error_tolerance) break;
    *sd *= acc / target;
  }
}

/* mcmc_cpu.c:212-271 */
void tune_target_a_cpu_v2(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, double *initCond,
                          double ratio, int max_reps)
{
  mcmc_int_v v;
  mcmc_int loc;
  alloc_pair(&v, mcin.ddata);
  const int chain_length = 5000;
  double sd = mct->rwsd, best_sd = sd;
  double *samples = (double *)malloc(sizeof(double) * mcin.ddata * chain_length);
  if (!samples) fprintf(stderr, "ERROR: Samples vector did not allocated.\n");

  SAY("\nStarting tuning process. Rwsd = %5.3f\n", sd);
  search_sd(data, r, v, mcin, chain_length, max_reps, ratio, &loc, samples, initCond, &sd, &best_sd, "\t");
  mct->rwsd = best_sd;
  SAY("Tuning finished. Selected rwsd = %5.3f\n\n", mct->rwsd);

  free(samples);
  free_pair(v);
}

/* mcmc_cpu.c:125-208 */
void tune_ess_cpu(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, double *initCond, int length)
{
  static const double target_a[9] = {0.10, 0.15, 0.20, 0.25, 0.30, 0.35, 0.40, 0.45, 0.50};
  mcmc_int_v v;
  mcmc_int loc;
  alloc_pair(&v, mcin.ddata);
  const int chain_length = length, runs = 40, lagidx = 500;
  double max_ess = -9999999;
  double sd = mct->rwsd, ess_sd = sd, best_sd = sd;
  double *samples = (double *)malloc(sizeof(double) * mcin.ddata * chain_length);
  double *autocorr = (double *)malloc(sizeof(double) * lagidx);
  if (!samples || !autocorr) fprintf(stderr, "ERROR: tuning vectors did not allocate.\n");

  SAY("\nStarting tuning process. Rwsd = %5.3f\n", sd);
  for (int a = 0; a < 9; a++) {
    SAY("\tStarting tuning for target ratio = %4.3f. Current rwsd = %5.3f\n", target_a[a], sd);
    search_sd(data, r, v, mcin, chain_length, runs, target_a[a], &loc, samples, initCond, &sd, &best_sd, "\t\t");
    /* :189-201 -- ESS of the LAST chain run, credited to the current sd */
    double circ = circular_autocorrelation(autocorr, samples, mcin.ddata, chain_length, lagidx);
    double ess = chain_length / (1 + 2 * circ);
    if (ess > max_ess) {
      max_ess = ess;
      ess_sd = sd;
      SAY("\tAccepted: ess = %8.3f, rwsd = %5.3f\n", max_ess, ess_sd);
    } else {
      SAY("\tess= %8.3f, rwsd = %5.3f\n", ess, sd);
    }
  }
  mct->rwsd = ess_sd;
  SAY("Tuning finished. Selected rwsd = %5.3f\n\n", mct->rwsd);

  free(samples);
  free(autocorr);
  free_pair(v);
}

/* mcmc_cpu.c:274-324 -- the legacy +-10 % stepper */
void tune_target_a_cpu(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, double *initCond)
{
  mcmc_int_v v;
  mcmc_int loc;
  alloc_pair(&v, mcin.ddata);
  const int chain_length = 5000;
  const double target = 0.25, tol = 0.01, step = 0.1;
  double sd = mct->rwsd;
  double *samples = (double *)malloc(sizeof(double) * mcin.ddata * chain_length);
  if (!samples) fprintf(stderr, "Samples vector did not allocated.\n");

  SAY("\nStarting tuning process. Rwsd = %5.3f\n", sd);
  for (int run = 0;; run++) {
    SAY("\tStarting Run %2d. Current rwsd = %5.3f, Acceptance = ", run, sd);
    double acc = tuning_round(data, r, v, mcin, chain_length, sd, &loc, samples, initCond);
    SAY("%4.3f\n", acc);
    if (acc > target + tol)
      sd *= (1 + step);
    else if (acc < target - tol)
      sd *= (1 - step);
    else
      break;
  }
  mct->rwsd = sd;
  SAY("Tuning finished. Selected rwsd = %5.3f\n\n", mct->rwsd);

  free(samples);
  free_pair(v);
}
