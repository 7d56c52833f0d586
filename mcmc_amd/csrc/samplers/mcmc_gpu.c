/*
 * mcmc_gpu.c -- double-precision MH sampler and tuners on the MI355X.
 *
 * Entry points, argument lists and observable behaviour follow the reference
 * (src/mcmc_gpu/mcmc_gpu.cu:20-84 gpu_sampler, :86-175 metropolis / burn-in,
 * :260-561 tuners, :563-637 short runs, :639-658 ratio / prior); the
 * likelihood call inside the loop goes to mhlik_loglik_f64 instead of
 * cublasDgemv + reduceSum_d + cudaMemcpy + CPU tail sum.
 *
 * Differences on purpose ("bugs not to copy"):
 *  - no per-call event creation / double device sync (mcmc_gpu.cu:225-244);
 *    kernel time comes from the context's two persistent events;
 *  - short_run_burn_in_gpu starts from initCond, as the CPU twin does
 *    (mcmc_cpu.c:333-334); the reference's GPU twin ignores it and starts
 *    from 0 (mcmc_gpu.cu:572-573) -- every caller passes zeros;
 *  - burnin == 0: the reference skips burn-in and samples from an
 *    uninitialised `current` (mcmc_gpu.cu:61-62); here `current` is zeros and
 *    cposterior is 0, the values the reference assigns at :35.
 */
#include "mh_core.h"

#include <math.h>
#include <stdlib.h>
#include <time.h>

static const int PRIOR_SD = 10; /* mcmc_gpu.cu:6 */

/* mcmc_gpu.cu:648-658 */
double log_prior_gpu(double *sample, mcmc_str mcin) { return mh_log_prior_d(sample, mcin.ddata, PRIOR_SD); }

/* mcmc_gpu.cu:639-646 */
double acceptance_ratio_gpu(mcmc_int_v mclocv, mcmc_int *mcloc, mcmc_str mcin, double plhood)
{
  mcloc->pposterior = log_prior_gpu(mclocv.proposed, mcin) + plhood;
  return mcloc->pposterior - mcloc->cposterior;
}

/* Role of gpu_likelihood_d (mcmc_gpu.cu:214-257): L(theta) on the device plus
 * the running averages the reference keeps in out_str.  cu* (cuBLAS) fields
 * stay 0: there is no separate gemv any more. */
double gpu_likelihood_d(mhlik_ctx *lik, mcmc_str mcin, double *samples, out_str *res)
{
  double L = 0;
  MH_CHECK(mhlik_loglik_f64(lik, samples, &L), lik);
  if (res && mcin.Ns > 0) {
    float ms = 0;
    int64_t bytes = 0;
    mhlik_last_kernel_ms(lik, &ms, &bytes);
    if (ms > 0) {
      res->kernelTime += ms / mcin.Ns;
      res->kernelBandwidth += ((double)bytes / ms / 1e6) / mcin.Ns;
      res->gpuTime += ms / mcin.Ns;
      res->gpuBandwidth += ((double)bytes / ms / 1e6) / mcin.Ns;
    }
  }
  return L;
}

static double lik_cb(void *env, mcmc_str mcin, mcmc_int_v v, out_str *res)
{
  return gpu_likelihood_d((mhlik_ctx *)env, mcin, v.proposed, res);
}

static void zero_gpu_times(out_str *res)
{
  res->cuTime = res->cuBandwidth = 0;
  res->kernelTime = res->kernelBandwidth = 0;
  res->gpuTime = res->gpuBandwidth = 0;
}

static void start_chain(mhlik_ctx *lik, mcmc_str mcin, mcmc_int_v v, mcmc_int *loc, const double *theta0)
{
  for (int k = 0; k < mcin.ddata; k++) v.current[k] = theta0[k];
  double clhood = gpu_likelihood_d(lik, mcin, v.current, NULL);
  loc->cposterior = log_prior_gpu(v.current, mcin) + clhood;
}

/* n MH iterations from the state in v.current / loc->cposterior; rows: where
 * the state after each iteration goes (n rows of d, or NULL). */
static void run_steps(mhlik_ctx *lik, gsl_rng *r, mcmc_str mcin, double sd, mcmc_int_v v, mcmc_int *loc, long n,
                      double *rows, int *accepted, out_str *res)
{
  const int d = mcin.ddata;
  if (mh_device_chain_enabled()) {
    mh_chain_steps_d(lik, r, mcin, sd, v, loc, n, rows, accepted, res, PRIOR_SD);
    return;
  }
  for (long i = 0; i < n; i++) {
    if (mh_step_d(lik_cb, lik, r, mcin, sd, v, loc, res, PRIOR_SD, 0) && accepted) *accepted += 1;
    if (rows)
      for (int k = 0; k < d; k++) rows[i * d + k] = v.current[k];
  }
}

/* mcmc_gpu.cu:86-130 */
void metropolis_gpu(mhlik_ctx *lik, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata,
                    mcmc_int_v mclocv, mcmc_int *mcloc, int *accepted_samples, out_str *res)
{
  zero_gpu_times(res);
  MH_SAY("Starting metropolis algorithm. Selected rwsd = %f\n", mct.rwsd);
  run_steps(lik, r, mcin, mct.rwsd, mclocv, mcloc, mcin.Ns, mcdata.samples, accepted_samples, res);
  MH_SAY("Metropolis algorithm finished. Accepted Samples = %d\n\n", *accepted_samples);
}

/* mcmc_gpu.cu:132-175 */
void burn_in_metropolis_gpu(mhlik_ctx *lik, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata,
                            mcmc_int_v mclocv, mcmc_int *mcloc)
{
  const int d = mcin.ddata;
  MH_SAY("Starting burn in process. Selected rwsd = %f\n", mct.rwsd);
  start_chain(lik, mcin, mclocv, mcloc, mcdata.burn);
  run_steps(lik, r, mcin, mct.rwsd, mclocv, mcloc, mcin.burnin - 1, mcdata.burn + d, NULL, NULL);
  MH_SAY("Burn in process finished.\n\n");
}

/* mcmc_gpu.cu:20-84 */
void gpu_sampler(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata, gpu_v_str gpu,
                 out_str *res)
{
  (void)gpu; /* launch knobs of the CUDA path: accepted, not used */
  int owned = 0;
  mhlik_ctx *lik = mh_acquire(data, mcin, MHLIK_F64, &owned);

  mcmc_int_v mclocv;
  mcmc_int mcloc;
  mcloc.cposterior = mcloc.pposterior = mcloc.acceptance = mcloc.u = 0;
  mh_alloc_vectors(&mclocv, mcin.ddata, 1, 0);

  clock_t t0 = clock();
  if (mcin.burnin != 0) burn_in_metropolis_gpu(lik, r, mcin, mct, mcdata, mclocv, &mcloc);
  clock_t burn = clock() - t0;

  int accepted = 0;
  t0 = clock();
  metropolis_gpu(lik, r, mcin, mct, mcdata, mclocv, &mcloc, &accepted, res);
  clock_t run = clock() - t0;

  res->burnTime = burn * 1000 / CLOCKS_PER_SEC;
  res->mcmcTime = run * 1000 / CLOCKS_PER_SEC;
  res->acceptance = (double)accepted / mcin.Ns;

  mh_free_vectors(mclocv);
  mh_release(lik, owned);
}

/* ------------------------------------------------------------- short runs */

/* mcmc_gpu.cu:563-600 */
static void short_run_burn_in_gpu(mhlik_ctx *lik, gsl_rng *r, mcmc_int_v v, mcmc_str mcin, double sd, mcmc_int *loc,
                                  const double *initCond)
{
  start_chain(lik, mcin, v, loc, initCond);
  run_steps(lik, r, mcin, sd, v, loc, mcin.burnin - 1, NULL, NULL, NULL);
}

/* mcmc_gpu.cu:602-637 */
static void short_run_metropolis_gpu(mhlik_ctx *lik, gsl_rng *r, mcmc_int_v v, mcmc_str mcin, int chain_length,
                                     double sd, mcmc_int *loc, double *samples, int *accepted)
{
  run_steps(lik, r, mcin, sd, v, loc, chain_length, samples, accepted, NULL);
}

static double tuning_round(mhlik_ctx *lik, gsl_rng *r, mcmc_int_v v, mcmc_str mcin, int chain_length, double sd,
                           mcmc_int *loc, double *samples, const double *initCond)
{
  int accepted = 0;
  short_run_burn_in_gpu(lik, r, v, mcin, sd, loc, initCond);
  short_run_metropolis_gpu(lik, r, v, mcin, chain_length, sd, loc, samples, &accepted);
  return accepted / (double)chain_length;
}

/* The search loop of mcmc_gpu.cu:434-462 / :323-351: scale sd by acc/target
 * until |acc - target| < 0.01 or `runs` rounds are spent.  *sd ends as the
 * reference leaves it (the next trial value unless the loop broke). */
static void search_sd(mhlik_ctx *lik, gsl_rng *r, mcmc_int_v v, mcmc_str mcin, int chain_length, int runs,
                      double target, mcmc_int *loc, double *samples, const double *initCond, double *sd,
                      double *best_sd, const char *indent)
{
  const double error_tolerance = 0.01;
  double min_error = 9999999;
  for (int run = 0; run < runs; run++) {
    MH_SAY("%sStarting Run %2d. Current rwsd = %5.3f\n", indent, run, *sd);
    double acc = tuning_round(lik, r, v, mcin, chain_length, *sd, loc, samples, initCond);
    double err = fabs(acc - target);
    if (err < min_error) {
      *best_sd = *sd;
      min_error = err;
      MH_SAY("%s\tAccepted: rwsd = %5.3f, acceptance = %4.3f, error = %4.3f\n", indent, *best_sd, acc, min_error);
    } else {
      MH_SAY("%s\trwsd = %5.3f, acceptance = %4.3f, error = %4.3f\n", indent, *sd, acc, err);
    }
    if (min_error < error_tolerance) break;
    *sd *= acc / target;
  }
}

/* ----------------------------------------------------------------- tuners */

/* mcmc_gpu.cu:381-474 */
void tune_target_a_gpu_v2(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, gpu_v_str gpu,
                          double *initCond, double ratio, int max_reps)
{
  (void)gpu;
  int owned = 0;
  mhlik_ctx *lik = mh_acquire(data, mcin, MHLIK_F64, &owned);
  mcmc_int_v v;
  mcmc_int loc;
  loc.cposterior = loc.pposterior = loc.acceptance = loc.u = 0;
  mh_alloc_vectors(&v, mcin.ddata, 1, 0);

  const int chain_length = 5000;
  double sd = mct->rwsd, best_sd = sd;
  double *samples = (double *)malloc(sizeof(double) * mcin.ddata * chain_length);
  if (!samples) fprintf(stderr, "ERROR: Samples vector did not allocated.\n");

  MH_SAY("\nStarting tuning process. Rwsd = %5.3f\n", sd);
  search_sd(lik, r, v, mcin, chain_length, max_reps, ratio, &loc, samples, initCond, &sd, &best_sd, "\t");
  mct->rwsd = best_sd;
  MH_SAY("Tuning finished. Selected rwsd = %5.3f\n\n", mct->rwsd);

  free(samples);
  mh_free_vectors(v);
  mh_release(lik, owned);
}

/* mcmc_gpu.cu:260-377 */
void tune_ess_gpu(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, gpu_v_str gpu, double *initCond,
                  int length)
{
  static const double target_a[9] = {0.10, 0.15, 0.20, 0.25, 0.30, 0.35, 0.40, 0.45, 0.50};
  (void)gpu;
  int owned = 0;
  mhlik_ctx *lik = mh_acquire(data, mcin, MHLIK_F64, &owned);
  mcmc_int_v v;
  mcmc_int loc;
  loc.cposterior = loc.pposterior = loc.acceptance = loc.u = 0;
  mh_alloc_vectors(&v, mcin.ddata, 1, 0);

  const int chain_length = length, runs = 40, lagidx = 500;
  double max_ess = -9999999;
  double sd = mct->rwsd, ess_sd = sd, best_sd = sd;
  double *samples = (double *)malloc(sizeof(double) * mcin.ddata * chain_length);
  double *autocorr = (double *)malloc(sizeof(double) * lagidx);
  if (!samples || !autocorr) fprintf(stderr, "ERROR: tuning vectors did not allocate.\n");

  MH_SAY("\nStarting tuning process. Rwsd = %5.3f\n", sd);
  for (int a = 0; a < 9; a++) {
    MH_SAY("\tStarting tuning for target ratio = %4.3f. Current rwsd = %5.3f\n", target_a[a], sd);
    search_sd(lik, r, v, mcin, chain_length, runs, target_a[a], &loc, samples, initCond, &sd, &best_sd, "\t\t");
    double circ = 0; /* circular_autocorrelation on the device (host loop: MCMC_AMD_HOST_STATS=1) */
    if (mh_device_stats_enabled())
      MH_CHECK(mhlik_autocorrelation_f64(lik, samples, chain_length, mcin.ddata, lagidx, autocorr, &circ), lik);
    else
      circ = circular_autocorrelation(autocorr, samples, mcin.ddata, chain_length, lagidx);
    double ess = chain_length / (1 + 2 * circ);
    if (ess > max_ess) {
      max_ess = ess;
      ess_sd = sd;
      MH_SAY("\tAccepted: ess = %8.3f, rwsd = %5.3f\n", max_ess, ess_sd);
    } else {
      MH_SAY("\tess= %8.3f, rwsd = %5.3f\n", ess, sd);
    }
  }
  mct->rwsd = ess_sd;
  MH_SAY("Tuning finished. Selected rwsd = %5.3f\n\n", mct->rwsd);

  free(samples);
  free(autocorr);
  mh_free_vectors(v);
  mh_release(lik, owned);
}

/* mcmc_gpu.cu:477-561 -- legacy +-10 % stepper (chain length is an argument
 * here, unlike the CPU twin's fixed 5000) */
void tune_target_a_gpu(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, gpu_v_str gpu,
                       double *initCond, int length)
{
  (void)gpu;
  int owned = 0;
  mhlik_ctx *lik = mh_acquire(data, mcin, MHLIK_F64, &owned);
  mcmc_int_v v;
  mcmc_int loc;
  loc.cposterior = loc.pposterior = loc.acceptance = loc.u = 0;
  mh_alloc_vectors(&v, mcin.ddata, 1, 0);

  const int chain_length = length;
  const double target = 0.25, tol = 0.01, step = 0.1;
  double sd = mct->rwsd;
  double *samples = (double *)malloc(sizeof(double) * mcin.ddata * chain_length);
  if (!samples) fprintf(stderr, "Samples vector did not allocated.\n");

  MH_SAY("\nStarting tuning process. Rwsd = %5.3f\n", sd);
  for (int run = 0;; run++) {
    MH_SAY("\tStarting Run %2d. Current rwsd = %5.3f, Acceptance = ", run, sd);
    double acc = tuning_round(lik, r, v, mcin, chain_length, sd, &loc, samples, initCond);
    MH_SAY("%4.3f\n", acc);
    if (acc > target + tol)
      sd *= (1 + step);
    else if (acc < target - tol)
      sd *= (1 - step);
    else
      break;
  }
  mct->rwsd = sd;
  MH_SAY("Tuning finished. Selected rwsd = %5.3f\n\n", mct->rwsd);

  free(samples);
  mh_free_vectors(v);
  mh_release(lik, owned);
}

/* mcmc_gpu.cu:660-706: the reference dumps cudaDeviceProp; here the library
 * reports what it is and how many compute units the device has. */
void print_gpu_info(void)
{
  printf(" --- %s ---\n", mhlik_version());
}
