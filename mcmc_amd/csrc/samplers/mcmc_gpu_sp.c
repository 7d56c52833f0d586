/*
 * mcmc_gpu_sp.c -- single-precision MH sampler and tuners on the MI355X.
 *
 * Follows src/mcmc_gpu_sp/mcmc_gpu_sp.cu: sp_sampler :20-84, metropolis_sp /
 * burn_in_metropolis_sp :86-177, tuners :267-481, short runs :484-559, ratio /
 * prior :561-580.  The chain state, posterior, prior sum and accept test are
 * all float there, and stay float here:
 *   - proposal  currentf + (float)ziggurat(r, (double)rwsdf)          (:105)
 *   - prior     float accumulator += log(double pdf)                  (:570-580)
 *   - accept    log(uf) <= acceptancef with uf = (float)uniform; the
 *               reference compiles as C++, where log(float) is the float
 *               overload, so this is logf                             (:112-114)
 * The likelihood call goes to mhlik_loglik_f32 (fp32 storage and row dots,
 * fp32 -log(1+exp(.)), partial sums kept in double, result rounded to float).
 *
 * Not copied: the reference frees the *double* device pointers, which it
 * never allocated, at the end of every SP entry point (:78-81, :379-382,
 * :476-479), and short_run_burn_in_sp ignores initCond (:493-494).
 */
#include "mh_core.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static const int PRIOR_SD = 10;

/* mcmc_gpu_sp.cu:570-580 */
float log_prior_sp(float *sample, mcmc_str mcin)
{
  float log_prob = 0;
  for (int k = 0; k < mcin.ddata; k++) log_prob += log(gsl_ran_gaussian_pdf(sample[k], PRIOR_SD));
  return log_prob;
}

/* mcmc_gpu_sp.cu:561-568 */
float acceptance_ratio_sp(mcmc_int_v mclocv, mcmc_int *mcloc, mcmc_str mcin, float plhood)
{
  mcloc->pposteriorf = log_prior_sp(mclocv.proposedf, mcin) + plhood;
  return mcloc->pposteriorf - mcloc->cposteriorf;
}

/* role of gpu_likelihood_f, mcmc_gpu_sp.cu:216-264 */
float gpu_likelihood_f(mhlik_ctx *lik, mcmc_str mcin, float *samplesf, out_str *res)
{
  float L = 0;
  MH_CHECK(mhlik_loglik_f32(lik, samplesf, &L), lik);
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

/* loop body of mcmc_gpu_sp.cu:101-128 and its twins */
static int step_f(mhlik_ctx *lik, gsl_rng *r, mcmc_str mcin, float sd, mcmc_int_v v, mcmc_int *loc, out_str *res)
{
  const int d = mcin.ddata;
  for (int k = 0; k < d; k++) v.proposedf[k] = v.currentf[k] + (float)gsl_ran_gaussian_ziggurat(r, (double)sd);

  const float plhood = gpu_likelihood_f(lik, mcin, v.proposedf, res);
  loc->acceptancef = acceptance_ratio_sp(v, loc, mcin, plhood);
  loc->uf = gsl_rng_uniform(r);

  const float logu = logf(loc->uf);
  const int accept = (logu <= loc->acceptancef);
  mh_trace_push(plhood, loc->acceptancef, logu, accept);
  if (accept) {
    for (int k = 0; k < d; k++) v.currentf[k] = v.proposedf[k];
    loc->cposteriorf = loc->pposteriorf;
  }
  return accept;
}

/* n iterations of step_f with the chain kept on the device (mhlik_chain_submit_f32 / _collect_f32):
 * the host draws the batch's numbers in step_f's order -- d float-rounded
 * Gaussians, then one uniform rounded to float -- and the device runs the
 * float chain: float proposal, float prior accumulator += double log(pdf),
 * float posterior and ratio, logf(uf) <= ratio. */
static void chain_steps_f(mhlik_ctx *lik, gsl_rng *r, mcmc_str mcin, float sd, mcmc_int_v v, mcmc_int *loc, long n,
                          float *rows_out, int *accepted, out_str *res)
{
  const int d = mcin.ddata;
  if (n <= 0) return;
  const long cap = n < MHLIK_CHAIN_MAX_SUBMIT ? n : MHLIK_CHAIN_MAX_SUBMIT;
  float *inc = (float *)malloc(sizeof(float) * (size_t)cap * d);
  float *logu = (float *)malloc(sizeof(float) * (size_t)cap * 2); /* per batch in flight, for the trace */
  double *lk = (double *)malloc(sizeof(double) * (size_t)cap);
  double *ratio = (double *)malloc(sizeof(double) * (size_t)cap);
  uint8_t *acc = (uint8_t *)malloc((size_t)cap);
  if (!inc || !logu || !lk || !ratio || !acc) {
    fprintf(stderr, "ERROR: chain batch vectors did not allocate.\n");
    exit(MHLIK_ENOMEM);
  }
  MH_CHECK(mhlik_chain_resume_f32(lik, v.currentf, loc->cposteriorf, (double)PRIOR_SD), lik);
  double ms_sum = 0;
  long ms_n = 0;
  /* two batches in flight: the host draws batch i+1 and files batch i-1 while the device runs batch i (mh_core.c) */
  long sub = 0, col = 0, k = 0, fly[2] = {0, 0};
  int nfly = 0, head = 0;
  while (col < n) {
    while (sub < n && nfly < 2) {
      const long m = mh_chain_batch_size(k++, n - sub);
      const int slot = (head + nfly) % 2;
      float *lu = logu + (size_t)slot * cap;
      for (long i = 0; i < m; i++) {
        for (int j = 0; j < d; j++) inc[i * d + j] = (float)gsl_ran_gaussian_ziggurat(r, (double)sd);
        loc->uf = gsl_rng_uniform(r);
        lu[i] = logf(loc->uf);
      }
      mhlik_chain_batch_f32 b;
      memset(&b, 0, sizeof b);
      b.n = m;
      b.increments = inc;
      b.log_u = lu;
      b.samples = rows_out ? rows_out + sub * d : NULL;
      MH_CHECK(mhlik_chain_submit_f32(lik, &b), lik);
      fly[slot] = m;
      nfly++;
      sub += m;
    }
    const long m = fly[head];
    const float *lu = logu + (size_t)head * cap;
    mhlik_chain_batch_f32 b;
    memset(&b, 0, sizeof b);
    b.n = m;
    b.samples = rows_out ? rows_out + col * d : NULL;
    b.lik = lk;
    b.ratio = ratio;
    b.accepted = acc;
    MH_CHECK(mhlik_chain_collect_f32(lik, &b), lik);
    if (accepted) *accepted += (int)b.n_accepted;
    for (long i = 0; i < m; i++) mh_trace_push(lk[i], ratio[i], lu[i], acc[i]);
    loc->acceptancef = (float)ratio[m - 1];
    if (b.kernel_samples > 0) {
      ms_sum += (double)b.kernel_ms * b.kernel_samples;
      ms_n += b.kernel_samples;
    }
    head = (head + 1) % 2;
    nfly--;
    col += m;
  }
  MH_CHECK(mhlik_chain_state_f32(lik, v.currentf, &loc->cposteriorf), lik);
  if (res && ms_n > 0) {
    float ms = 0;
    int64_t bytes = 0;
    mhlik_last_kernel_ms(lik, &ms, &bytes);
    ms = (float)(ms_sum / ms_n);
    res->kernelTime = ms;
    res->kernelBandwidth = (float)((double)bytes / ms / 1e6);
    res->gpuTime = res->kernelTime;
    res->gpuBandwidth = res->kernelBandwidth;
  }
  free(inc);
  free(logu);
  free(lk);
  free(ratio);
  free(acc);
}

static void run_steps_f(mhlik_ctx *lik, gsl_rng *r, mcmc_str mcin, float sd, mcmc_int_v v, mcmc_int *loc, long n,
                        float *rows, int *accepted, out_str *res)
{
  const int d = mcin.ddata;
  if (mh_device_chain_enabled()) {
    chain_steps_f(lik, r, mcin, sd, v, loc, n, rows, accepted, res);
    return;
  }
  for (long i = 0; i < n; i++) {
    if (step_f(lik, r, mcin, sd, v, loc, res) && accepted) *accepted += 1;
    if (rows)
      for (int k = 0; k < d; k++) rows[i * d + k] = v.currentf[k];
  }
}

static void start_chain_f(mhlik_ctx *lik, mcmc_str mcin, mcmc_int_v v, mcmc_int *loc, const float *theta0)
{
  for (int k = 0; k < mcin.ddata; k++) v.currentf[k] = theta0[k];
  float clhood = gpu_likelihood_f(lik, mcin, v.currentf, NULL);
  loc->cposteriorf = log_prior_sp(v.currentf, mcin) + clhood;
}

/* mcmc_gpu_sp.cu:20-84 */
void sp_sampler(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata, gpu_v_str gpu,
                out_str *res)
{
  (void)gpu;
  int owned = 0;
  mhlik_ctx *lik = mh_acquire(data, mcin, MHLIK_F32, &owned);
  const int d = mcin.ddata;

  mcmc_int_v v;
  mcmc_int loc;
  loc.cposteriorf = loc.pposteriorf = loc.acceptancef = loc.uf = 0;
  mh_alloc_vectors(&v, d, 0, 1);

  clock_t t0 = clock();
  if (mcin.burnin != 0) { /* burn_in_metropolis_sp, :132-177 */
    MH_SAY("Starting burn in process. Selected rwsdf = %f\n", mct.rwsdf);
    start_chain_f(lik, mcin, v, &loc, mcdata.burnf);
    run_steps_f(lik, r, mcin, mct.rwsdf, v, &loc, mcin.burnin - 1, mcdata.burnf + d, NULL, NULL);
    MH_SAY("Burn in process finished.\n\n");
  }
  clock_t burn = clock() - t0;

  int accepted = 0;
  t0 = clock();
  { /* metropolis_sp, :86-130 */
    res->cuTime = res->cuBandwidth = 0;
    res->kernelTime = res->kernelBandwidth = 0;
    res->gpuTime = res->gpuBandwidth = 0;
    MH_SAY("Starting metropolis algorithm. Selected rwsdf = %f\n", mct.rwsdf);
    run_steps_f(lik, r, mcin, mct.rwsdf, v, &loc, mcin.Ns, mcdata.samplesf, &accepted, res);
    MH_SAY("Metropolis algorithm finished. Accepted Samples = %d\n\n", accepted);
  }
  clock_t run = clock() - t0;

  res->burnTime = burn * 1000 / CLOCKS_PER_SEC;
  res->mcmcTime = run * 1000 / CLOCKS_PER_SEC;
  res->acceptance = (float)accepted / mcin.Ns;

  mh_free_vectors(v);
  mh_release(lik, owned);
}

/* short_run_burn_in_sp + short_run_metropolis_sp, :484-559 */
static float tuning_round_f(mhlik_ctx *lik, gsl_rng *r, mcmc_int_v v, mcmc_str mcin, int chain_length, float sd,
                            mcmc_int *loc, float *samples, const float *initCond)
{
  int accepted = 0;
  start_chain_f(lik, mcin, v, loc, initCond);
  run_steps_f(lik, r, mcin, sd, v, loc, mcin.burnin - 1, NULL, NULL, NULL);
  run_steps_f(lik, r, mcin, sd, v, loc, chain_length, samples, &accepted, NULL);
  return accepted / (float)chain_length;
}

/* search loop of :441-469 / :330-358, float arithmetic throughout */
static void search_sd_f(mhlik_ctx *lik, gsl_rng *r, mcmc_int_v v, mcmc_str mcin, int chain_length, int runs,
                        float target, mcmc_int *loc, float *samples, const float *initCond, float *sd,
                        float *best_sd, const char *indent)
{
  const float error_tolerance = 0.01;
  float min_error = 9999999;
  for (int run = 0; run < runs; run++) {
    MH_SAY("%sStarting Run %2d. Current rwsdf = %5.3f\n", indent, run, *sd);
    float acc = tuning_round_f(lik, r, v, mcin, chain_length, *sd, loc, samples, initCond);
    float err = fabsf(acc - target);
    if (err < min_error) {
      *best_sd = *sd;
      min_error = err;
      MH_SAY("%s\tAccepted: rwsdf = %5.3f, acceptance = %4.3f, error = %4.3f\n", indent, *best_sd, acc, min_error);
    } else {
      MH_SAY("%s\trwsdf = %5.3f, acceptance = %4.3f, error = %4.3f\n", indent, *sd, acc, err);
    }
    if (min_error < error_tolerance) break;
    *sd *= acc / target;
  }
}

/* mcmc_gpu_sp.cu:388-481 */
void tune_target_a_sp_v2(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, gpu_v_str gpu,
                         float *initCond, float ratio, int max_reps)
{
  (void)gpu;
  int owned = 0;
  mhlik_ctx *lik = mh_acquire(data, mcin, MHLIK_F32, &owned);
  mcmc_int_v v;
  mcmc_int loc;
  loc.cposteriorf = loc.pposteriorf = loc.acceptancef = loc.uf = 0;
  mh_alloc_vectors(&v, mcin.ddata, 0, 1);

  const int chain_length = 5000;
  float sd = mct->rwsdf, best_sd = sd;
  float *samples = (float *)malloc(sizeof(float) * mcin.ddata * chain_length);
  if (!samples) fprintf(stderr, "ERROR: Samples vector did not allocated.\n");

  MH_SAY("\nStarting tuning process. rwsdf = %5.3f\n", sd);
  search_sd_f(lik, r, v, mcin, chain_length, max_reps, ratio, &loc, samples, initCond, &sd, &best_sd, "\t");
  mct->rwsdf = best_sd;
  MH_SAY("Tuning finished. Selected rwsdf = %5.3f\n\n", mct->rwsdf);

  free(samples);
  mh_free_vectors(v);
  mh_release(lik, owned);
}

/* mcmc_gpu_sp.cu:267-384 */
void tune_ess_sp(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, gpu_v_str gpu, float *initCond,
                 int length)
{
  static const float target_a[9] = {0.10, 0.15, 0.20, 0.25, 0.30, 0.35, 0.40, 0.45, 0.50};
  (void)gpu;
  int owned = 0;
  mhlik_ctx *lik = mh_acquire(data, mcin, MHLIK_F32, &owned);
  mcmc_int_v v;
  mcmc_int loc;
  loc.cposteriorf = loc.pposteriorf = loc.acceptancef = loc.uf = 0;
  mh_alloc_vectors(&v, mcin.ddata, 0, 1);

  const int chain_length = length, runs = 40, lagidx = 500;
  float max_ess = -9999999;
  float sd = mct->rwsdf, ess_sd = sd, best_sd = sd;
  float *samples = (float *)malloc(sizeof(float) * mcin.ddata * chain_length);
  float *autocorr = (float *)malloc(sizeof(float) * lagidx);
  if (!samples || !autocorr) fprintf(stderr, "ERROR: tuning vectors did not allocate.\n");

  MH_SAY("\nStarting tuning process. rwsdf = %5.3f\n", sd);
  for (int a = 0; a < 9; a++) {
    MH_SAY("\tStarting tuning for target ratio = %4.3f. Current rwsdf = %5.3f\n", target_a[a], sd);
    search_sd_f(lik, r, v, mcin, chain_length, runs, target_a[a], &loc, samples, initCond, &sd, &best_sd, "\t\t");
    float circ = 0; /* circular_autocorrelation_sp on the device (host loop: MCMC_AMD_HOST_STATS=1) */
    if (mh_device_stats_enabled())
      MH_CHECK(mhlik_autocorrelation_f32(lik, samples, chain_length, mcin.ddata, lagidx, autocorr, &circ), lik);
    else
      circ = circular_autocorrelation_sp(autocorr, samples, mcin.ddata, chain_length, lagidx);
    float ess = chain_length / (1 + 2 * circ);
    if (ess > max_ess) {
      max_ess = ess;
      ess_sd = sd;
      MH_SAY("\tAccepted: ess = %8.3f, rwsdf = %5.3f\n", max_ess, ess_sd);
    } else {
      MH_SAY("\tess= %8.3f, rwsdf = %5.3f\n", ess, sd);
    }
  }
  mct->rwsdf = ess_sd;
  MH_SAY("Tuning finished. Selected rwsdf = %5.3f\n\n", mct->rwsdf);

  free(samples);
  free(autocorr);
  mh_free_vectors(v);
  mh_release(lik, owned);
}
