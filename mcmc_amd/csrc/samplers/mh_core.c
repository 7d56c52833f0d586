/*
 * mh_core.c -- see mh_core.h.
 */
#include "mh_core.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

int mcmc_amd_verbose = 1;
void mcmc_amd_set_verbose(int on) { mcmc_amd_verbose = on; }

static mhlik_ctx *g_attached = NULL;
void mcmc_amd_attach_context(mhlik_ctx *lik) { g_attached = lik; }
mhlik_ctx *mcmc_amd_attached_context(void) { return g_attached; }

static mcmc_amd_trace_rec *g_trace = NULL;
static long g_trace_cap = 0, g_trace_n = 0;

void mcmc_amd_trace_begin(mcmc_amd_trace_rec *buf, long capacity)
{
  g_trace = buf;
  g_trace_cap = capacity;
  g_trace_n = 0;
}

long mcmc_amd_trace_end(void)
{
  long n = g_trace_n;
  g_trace = NULL;
  g_trace_cap = g_trace_n = 0;
  return n;
}

void mh_trace_push(double lik, double ratio, double logu, int accept)
{
  if (!g_trace) return;
  if (g_trace_n < g_trace_cap) {
    mcmc_amd_trace_rec *t = &g_trace[g_trace_n];
    t->lik = lik;
    t->ratio = ratio;
    t->logu = logu;
    t->accept = accept;
  }
  g_trace_n++;
}

void mh_check(int status, mhlik_ctx *lik, const char *file, int line)
{
  if (status == MHLIK_OK) return;
  fprintf(stderr, "GPUassert: %s %s %d\n", mhlik_last_error(lik), file, line);
  exit(status);
}

mhlik_ctx *mh_acquire(data_str data, mcmc_str mcin, int precision, int *owned)
{
  *owned = 0;
  if (g_attached) {
    mhlik_info info;
    /* the attached context is used when its shape matches this call */
    if (mhlik_get_info(g_attached, &info) == MHLIK_OK && info.n_rows == (int64_t)mcin.Nd &&
        info.n_cols == mcin.ddata && (info.precision & precision) == precision)
      return g_attached;
  }
  mhlik_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_rows = mcin.Nd;
  cfg.n_cols = mcin.ddata;
  cfg.precision = precision;
  cfg.device = 0; /* the reference pins device 0 (mcmc_gpu.cu:28) */
  cfg.world = 1;
  cfg.timing = 1; /* out_str carries per-call kernel times (mcmc_gpu.cu:248-254) */
  mhlik_ctx *lik = NULL;
  int rc = mhlik_create(&cfg, &lik);
  if (rc != MHLIK_OK) {
    fprintf(stderr, "GPUassert: %s %s %d\n", mhlik_last_error(NULL), __FILE__, __LINE__);
    exit(rc);
  }
  if (precision & MHLIK_F64) {
    if (!data.data) {
      fprintf(stderr, "ERROR: data.data is NULL and no context is attached.\n");
      exit(MHLIK_EINVAL);
    }
    MH_CHECK(mhlik_upload_f64(lik, data.data, mcin.Nd, 0), lik);
  }
  if (precision & MHLIK_F32) {
    if (!data.dataf) {
      fprintf(stderr, "ERROR: data.dataf is NULL and no context is attached.\n");
      exit(MHLIK_EINVAL);
    }
    MH_CHECK(mhlik_upload_f32(lik, data.dataf, mcin.Nd, 0), lik);
  }
  *owned = 1;
  return lik;
}

void mh_release(mhlik_ctx *lik, int owned)
{
  if (owned) mhlik_destroy(lik);
}

void mh_alloc_vectors(mcmc_int_v *v, int d, int want_double, int want_float)
{
  v->proposed = v->current = NULL;
  v->proposedf = v->currentf = NULL;
  if (want_double) {
    v->proposed = (double *)calloc((size_t)d, sizeof(double));
    v->current = (double *)calloc((size_t)d, sizeof(double));
    if (!v->proposed || !v->current) fprintf(stderr, "ERROR: MH work vectors did not allocate.\n");
  }
  if (want_float) {
    v->proposedf = (float *)calloc((size_t)d, sizeof(float));
    v->currentf = (float *)calloc((size_t)d, sizeof(float));
    if (!v->proposedf || !v->currentf) fprintf(stderr, "ERROR: MH float work vectors did not allocate.\n");
  }
}

void mh_free_vectors(mcmc_int_v v)
{
  free(v.proposed);
  free(v.current);
  free(v.proposedf);
  free(v.currentf);
}

/* sum_j log N(theta_j; 0, prior_sd) -- mcmc_gpu.cu:648-658, mcmc_mp.cu:638-648 */
double mh_log_prior_d(const double *sample, int d, int prior_sd)
{
  double log_prob = 0;
  for (int k = 0; k < d; k++) log_prob += log(gsl_ran_gaussian_pdf(sample[k], prior_sd));
  return log_prob;
}

/* The loop body of mcmc_gpu.cu:101-128 (and its burn-in / short-run twins,
 * :150-173, :580-599, :612-636; MP: mcmc_mp.cu:384-416). */
int mh_step_d(mh_lik_d lik, void *env, gsl_rng *r, mcmc_str mcin, double sd, mcmc_int_v v, mcmc_int *loc,
              out_str *res, int prior_sd, int exp_rule)
{
  const int d = mcin.ddata;
  for (int k = 0; k < d; k++) {
    v.proposed[k] = v.current[k] + gsl_ran_gaussian_ziggurat(r, sd);
    if (v.proposedf) v.proposedf[k] = (float)v.proposed[k]; /* mcmc_mp.cu:390 */
  }

  const double plhood = lik(env, mcin, v, res);

  loc->pposterior = mh_log_prior_d(v.proposed, d, prior_sd) + plhood;
  const double log_ratio = loc->pposterior - loc->cposterior;
  int accept;
  if (exp_rule) {
    loc->acceptance = exp(log_ratio); /* acceptance_ratio_mp, mcmc_mp.cu:635 */
    loc->u = gsl_rng_uniform(r);
    accept = (loc->u <= loc->acceptance); /* mcmc_mp.cu:402 */
    mh_trace_push(plhood, loc->acceptance, loc->u, accept);
  } else {
    loc->acceptance = log_ratio;
    loc->u = gsl_rng_uniform(r);
    const double logu = log(loc->u);
    accept = (logu <= loc->acceptance); /* mcmc_gpu.cu:114 */
    mh_trace_push(plhood, loc->acceptance, logu, accept);
  }
  if (accept) {
    for (int k = 0; k < d; k++) v.current[k] = v.proposed[k];
    if (v.currentf)
      for (int k = 0; k < d; k++) v.currentf[k] = (float)v.current[k];
    loc->cposterior = loc->pposterior;
  }
  return accept;
}

/* ---- chain on the device (mhlik_chain_*, include/mhlik.h) ----------------
 * The same iterations as n calls of mh_step_d, with the host drawing the
 * random numbers of the whole batch first -- in the order the loop above
 * consumes them: d Gaussians, then one uniform, per iteration -- and the
 * device doing proposal, likelihood, prior, ratio and accept test back to
 * back.  MCMC_AMD_HOST_LOOP=1 keeps the per-iteration host loop instead. */
int mh_device_chain_enabled(void)
{
  const char *e = getenv("MCMC_AMD_HOST_LOOP");
  return (e && e[0] == '1') ? 0 : 1;
}

/* tune_ess_*: autocorrelation of the trial chains on the device (row f4) */
int mh_device_stats_enabled(void)
{
  const char *e = getenv("MCMC_AMD_HOST_STATS");
  return (e && e[0] == '1') ? 0 : 1;
}

/* Batch sizes of the submit / collect pipeline: a short first batch so that the device starts after a few dozen
 * draws, then longer ones; two batches are in flight, so the host draws batch i+1 and files batch i-1 while the
 * device runs batch i.  A run of up to 64 iterations is one batch. */
long mh_chain_batch_size(long k, long left)
{
  static const long ramp[] = {64, 256, 1024};
  const long m = k < 3 ? ramp[k] : MHLIK_CHAIN_MAX_SUBMIT;
  return left < m ? left : m;
}

void mh_chain_steps_d(mhlik_ctx *lik, gsl_rng *r, mcmc_str mcin, double sd, mcmc_int_v v, mcmc_int *loc, long n,
                      double *rows_out, int *accepted, out_str *res, int prior_sd)
{
  const int d = mcin.ddata;
  if (n <= 0) return;
  const long cap = n < MHLIK_CHAIN_MAX_SUBMIT ? n : MHLIK_CHAIN_MAX_SUBMIT;
  /* the library copies increments and log u at submit, so one set of input arrays serves every batch; log u is kept
   * per batch in flight for the trace */
  double *inc = (double *)malloc(sizeof(double) * (size_t)cap * d);
  double *logu = (double *)malloc(sizeof(double) * (size_t)cap * 2);
  double *lk = (double *)malloc(sizeof(double) * (size_t)cap);
  double *ratio = (double *)malloc(sizeof(double) * (size_t)cap);
  uint8_t *acc = (uint8_t *)malloc((size_t)cap);
  if (!inc || !logu || !lk || !ratio || !acc) {
    fprintf(stderr, "ERROR: chain batch vectors did not allocate.\n");
    exit(MHLIK_ENOMEM);
  }
  MH_CHECK(mhlik_chain_resume_f64(lik, v.current, loc->cposterior, (double)prior_sd), lik);
  double ms_sum = 0;
  long ms_n = 0;
  long sub = 0, col = 0, k = 0, fly[2] = {0, 0};
  int nfly = 0, head = 0;
  while (col < n) {
    while (sub < n && nfly < 2) {
      const long m = mh_chain_batch_size(k++, n - sub);
      const int slot = (head + nfly) % 2;
      double *lu = logu + (size_t)slot * cap;
      for (long i = 0; i < m; i++) {
        for (int j = 0; j < d; j++) inc[i * d + j] = gsl_ran_gaussian_ziggurat(r, sd);
        loc->u = gsl_rng_uniform(r);
        lu[i] = log(loc->u);
      }
      mhlik_chain_batch_f64 b;
      memset(&b, 0, sizeof b);
      b.n = m;
      b.increments = inc;
      b.log_u = lu;
      b.samples = rows_out ? rows_out + sub * d : NULL;
      MH_CHECK(mhlik_chain_submit_f64(lik, &b), lik);
      fly[slot] = m;
      nfly++;
      sub += m;
    }
    const long m = fly[head];
    const double *lu = logu + (size_t)head * cap;
    mhlik_chain_batch_f64 b;
    memset(&b, 0, sizeof b);
    b.n = m;
    b.samples = rows_out ? rows_out + col * d : NULL;
    b.lik = lk;
    b.ratio = ratio;
    b.accepted = acc;
    MH_CHECK(mhlik_chain_collect_f64(lik, &b), lik);
    if (accepted) *accepted += (int)b.n_accepted;
    for (long i = 0; i < m; i++) mh_trace_push(lk[i], ratio[i], lu[i], acc[i]);
    loc->acceptance = ratio[m - 1];
    if (b.kernel_samples > 0) {
      ms_sum += (double)b.kernel_ms * b.kernel_samples;
      ms_n += b.kernel_samples;
    }
    head = (head + 1) % 2;
    nfly--;
    col += m;
  }
  MH_CHECK(mhlik_chain_state_f64(lik, v.current, &loc->cposterior), lik);
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

/* ---- RNG set-up: src/util/alloc_util.c:249-277 ---- */
void init_rng(gsl_rng **r)
{
  gsl_rng_env_setup();
  *r = gsl_rng_alloc(gsl_rng_default);
  if (*r == NULL) fprintf(stderr, "ERROR: RNG Allocation did not complete successfully!\n");
}

void init_rng_seed(gsl_rng **r, unsigned long int seed)
{
  gsl_rng_default_seed = seed;
  gsl_rng_env_setup();
  *r = gsl_rng_alloc(gsl_rng_default);
  if (*r == NULL) fprintf(stderr, "ERROR: RNG Allocation did not complete successfully!\n");
}

void free_rng(gsl_rng *r) { gsl_rng_free(r); }
