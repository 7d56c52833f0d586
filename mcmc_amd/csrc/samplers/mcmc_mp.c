/*
 * mcmc_mp.c -- mixed-precision MH sampler on the MI355X.
 *
 * Follows src/mcmc_mp/mcmc_mp.cu: mp_sampler :283-366, metropolis_mp :368-418,
 * burn_in_metropolis_mp :420-473, mp_likelihood :475-552, acceptance / prior
 * :629-648, get_lowerbound :652-744, get_blockSz :824-943, short_run_mp
 * :945-995.  Kept from it: prior N(0, 5^2) (:16), the accept test
 * u <= exp(log ratio) (:402, :635), the chain start at theta = 1 (:433), the
 * draw order per iteration (d Gaussians, gsl_rng_uniform_int for the
 * resampling phase inside the likelihood, then one uniform).
 *
 * The reference marks this path "in progress" (README.md:10,14): it prints
 * debug lines every iteration, takes log() of a log in the re-examined dark
 * branch (:63-65, NaN), frees a cudaMallocManaged pointer with free() (:939)
 * and draws the per-block uniforms from cuRAND in atomic-counter order, which
 * is not reproducible.  Here the evident intent is implemented (see
 * mhlik_mp.hpp) with counter-based per-block uniforms.
 */
#include "mh_core.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static const int PRIOR_SD = 5;      /* mcmc_mp.cu:16 */
static const int MIN_BLOCKSZ = 64;  /* :17 */
static const int MAX_BLOCKSZ = 1024; /* :18 */
static const int LOWERBOUND_REPS = 50000; /* :23 */
static const uint64_t U_SEED = 1234ULL;   /* curandSetPseudoRandomGeneratorSeed, :329 */

/* mcmc_mp.cu:638-648 */
double log_prior_mp(double *sample, mcmc_str mcin) { return mh_log_prior_d(sample, mcin.ddata, PRIOR_SD); }

/* mcmc_mp.cu:629-636 -- returns the ratio itself, not its log */
double acceptance_ratio_mp(mcmc_int_v mclocv, mcmc_int *mcloc, mcmc_str mcin, double plhood)
{
  mcloc->pposterior = log_prior_mp(mclocv.proposed, mcin) + plhood;
  return exp(mcloc->pposterior - mcloc->cposterior);
}

typedef struct {
  mhlik_ctx *lik;
  gsl_rng *r;
  mp_str *mp;
  double bound;
  uint64_t calls;
} mp_env;

/* mp_likelihood, mcmc_mp.cu:475-552 */
static double mp_likelihood(mp_env *e, const double *theta)
{
  mhlik_mp_state st;
  st.block_rows = e->mp->dataBlockSz;
  st.blocks = e->mp->dataBlocks;
  st.resamp_factor = e->mp->ResampFactor;
  st.resamp_choose = (int)gsl_rng_uniform_int(e->r, (unsigned long)e->mp->ResampFactor); /* :504 */
  st.u_seed = U_SEED;
  st.u_call = e->calls++;
  st.bound_eps = e->bound;
  double L = 0;
  int32_t nb = 0, nd = 0;
  MH_CHECK(mhlik_mp_loglik(e->lik, &st, theta, &L, &nb, &nd), e->lik);
  e->mp->brightBlocks = nb; /* :547-549 */
  e->mp->darkBlocks = nd;
  e->mp->ratio = ((float)nb) / e->mp->dataBlocks;
  return L;
}

static double lik_cb(void *env, mcmc_str mcin, mcmc_int_v v, out_str *res)
{
  (void)mcin;
  (void)res;
  return mp_likelihood((mp_env *)env, v.proposed);
}

static mhlik_ctx *mp_context(data_str data, mcmc_str mcin, int *owned)
{
  mhlik_ctx *att = mcmc_amd_attached_context();
  *owned = 0;
  if (att) {
    mhlik_info info;
    if (mhlik_get_info(att, &info) == MHLIK_OK && info.n_rows == (int64_t)mcin.Nd && info.n_cols == mcin.ddata &&
        info.precision == (MHLIK_F64 | MHLIK_F32))
      return att;
  }
  mhlik_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_rows = mcin.Nd;
  cfg.n_cols = mcin.ddata;
  cfg.precision = MHLIK_F64 | MHLIK_F32;
  cfg.device = 0;
  cfg.world = 1;
  cfg.segments = 1; /* data blocks are counted from row 0 */
  cfg.timing = 1;
  mhlik_ctx *lik = NULL;
  int rc = mhlik_create(&cfg, &lik);
  if (rc != MHLIK_OK) {
    fprintf(stderr, "GPUassert: %s %s %d\n", mhlik_last_error(NULL), __FILE__, __LINE__);
    exit(rc);
  }
  if (!data.data || !data.dataf) {
    fprintf(stderr, "ERROR: mp path needs data.data and data.dataf.\n");
    exit(MHLIK_EINVAL);
  }
  MH_CHECK(mhlik_upload_f64(lik, data.data, mcin.Nd, 0), lik);  /* :320 */
  MH_CHECK(mhlik_upload_f32(lik, data.dataf, mcin.Nd, 0), lik); /* :321 */
  *owned = 1;
  return lik;
}

static void start_chain_mp(mp_env *e, mcmc_str mcin, mcmc_int_v v, mcmc_int *loc)
{
  for (int k = 0; k < mcin.ddata; k++) { /* :430-435: theta0 = 1, not burn[0] */
    v.current[k] = 1;
    v.currentf[k] = (float)v.current[k];
  }
  double clhood = mp_likelihood(e, v.current);
  loc->cposterior = log_prior_mp(v.current, mcin) + clhood;
}

/* mcmc_mp.cu:283-366 */
void mp_sampler(data_str data, gsl_rng *r, mp_str *mpVar, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata,
                double lhoodBound, out_str *res)
{
  int owned = 0;
  mhlik_ctx *lik = mp_context(data, mcin, &owned);
  const int d = mcin.ddata;

  mcmc_int_v v;
  mcmc_int loc;
  loc.cposterior = loc.pposterior = loc.acceptance = loc.u = 0;
  mh_alloc_vectors(&v, d, 1, 1);

  MH_CHECK(mhlik_mp_reset(lik, mpVar->dataBlockSz, mpVar->dataBlocks), lik);
  mpVar->brightBlocks = mpVar->dataBlocks; /* :333-334 */
  mpVar->darkBlocks = 0;
  mp_env env = {lik, r, mpVar, lhoodBound, 0};

  clock_t t0 = clock();
  if (mcin.burnin != 0) { /* burn_in_metropolis_mp, :420-473 */
    MH_SAY("Starting burn in process. Selected rwsd = %f\n", mct.rwsd);
    start_chain_mp(&env, mcin, v, &loc);
    for (int i = 1; i < mcin.burnin; i++) {
      mh_step_d(lik_cb, &env, r, mcin, mct.rwsd, v, &loc, NULL, PRIOR_SD, 1);
      for (int k = 0; k < d; k++) mcdata.burn[(long)i * d + k] = v.current[k];
    }
    MH_SAY("Burn in process finished.\n\n");
  }
  clock_t burn = clock() - t0;

  int accepted = 0;
  t0 = clock();
  { /* metropolis_mp, :368-418 */
    res->cuTime = res->cuBandwidth = 0;
    res->kernelTime = res->kernelBandwidth = 0;
    res->gpuTime = res->gpuBandwidth = 0;
    MH_SAY("Starting metropolis algorithm. Selected rwsd = %f\n", mct.rwsd);
    for (int i = 0; i < mcin.Ns; i++) {
      if (mh_step_d(lik_cb, &env, r, mcin, mct.rwsd, v, &loc, res, PRIOR_SD, 1)) accepted += 1;
      for (int k = 0; k < d; k++) mcdata.samples[(long)i * d + k] = v.current[k];
    }
    MH_SAY("Metropolis algorithm finished. Accepted Samples = %d\n\n", accepted);
  }
  clock_t run = clock() - t0;

  res->burnTime = burn * 1000 / CLOCKS_PER_SEC;
  res->mcmcTime = run * 1000 / CLOCKS_PER_SEC;
  res->acceptance = (double)accepted / mcin.Ns;

  mh_free_vectors(v);
  mh_release(lik, owned);
}

/* ---- lower bound search, mcmc_mp.cu:652-744 ------------------------------- */

typedef struct {
  mhlik_ctx *lik;
  double eps_last;
} bound_env;

/* likelihood_bound + call_boundKernel, :746-821: returns the sum of the
 * double sigmoids (that is what the reference accumulates and uses as the
 * "likelihood" of this search) and records the epsilon extreme. */
static double bound_cb(void *env, mcmc_str mcin, mcmc_int_v v, out_str *res)
{
  (void)mcin;
  (void)res;
  bound_env *e = (bound_env *)env;
  double s = 0, eps = 0;
  MH_CHECK(mhlik_mp_bound(e->lik, v.proposed, &s, &eps), e->lik);
  e->eps_last = eps;
  return s;
}

double get_lowerbound(data_str data, double *initialCond, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct)
{
  (void)initialCond; /* the reference starts from 0 whatever is passed (:687-688) */
  int owned = 0;
  mhlik_ctx *lik = mp_context(data, mcin, &owned);
  mcmc_int_v v;
  mcmc_int loc;
  loc.cposterior = loc.pposterior = loc.acceptance = loc.u = 0;
  mh_alloc_vectors(&v, mcin.ddata, 1, 1);
  bound_env env = {lik, 0.0};

  double global_epsilon = 0;
  MH_SAY("Starting lowerbound process. Current Lowerbound = %f\n", global_epsilon);
  for (int k = 0; k < mcin.ddata; k++) v.current[k] = v.proposed[k] = 0;
  double clhood = bound_cb(&env, mcin, v, NULL); /* proposed == current == 0 */
  loc.cposterior = log_prior_mp(v.current, mcin) + clhood;
  global_epsilon = env.eps_last;
  MH_SAY("New Lowerbound = %.64f\n", global_epsilon);

  for (int i = 1; i < LOWERBOUND_REPS; i++) {
    /* the step computes the bound at the proposal; compare before accepting (:712-716) */
    int acc_before = 0;
    (void)acc_before;
    mh_step_d(bound_cb, &env, r, mcin, mct.rwsd, v, &loc, NULL, PRIOR_SD, 1);
    if (global_epsilon < env.eps_last) {
      global_epsilon = env.eps_last;
      MH_SAY("New Lowerbound = %.64f\n", global_epsilon);
    }
  }
  MH_SAY("Lowerbound process finished. Selected Lowerbound = %.64f\n", global_epsilon);

  mh_free_vectors(v);
  mh_release(lik, owned);
  return global_epsilon;
}

/* ---- block size / resampling factor tuner, mcmc_mp.cu:824-943 -------------- */

/* short_run_mp, :945-995: 500 iterations from theta = 1 */
static void short_run_mp(mp_env *e, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_int_v v, mcmc_int *loc)
{
  const int shortRunIdx = 500;
  start_chain_mp(e, mcin, v, loc);
  for (int i = 1; i < shortRunIdx; i++) mh_step_d(lik_cb, e, r, mcin, mct.rwsd, v, loc, NULL, PRIOR_SD, 1);
}

void get_blockSz(data_str data, gsl_rng *r, mp_str *mpVar, mcmc_str mcin, mcmc_tune_str mct, double *initialCond,
                 double lhoodBound)
{
  (void)initialCond;
  int owned = 0;
  mhlik_ctx *lik = mp_context(data, mcin, &owned);
  mcmc_int_v v;
  mcmc_int loc;
  loc.cposterior = loc.pposterior = loc.acceptance = loc.u = 0;
  mh_alloc_vectors(&v, mcin.ddata, 1, 1);

  int first_it = 1;
  float bestTime = 0;
  int bestBlockSz = 0, bestResamp = 0;
  mp_str cur;
  memset(&cur, 0, sizeof cur);

  for (cur.dataBlockSz = MIN_BLOCKSZ; cur.dataBlockSz <= MAX_BLOCKSZ; cur.dataBlockSz *= 2) {
    cur.dataBlocks = mcin.Nd / cur.dataBlockSz; /* ceil() of an integer division in the reference (:871) */
    if (cur.dataBlocks < 1) break;
    for (int resamp = cur.dataBlocks / 10; resamp >= 10; resamp /= 10) {
      cur.ResampFactor = resamp;
      MH_CHECK(mhlik_mp_reset(lik, cur.dataBlockSz, cur.dataBlocks), lik);
      cur.brightBlocks = cur.dataBlocks;
      cur.darkBlocks = 0;
      mp_env env = {lik, r, &cur, lhoodBound, 0};

      struct timespec t0, t1;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      short_run_mp(&env, r, mcin, mct, v, &loc);
      clock_gettime(CLOCK_MONOTONIC, &t1);
      float time = (float)((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);

      if (first_it || time < bestTime) {
        first_it = 0;
        bestTime = time;
        bestBlockSz = cur.dataBlockSz;
        bestResamp = cur.ResampFactor;
        MH_SAY("Update: New BlockSz = %d, New Resample Factor = %d\n", bestBlockSz, bestResamp);
      } else {
        MH_SAY("Current BlockSz = %d, Current Resample Factor = %d, Current Time = %fms, Best BlockSz = %d, "
               "Best Resample Factor = %d, Best Time = %fms\n",
               cur.dataBlockSz, cur.ResampFactor, time, bestBlockSz, bestResamp, bestTime);
      }
    }
  }
  if (bestBlockSz > 0) {
    mpVar->dataBlockSz = bestBlockSz;
    mpVar->ResampFactor = bestResamp;
    mpVar->dataBlocks = mcin.Nd / bestBlockSz;
  }
  MH_SAY("Block size and Resample Factor tuning finished. Selected Parameters: BlockSz = %d, DataBlocks = %d, "
         "Resample Factor = %d\n",
         mpVar->dataBlockSz, mpVar->dataBlocks, mpVar->ResampFactor);

  mh_free_vectors(v);
  mh_release(lik, owned);
}
