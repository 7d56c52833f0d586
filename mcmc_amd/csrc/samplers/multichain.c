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
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static const int PRIOR_SD = 10;

static double kc_now(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

/* ------------------------------------------------------------------------- *
 * The lock-step loop on the device (mhlik_kchain_*).
 *
 * What an iteration takes from chain k's generator does not depend on any
 * chain's state -- d Gaussians, [mixed precision: the re-examination phase,]
 * one uniform, in the reference's order (mcmc_gpu.cu:105,111; mcmc_mp.cu:389,
 * 504,400) -- so the host draws the numbers of a whole chunk of iterations,
 * chain by chain from r[k], and the device runs the chunk without a host round
 * trip.  A helper thread draws chunk i+1 while the device runs chunk i (the
 * generators are touched by one thread at a time), so the host costs the
 * device nothing.  MCMC_AMD_HOST_LOOP=1 keeps the per-iteration host loops
 * below; the chains are the same either way (tests/test_gpu_golden.py).
 * ------------------------------------------------------------------------- */
enum { KC_CHUNK = 256 };

typedef struct {
  int K, d, mp, shared_phase;
  unsigned long resamp_factor;
  gsl_rng **r;
  double sd;
  long m;          /* iterations to draw */
  double *inc;     /* [m][K][d] */
  double *u;       /* [m][K]: log(u), or u itself for the mixed-precision rule */
  int32_t *choose; /* [m][K] (mp) */
} kc_draw_job;

static void *kc_draw(void *arg)
{
  kc_draw_job *j = (kc_draw_job *)arg;
  const int K = j->K, d = j->d;
  for (long it = 0; it < j->m; it++) {
    for (int k = 0; k < K; k++) {
      double *inc = j->inc + ((size_t)it * K + k) * d;
      for (int c = 0; c < d; c++) inc[c] = gsl_ran_gaussian_ziggurat(j->r[k], j->sd);
      if (j->mp) j->choose[(size_t)it * K + k] = (int32_t)gsl_rng_uniform_int(j->r[k], j->resamp_factor);
      const double u = gsl_rng_uniform(j->r[k]);
      j->u[(size_t)it * K + k] = j->mp ? u : log(u);
    }
    if (j->mp && j->shared_phase)
      for (int k = 1; k < K; k++) j->choose[(size_t)it * K + k] = j->choose[(size_t)it * K];
  }
  return NULL;
}

typedef struct {
  double *inc[2], *u[2];
  int32_t *choose[2];
  double *rows[2]; /* [KC_CHUNK][K][d] states after each iteration */
  uint8_t *acc[2]; /* [KC_CHUNK][K] */
  int pin_inc[2], pin_rows[2]; /* which of inc / rows came from mhlik_host_alloc (the rest is malloc'ed) */
} kc_buffers;

/* page-locked when the library can provide it (full-rate copies), ordinary memory otherwise */
static void *kc_host(size_t bytes, int *pinned)
{
  void *p = NULL;
  if (mhlik_host_alloc(bytes, &p) == MHLIK_OK && p) {
    *pinned = 1;
    return p;
  }
  *pinned = 0;
  return malloc(bytes);
}

/* The chunk buffers are kept between sampler calls of the same shape (page-locked allocation costs milliseconds);
 * they are released when a call of another shape arrives, and left to the OS at exit.  g_kc_mu guards the cache: a
 * second caller that arrives while it is in use (another thread, a nested call) takes buffers of its own. */
static kc_buffers g_kc;
static int g_kc_K = 0, g_kc_d = 0, g_kc_mp = 0, g_kc_busy = 0;
static pthread_mutex_t g_kc_mu = PTHREAD_MUTEX_INITIALIZER;

static void kc_alloc(kc_buffers *b, int K, int d, int mp)
{
  for (int i = 0; i < 2; i++) {
    b->inc[i] = (double *)kc_host(sizeof(double) * (size_t)KC_CHUNK * K * d, &b->pin_inc[i]);
    b->rows[i] = (double *)kc_host(sizeof(double) * (size_t)KC_CHUNK * K * d, &b->pin_rows[i]);
    b->u[i] = (double *)malloc(sizeof(double) * (size_t)KC_CHUNK * K);
    b->acc[i] = (uint8_t *)malloc((size_t)KC_CHUNK * K);
    b->choose[i] = mp ? (int32_t *)malloc(sizeof(int32_t) * (size_t)KC_CHUNK * K) : NULL;
    if (!b->inc[i] || !b->rows[i] || !b->u[i] || !b->acc[i] || (mp && !b->choose[i])) {
      fprintf(stderr, "ERROR: K-chain batch vectors did not allocate.\n");
      exit(MHLIK_ENOMEM);
    }
  }
}

static void kc_free_one(void *p, int pinned)
{
  if (pinned)
    mhlik_host_free(p);
  else
    free(p);
}

static void kc_free(kc_buffers *b)
{
  for (int i = 0; i < 2; i++) {
    kc_free_one(b->inc[i], b->pin_inc[i]);
    kc_free_one(b->rows[i], b->pin_rows[i]);
    free(b->u[i]);
    free(b->acc[i]);
    free(b->choose[i]);
  }
}

/* Where the rows of a finished chunk go: rows 0..nburn-1 of the run to burn[k][it+1], the rest to samples[k][.]. */
typedef struct {
  int K, d;
  long nburn, Ns, bl, first, m;
  const double *rows;
  const uint8_t *acc;
  double *samples, *burn;
  int *accepted;
} kc_scatter_job;

static void kc_scatter(const kc_scatter_job *s)
{
  const int K = s->K, d = s->d;
  for (long i = 0; i < s->m; i++) {
    const long it = s->first + i;
    const int sampling = it >= s->nburn;
    for (int k = 0; k < K; k++) {
      const double *row = s->rows + ((size_t)i * K + k) * d;
      if (sampling) {
        memcpy(s->samples + ((long)k * s->Ns + (it - s->nburn)) * d, row, sizeof(double) * d);
        s->accepted[k] += s->acc[(size_t)i * K + k];
      } else {
        memcpy(s->burn + ((long)k * s->bl + it + 1) * d, row, sizeof(double) * d);
      }
    }
  }
}

/* what the helper thread does while the device runs a chunk: file the previous chunk's rows, draw the next chunk */
typedef struct {
  kc_scatter_job *scatter; /* NULL: nothing to file */
  kc_draw_job *draw;       /* NULL: nothing to draw */
} kc_helper_job;

static void *kc_helper(void *arg)
{
  kc_helper_job *h = (kc_helper_job *)arg;
  if (h->scatter) kc_scatter(h->scatter);
  if (h->draw) kc_draw(h->draw);
  return NULL;
}

/* Chunk sizes: the first chunk's numbers are drawn, and the last chunk's rows filed, with the device idle, so both are
 * short -- 16, 64, then KC_CHUNK-sized chunks, then 64 and 16 again at the end of the run. */
static long kc_next_chunk(long nth, long remaining)
{
  long cap = nth == 0 ? 16 : nth == 1 ? 64 : KC_CHUNK;
  if (remaining <= 16) return remaining;
  if (remaining <= 80) return remaining - 16 < cap ? remaining - 16 : cap;
  return remaining - 80 < cap ? remaining - 80 : cap;
}

/* total iterations from the device-side state. */
static void kc_run(mhlik_ctx *lik, int K, int d, gsl_rng **r, double sd, long nburn, long Ns, long bl, double *samples,
                   double *burn, int *accepted, const mp_str *mp, double bound, int shared_phase, const uint64_t *seeds,
                   uint64_t first_call, int32_t *bright, int32_t *dark)
{
  const long total = nburn + Ns;
  kc_buffers B;
  int cached = 0;
  pthread_mutex_lock(&g_kc_mu);
  if (!g_kc_busy) {
    if (g_kc_K && (g_kc_K != K || g_kc_d != d || g_kc_mp != (mp != NULL))) {
      kc_free(&g_kc);
      g_kc_K = 0;
    }
    if (!g_kc_K) {
      kc_alloc(&g_kc, K, d, mp != NULL);
      g_kc_K = K;
      g_kc_d = d;
      g_kc_mp = mp != NULL;
    }
    B = g_kc;
    g_kc_busy = cached = 1;
  }
  pthread_mutex_unlock(&g_kc_mu);
  if (!cached) kc_alloc(&B, K, d, mp != NULL);
  kc_draw_job job[2];
  for (int i = 0; i < 2; i++) {
    job[i].K = K;
    job[i].d = d;
    job[i].mp = mp != NULL;
    job[i].shared_phase = shared_phase;
    job[i].resamp_factor = mp ? (unsigned long)mp->ResampFactor : 1;
    job[i].r = r;
    job[i].sd = sd;
    job[i].inc = B.inc[i];
    job[i].u = B.u[i];
    job[i].choose = B.choose[i];
  }
  kc_scatter_job sc;
  sc.K = K;
  sc.d = d;
  sc.nburn = nburn;
  sc.Ns = Ns;
  sc.bl = bl;
  sc.samples = samples;
  sc.burn = burn;
  sc.accepted = accepted;
  long done = 0, nth = 0;
  int cur = 0, have_prev = 0;
  const int kc_timing = getenv("MCMC_AMD_KC_TIMING") != NULL;
  const double t_first = kc_now();
  job[0].m = kc_next_chunk(0, total);
  kc_draw(&job[0]);
  if (kc_timing) fprintf(stderr, "kc_run: first draw of %ld its %.3f ms\n", job[0].m, kc_now() - t_first);
  while (done < total) {
    const long m = job[cur].m;
    const long next = total - done - m > 0 ? kc_next_chunk(++nth, total - done - m) : 0;
    kc_helper_job hj;
    hj.scatter = have_prev ? &sc : NULL; /* sc still describes the previous chunk */
    hj.draw = NULL;
    if (next > 0) {
      job[cur ^ 1].m = next;
      hj.draw = &job[cur ^ 1];
    }
    pthread_t th;
    int threaded = 0;
    if (hj.scatter || hj.draw) {
      threaded = pthread_create(&th, NULL, kc_helper, &hj) == 0;
      if (!threaded) kc_helper(&hj);
    }
    mhlik_kchain_mp mpa;
    mhlik_kchain_batch b;
    memset(&b, 0, sizeof b);
    b.n = m;
    b.increments = job[cur].inc;
    b.log_u = job[cur].u;
    b.samples = B.rows[cur];
    b.accepted = B.acc[cur];
    if (mp) {
      memset(&mpa, 0, sizeof mpa);
      mpa.resamp_factor = mp->ResampFactor;
      mpa.resamp_choose = job[cur].choose;
      mpa.u_seed = seeds;
      mpa.u_call = first_call + (uint64_t)done;
      mpa.bound_eps = bound;
      b.mp = &mpa;
      b.bright_blocks = bright;
      b.dark_blocks = dark;
    }
    const double t_run0 = kc_now();
    const int rc = mhlik_kchain_run(lik, &b);
    const double t_run1 = kc_now();
    if (threaded) pthread_join(th, NULL);
    if (kc_timing)
      fprintf(stderr, "kc_run: chunk %ld its: run %.3f ms (%.4f ms/it, kernel %.4f), helper join +%.3f ms\n", m,
              t_run1 - t_run0, (t_run1 - t_run0) / m, b.kernel_ms, kc_now() - t_run1);
    MH_CHECK(rc, lik);
    sc.first = done;
    sc.m = m;
    sc.rows = B.rows[cur];
    sc.acc = B.acc[cur];
    have_prev = 1;
    done += m;
    cur ^= 1;
  }
  const double t_last = kc_now();
  if (have_prev) kc_scatter(&sc);
  if (kc_timing) fprintf(stderr, "kc_run: last scatter %.3f ms, whole run %.3f ms\n", kc_now() - t_last, kc_now() - t_first);
  if (cached) {
    pthread_mutex_lock(&g_kc_mu);
    g_kc_busy = 0;
    pthread_mutex_unlock(&g_kc_mu);
  } else {
    kc_free(&B);
  }
}

/* ---- the per-iteration host loop (MCMC_AMD_HOST_LOOP=1) ---- */
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
  if (mh_device_chain_enabled()) { /* the lock-step loop on the device */
    const long bl = mcin.burnin > 0 ? mcin.burnin : 1;
    const long nburn = mcin.burnin > 0 ? mcin.burnin - 1 : 0;
    double *cur = (double *)calloc((size_t)K * d, sizeof(double));
    double *cpost = (double *)calloc((size_t)K, sizeof(double));
    const int mode = use_f32 ? MHLIK_KCHAIN_F32 : MHLIK_KCHAIN_F64;
    for (int k = 0; k < K; k++) {
      accepted[k] = 0;
      if (mcin.burnin > 0) memcpy(cur + (long)k * d, burn + (long)k * bl * d, sizeof(double) * d);
    }
    if (mcin.burnin > 0) /* the chain start, burn_in_metropolis_gpu (mcmc_gpu.cu:139-146) */
      MH_CHECK(mhlik_kchain_begin(lik, mode, K, cur, PRIOR_SD, NULL, cpost), lik);
    else
      MH_CHECK(mhlik_kchain_resume(lik, mode, K, cur, cpost, PRIOR_SD), lik);
    if (nburn + mcin.Ns > 0)
      kc_run(lik, K, d, r, mct.rwsd, nburn, mcin.Ns, bl, samples, burn, accepted, NULL, 0.0, 0, NULL, 0, NULL, NULL);
    free(cur);
    free(cpost);
    return;
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

  const long total_it = (mcin.burnin > 0 ? mcin.burnin - 1 : 0) + mcin.Ns;
  if (mh_device_chain_enabled() && mcin.burnin > 0 && total_it > 0) { /* the lock-step loop on the device */
    const long bl = mcin.burnin;
    double *cur0 = (double *)calloc((size_t)K * d, sizeof(double));
    int32_t *choose0 = (int32_t *)calloc((size_t)K, sizeof(int32_t));
    uint64_t *seeds = (uint64_t *)calloc((size_t)K, sizeof(uint64_t));
    int32_t *nb = (int32_t *)calloc((size_t)K, sizeof(int32_t));
    int32_t *nd = (int32_t *)calloc((size_t)K, sizeof(int32_t));
    for (int k = 0; k < K; k++) {
      seeds[k] = MP_U_SEED;
      accepted[k] = 0;
    }
    for (long i = 0; i < (long)K * d; i++) cur0[i] = theta0 ? theta0[i % d] : 1; /* :433 starts every chain at 1 */
    /* start of burn_in_metropolis_mp, :430-440: the first evaluation draws its phase like every other (:504) */
    for (int k = 0; k < K; k++) choose0[k] = (int32_t)gsl_rng_uniform_int(r[k], (unsigned long)mpVar->ResampFactor);
    if (shared_phase)
      for (int k = 1; k < K; k++) choose0[k] = choose0[0];
    mhlik_kchain_mp first;
    memset(&first, 0, sizeof first);
    first.resamp_factor = mpVar->ResampFactor;
    first.resamp_choose = choose0;
    first.u_seed = seeds;
    first.u_call = 0;
    first.bound_eps = lhoodBound;
    MH_CHECK(mhlik_kchain_begin(lik, MHLIK_KCHAIN_MP, K, cur0, MP_PRIOR_SD, &first, NULL), lik);
    kc_run(lik, K, d, r, mct.rwsd, mcin.burnin - 1, mcin.Ns, bl, samples, burn, accepted, mpVar, lhoodBound, shared_phase,
           seeds, 1, nb, nd);
    long nb_sum = 0;
    for (int k = 0; k < K; k++) {
      if (bright_blocks) bright_blocks[k] = nb[k];
      nb_sum += nb[k];
    }
    mpVar->brightBlocks = (int)(nb_sum / K); /* mean over the chains */
    mpVar->darkBlocks = mpVar->dataBlocks - mpVar->brightBlocks;
    mpVar->ratio = ((float)mpVar->brightBlocks) / mpVar->dataBlocks;
    free(cur0);
    free(choose0);
    free(seeds);
    free(nb);
    free(nd);
    return;
  }
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
