/*
 * tests/doubles/kc_pipeline_main.c -- drives samplers/multichain.c's K-chain
 * chunk pipeline against tests/doubles/mhlik_double.c, for the thread and
 * address sanitizers: two threads call the samplers at once (one takes the
 * cached buffers, the other its own), shapes change between rounds (the cache is
 * released and rebuilt), and every output row is checked against a replay of the
 * same random streams.  Prints "kc pipeline ok" and exits 0.
 */
#include "mcmc_samplers.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int K, d, Ns, burnin, mp, seed0, fails;
} job;

static void *run(void *arg)
{
  job *j = (job *)arg;
  const int K = j->K, d = j->d;
  mhlik_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_rows = 1000;
  cfg.n_cols = d;
  mhlik_ctx *lik = NULL;
  mhlik_create(&cfg, &lik);
  gsl_rng **r = (gsl_rng **)malloc(sizeof(gsl_rng *) * K), **r2 = (gsl_rng **)malloc(sizeof(gsl_rng *) * K);
  for (int k = 0; k < K; k++) {
    r[k] = gsl_rng_alloc(gsl_rng_mt19937);
    gsl_rng_set(r[k], (unsigned long)(j->seed0 + k));
    r2[k] = gsl_rng_alloc(gsl_rng_mt19937);
    gsl_rng_set(r2[k], (unsigned long)(j->seed0 + k));
  }
  mcmc_str mcin;
  memset(&mcin, 0, sizeof mcin);
  mcin.ddata = d;
  mcin.Nd = 1000;
  mcin.Ns = j->Ns;
  mcin.burnin = j->burnin;
  mcmc_tune_str mct;
  memset(&mct, 0, sizeof mct);
  mct.rwsd = 0.1;
  double *samples = (double *)calloc((size_t)K * j->Ns * d, sizeof(double));
  double *burn = (double *)calloc((size_t)K * j->burnin * d, sizeof(double));
  int *accepted = (int *)calloc((size_t)K, sizeof(int));
  int *bright = (int *)calloc((size_t)K, sizeof(int));
  mp_str mp;
  memset(&mp, 0, sizeof mp);
  mp.dataBlocks = 16;
  mp.dataBlockSz = 64;
  mp.ResampFactor = 4;
  if (j->mp)
    mcmc_amd_mp_multichain_sampler(lik, K, r, &mp, mcin, mct, samples, burn, accepted, 1e-7, 0, bright, NULL);
  else
    mcmc_amd_multichain_sampler(lik, K, r, mcin, mct, samples, burn, accepted, 0);
  /* replay: chain k from generator k, in the order the sampler draws (d Gaussians, [phase,] one uniform) */
  for (int k = 0; k < K; k++) {
    double *cur = (double *)calloc((size_t)d, sizeof(double));
    if (j->mp) {
      for (int c = 0; c < d; c++) cur[c] = 1.0; /* mp_sampler starts at theta = 1 (mcmc_mp.cu:433) */
      (void)gsl_rng_uniform_int(r2[k], 4);      /* the chain start's own phase (mcmc_mp.cu:504) */
    }
    int acc = 0;
    for (long it = 0; it < (long)j->burnin - 1 + j->Ns; it++) {
      double inc[64];
      for (int c = 0; c < d; c++) inc[c] = gsl_ran_gaussian_ziggurat(r2[k], mct.rwsd);
      if (j->mp) (void)gsl_rng_uniform_int(r2[k], 4);
      const double u = gsl_rng_uniform(r2[k]);
      const int a = j->mp ? u < 0.3 : log(u) < -1.0;
      if (a)
        for (int c = 0; c < d; c++) cur[c] += inc[c];
      const double *row = it < j->burnin - 1 ? burn + ((size_t)k * j->burnin + it + 1) * d
                                              : samples + ((size_t)k * j->Ns + (it - (j->burnin - 1))) * d;
      if (memcmp(row, cur, sizeof(double) * d) != 0) j->fails++;
      if (it >= j->burnin - 1) acc += a;
    }
    if (acc != accepted[k]) j->fails++;
    free(cur);
  }
  for (int k = 0; k < K; k++) {
    gsl_rng_free(r[k]);
    gsl_rng_free(r2[k]);
  }
  free(r); free(r2); free(samples); free(burn); free(accepted); free(bright);
  mhlik_destroy(lik);
  return NULL;
}

int main(void)
{
  mcmc_amd_set_verbose(0);
  int fails = 0;
  for (int round = 0; round < 3; round++) {
    job a = {7 + round, 5, 700, 40, 0, 100 * round, 0}, b = {3, 9, 333, 2, round == 1, 1000 + round, 0};
    pthread_t ta, tb;
    pthread_create(&ta, NULL, run, &a);
    pthread_create(&tb, NULL, run, &b);
    pthread_join(ta, NULL);
    pthread_join(tb, NULL);
    fails += a.fails + b.fails;
    if (a.fails + b.fails) printf("round %d: %d + %d mismatches\n", round, a.fails, b.fails);
  }
  job c = {4, 3, 20, 1, 1, 5, 0};
  run(&c);
  fails += c.fails;
  if (fails) {
    printf("kc pipeline: %d mismatches\n", fails);
    return 1;
  }
  printf("kc pipeline ok\n");
  return 0;
}
