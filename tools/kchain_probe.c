/* kchain_probe.c -- the K-chain C samplers (device-resident lock-step loop: double, float, mixed precision with one
 * shared phase and with a phase per chain) on a device-filled context, through libmcmc_amd.so + libmhlik.so, with a
 * checksum per run.  No Python in the process: this is what tools/r03_host_asan.sh runs with libmhlik.so's host code
 * under ASan + UBSan.  Run twice (MCMC_AMD_HOST_LOOP=0 / 1) the checksums must agree.
 * gcc tools/kchain_probe.c -Iinclude -Imcmc_amd/csrc/gslcompat -Lmcmc_amd/lib -lmcmc_amd -lmhlik -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mcmc_samplers.h"
#include "mhlik.h"

static unsigned long long checksum(const void *p, size_t bytes)
{
  const unsigned char *b = p;
  unsigned long long h = 1469598103934665603ull;
  for (size_t i = 0; i < bytes; i++) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

int main(int argc, char **argv)
{
  const long N = argc > 1 ? atol(argv[1]) : 40000;
  const int d = argc > 2 ? atoi(argv[2]) : 6;
  const int K = argc > 3 ? atoi(argv[3]) : 16;
  const int Ns = 600, burnin = 100;
  mhlik_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_rows = N; cfg.n_cols = d; cfg.precision = MHLIK_F64 | MHLIK_F32; cfg.world = 1; cfg.segments = 1; cfg.timing = 1;
  mhlik_ctx *ctx;
  if (mhlik_create(&cfg, &ctx)) { printf("create: %s\n", mhlik_last_error(NULL)); return 1; }
  if (mhlik_fill_synthetic(ctx, 4242)) { printf("fill: %s\n", mhlik_last_error(ctx)); return 1; }
  gsl_rng **r = malloc(sizeof(*r) * K);
  double *samples = malloc(sizeof(double) * (size_t)K * Ns * d), *burn = malloc(sizeof(double) * (size_t)K * burnin * d);
  int *accepted = malloc(sizeof(int) * K), *bright = malloc(sizeof(int) * K);
  mcmc_str mcin;
  memset(&mcin, 0, sizeof mcin);
  mcin.ddata = d; mcin.Nd = (int)N; mcin.Ns = Ns; mcin.burnin = burnin;
  mcmc_tune_str mct = {2.38 / sqrt(d) * 0.02, (float)(2.38 / sqrt(d) * 0.02)};
  for (int mode = 0; mode < 4; mode++) {
    for (int k = 0; k < K; k++) init_rng_seed(&r[k], 100 + k);
    memset(samples, 0, sizeof(double) * (size_t)K * Ns * d);
    memset(burn, 0, sizeof(double) * (size_t)K * burnin * d);
    if (mode < 2) {
      mcmc_amd_multichain_sampler(ctx, K, r, mcin, mct, samples, burn, accepted, mode);
    } else {
      mp_str mp;
      memset(&mp, 0, sizeof mp);
      mp.dataBlockSz = 64; mp.dataBlocks = (int)(N / 64); mp.ResampFactor = 4;
      double *theta0 = calloc(d, sizeof(double));
      mcmc_amd_mp_multichain_sampler(ctx, K, r, &mp, mcin, mct, samples, burn, accepted, 1.7e-7, mode == 2, bright, theta0);
      free(theta0);
    }
    long acc = 0;
    for (int k = 0; k < K; k++) acc += accepted[k];
    printf("%s: accepted %ld of %d, samples %016llx burn %016llx\n",
           mode == 0 ? "double" : mode == 1 ? "float" : mode == 2 ? "mixed precision, shared phase" : "mixed precision, phase per chain",
           acc, K * Ns, checksum(samples, sizeof(double) * (size_t)K * Ns * d), checksum(burn, sizeof(double) * (size_t)K * burnin * d));
    for (int k = 0; k < K; k++) free_rng(r[k]);
  }
  mhlik_destroy(ctx);
  return 0;
}
