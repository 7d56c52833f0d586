/* chain_probe.c -- a few iterations of the device-resident chain through the C-ABI, printed (development probe:
 * compares the one-launch and the two-kernel path under different builds of libmhlik.so).
 * gcc tools/chain_probe.c -Iinclude -Lmcmc_amd/lib -lmhlik -Wl,-rpath,$PWD/mcmc_amd/lib -lm -o tools/chain_probe */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mhlik.h"

int main(int argc, char **argv)
{
  const long N = argc > 1 ? atol(argv[1]) : 5000;
  const int d = argc > 2 ? atoi(argv[2]) : 3;
  const int n = argc > 3 ? atoi(argv[3]) : 12;
  mhlik_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.n_rows = N; cfg.n_cols = d; cfg.precision = MHLIK_F64; cfg.world = 1; cfg.timing = 1;
  mhlik_ctx *ctx;
  if (mhlik_create(&cfg, &ctx)) { printf("create: %s\n", mhlik_last_error(NULL)); return 1; }
  if (mhlik_fill_synthetic(ctx, 12345)) { printf("fill: %s\n", mhlik_last_error(ctx)); return 1; }
  double *theta0 = calloc(d, sizeof(double)), cpost = 0;
  if (mhlik_chain_begin_f64(ctx, theta0, 10.0, &cpost)) { printf("begin: %s\n", mhlik_last_error(ctx)); return 1; }
  printf("cposterior %.17g\n", cpost);
  double *inc = malloc(sizeof(double) * n * d), *logu = malloc(sizeof(double) * n), *samples = malloc(sizeof(double) * n * d);
  double *lik = malloc(sizeof(double) * n), *ratio = malloc(sizeof(double) * n);
  uint8_t *acc = malloc(n);
  unsigned s = 99;
  for (int i = 0; i < n * d; i++) { s = s * 1664525u + 1013904223u; inc[i] = ((s >> 8) / 16777216.0 - 0.5) * 0.02; }
  for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; logu[i] = log((s >> 8) / 16777216.0 + 1e-9); }
  mhlik_chain_batch_f64 b;
  memset(&b, 0, sizeof b);
  b.n = n; b.increments = inc; b.log_u = logu; b.samples = samples; b.lik = lik; b.ratio = ratio; b.accepted = acc;
  if (mhlik_chain_run_f64(ctx, &b)) { printf("run: %s\n", mhlik_last_error(ctx)); return 1; }
  for (int i = 0; i < n; i++) printf("%2d lik %.17g ratio %.17g logu %.6g acc %d  s0 %.17g\n", i, lik[i], ratio[i], logu[i], acc[i], samples[i * d]);
  printf("accepted %lld kernel_ms %g\n", (long long)b.n_accepted, b.kernel_ms);
  mhlik_destroy(ctx);
  return 0;
}
