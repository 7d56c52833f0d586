/*
 * compare_host.c -- the two comparison harnesses BASELINE.json names, one source:
 *   -DCOMPARE_IMPL=1  bin/compare_gpu_sp   (tests/compare_gpu_sp.c:14-143 of the reference: tune + sp_sampler, then
 *                                           tune + gpu_sampler on the same data, one CSV row of times + ESS each)
 *   -DCOMPARE_IMPL=2  bin/compare_cpu_gpu  (tests/compare_cpu_gpu.c:14-123: tune on the GPU path, cpu_sampler, then
 *                                           gpu_sampler)
 * Same call sequence, same CSV row ("Dim, Nd, sampler_t(ms), metrop_t(ms), burn_t(ms), ess, ess/t(s)" with %.32f
 * fields) appended to the two output files.  The reference picks its paths from -d (data/compare_gpu_sp/synthetic.csv,
 * out/compare_gpu_sp/sp_200.csv, ...); here -in <csv> and -out <prefix> override them (rows go to <prefix>sp.csv /
 * <prefix>gpu.csv / <prefix>cpu.csv).  Times are wall time (the reference's clock() counts CPU time only).
 *
 * Like the reference, compare_cpu_gpu lets the second sampler CONTINUE the first one's random stream
 * (compare_cpu_gpu.c:96-105), so its two chains see different proposals; -reseed 1 (not in the reference) seeds the
 * generator again before the second run, which makes the two chains comparable sample for sample -- both files then
 * hold the same ESS, and the run ends non-zero if the two sample arrays differ.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mcmc_host_util.h"
#include "mcmc_samplers.h"

#ifndef COMPARE_IMPL
#define COMPARE_IMPL 1
#endif

static double now_ms(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static void row(FILE *fp, mcmc_str mcin, out_str res)
{
  fprintf(fp, "%d, %d, %.32f, %.32f, %.32f, %.32f, %.32f\n", mcin.ddata, mcin.Nd, res.samplerTime, res.mcmcTime,
          res.burnTime, res.ess, res.ess / (res.samplerTime * 1000));
}

int main(int argc, char *argv[])
{
  data_str data;
  mcmc_str mcin;
  mcmc_tune_str mct;
  mcmc_v_str mcdata;
  sec_str sec;
  sec_v_str secv;
  gpu_v_str gpu;
  out_str res;
  gsl_rng *r = NULL;
  char indir[4096], out_a[4096], out_b[4096];
  memset(&data, 0, sizeof data);
  memset(&mcin, 0, sizeof mcin);
  memset(&mct, 0, sizeof mct);
  memset(&mcdata, 0, sizeof mcdata);
  memset(&sec, 0, sizeof sec);
  memset(&secv, 0, sizeof secv);
  memset(&gpu, 0, sizeof gpu);
  memset(&res, 0, sizeof res);

  /* -reseed 1 is ours; take it out of argv before the reference's parser sees it */
  int reseed = 0, an = 0;
  char **av = (char **)calloc((size_t)argc + 1, sizeof(char *));
  for (int i = 0; i < argc; i++) {
    if (strcmp(argv[i], "-reseed") == 0 && i + 1 < argc) {
      reseed = atoi(argv[++i]);
      continue;
    }
    av[an++] = argv[i];
  }
  init_rng(&r);
  read_inputs_gpu(an, av, &mcin, &sec, &gpu);
  mcin.impl = CPU; /* both reference harnesses leave it at CPU (compare_gpu_sp.c:34) */
  if (mcin.Nd <= 0 || mcin.ddata <= 0 || mcin.Ns <= 0 || mcin.burnin < 0) {
    fprintf(stderr, "usage: %s -d 3 -sz N -dim d -samp Ns -burn B -lag L -tune 0|1|2 [-in data.csv] [-out prefix] [-reseed 1]\n",
            argv[0]);
    return 1;
  }
  const char *first = COMPARE_IMPL == 1 ? "sp" : "cpu";
  snprintf(indir, sizeof indir, "data/%s/synthetic.csv", COMPARE_IMPL == 1 ? "compare_gpu_sp" : "compare_gpu_cpu");
  snprintf(out_a, sizeof out_a, "out/%s/%s_%d.csv", COMPARE_IMPL == 1 ? "compare_gpu_sp" : "compare_gpu_cpu", first, mcin.ddata);
  snprintf(out_b, sizeof out_b, "out/%s/gpu_%d.csv", COMPARE_IMPL == 1 ? "compare_gpu_sp" : "compare_gpu_cpu", mcin.ddata);
  if (mcmc_amd_cli_input()) snprintf(indir, sizeof indir, "%s", mcmc_amd_cli_input());
  if (mcmc_amd_cli_output()) {
    snprintf(out_a, sizeof out_a, "%s%s.csv", mcmc_amd_cli_output(), first);
    snprintf(out_b, sizeof out_b, "%sgpu.csv", mcmc_amd_cli_output());
  }

  malloc_data_vectors(&data, mcin);
  malloc_sample_vectors(&mcdata, mcin);
  malloc_autocorrelation_vectors(&secv, sec);
  read_data(indir, ColMajor, data, mcin);
#if COMPARE_IMPL == 1
  malloc_data_vectors_sp(&data, mcin);
  malloc_sample_vectors_sp(&mcdata, mcin);
  malloc_autocorrelation_vectors_sp(&secv, sec);
  read_data_sp(indir, ColMajor, data, mcin);
#endif
  FILE *fa = fopen(out_a, "a"), *fb = fopen(out_b, "a");
  if (!fa || !fb) {
    fprintf(stderr, "ERROR: cannot open %s / %s for appending.\n", out_a, out_b);
    return 1;
  }
  const unsigned long seed = gsl_rng_default_seed;
  double t0;
  int differ = 0;

#if COMPARE_IMPL == 1
  mct.rwsdf = 2.38 / (float)(sqrt(mcin.ddata)); /* compare_gpu_sp.c:95-101 */
  for (int i = 0; i < mcin.ddata; i++) mcdata.burnf[i] = 0;
  if (mcin.tune == 1)
    tune_target_a_sp_v2(data, r, mcin, &mct, gpu, mcdata.burnf, 0.25, 100);
  else if (mcin.tune == 2)
    tune_ess_sp(data, r, mcin, &mct, gpu, mcdata.burnf, 5000);
  printf("Starting run for Nd:%d, dim:%d\n", mcin.Nd, mcin.ddata);
  t0 = now_ms();
  sp_sampler(data, r, mcin, mct, mcdata, gpu, &res);
  res.samplerTime = now_ms() - t0;
  printf("Run for Nd:%d, dim:%d Completed..\n", mcin.Nd, mcin.ddata);
  res.ess = get_ess_sp(mcdata.samplesf, mcin.Ns, mcin.ddata, sec.lagidx, secv.circf);
  row(fa, mcin, res);
  printf("sp : sampler %.3f ms, metropolis %.3f ms, burn-in %.3f ms, ess %.3f, kernel %.4f ms (%.1f GB/s)\n", res.samplerTime,
         res.mcmcTime, res.burnTime, res.ess, res.kernelTime, res.kernelBandwidth);
  if (reseed) gsl_rng_set(r, seed);

  mct.rwsd = 2.38 / sqrt(mcin.ddata); /* :110-116 */
  for (int i = 0; i < mcin.ddata; i++) mcdata.burn[i] = 0;
  if (mcin.tune == 1)
    tune_target_a_gpu_v2(data, r, mcin, &mct, gpu, mcdata.burn, 0.25, 100);
  else if (mcin.tune == 2)
    tune_ess_gpu(data, r, mcin, &mct, gpu, mcdata.burn, 5000);
#else
  mct.rwsd = 2.38 / sqrt(mcin.ddata); /* compare_cpu_gpu.c:86-92 */
  for (int i = 0; i < mcin.ddata; i++) mcdata.burn[i] = 0;
  if (mcin.tune == 1)
    tune_target_a_gpu_v2(data, r, mcin, &mct, gpu, mcdata.burn, 0.25, 100);
  else if (mcin.tune == 2)
    tune_ess_gpu(data, r, mcin, &mct, gpu, mcdata.burn, 5000);
  if (reseed) gsl_rng_set(r, seed);
  printf("Starting run for Nd:%d, dim:%d\n", mcin.Nd, mcin.ddata);
  t0 = now_ms();
  cpu_sampler(data, r, mcin, mct, mcdata, &res);
  res.samplerTime = now_ms() - t0;
  printf("Run for Nd:%d, dim:%d Completed..\n", mcin.Nd, mcin.ddata);
  res.ess = get_ess(mcdata.samples, mcin.Ns, mcin.ddata, sec.lagidx, secv.circ);
  row(fa, mcin, res);
  printf("cpu: sampler %.3f ms, metropolis %.3f ms, burn-in %.3f ms, ess %.3f\n", res.samplerTime, res.mcmcTime,
         res.burnTime, res.ess);
  double *first_samples = NULL;
  if (reseed) {
    first_samples = (double *)malloc(sizeof(double) * (size_t)mcin.Ns * mcin.ddata);
    memcpy(first_samples, mcdata.samples, sizeof(double) * (size_t)mcin.Ns * mcin.ddata);
    gsl_rng_set(r, seed);
    for (int i = 0; i < mcin.ddata; i++) mcdata.burn[i] = 0;
  }
#endif
  printf("Starting run for Nd:%d, dim:%d\n", mcin.Nd, mcin.ddata);
  t0 = now_ms();
  gpu_sampler(data, r, mcin, mct, mcdata, gpu, &res);
  res.samplerTime = now_ms() - t0;
  printf("Run for Nd:%d, dim:%d Completed..\n", mcin.Nd, mcin.ddata);
  res.ess = get_ess(mcdata.samples, mcin.Ns, mcin.ddata, sec.lagidx, secv.circ);
  row(fb, mcin, res);
  printf("gpu: sampler %.3f ms, metropolis %.3f ms, burn-in %.3f ms, ess %.3f, kernel %.4f ms (%.1f GB/s)\n", res.samplerTime,
         res.mcmcTime, res.burnTime, res.ess, res.kernelTime, res.kernelBandwidth);
#if COMPARE_IMPL == 2
  if (first_samples) {
    differ = memcmp(first_samples, mcdata.samples, sizeof(double) * (size_t)mcin.Ns * mcin.ddata) != 0;
    printf("re-seeded: the two chains are %s\n", differ ? "DIFFERENT" : "identical sample for sample");
    free(first_samples);
  }
#endif
  fclose(fa);
  fclose(fb);

  free_autocorrelation_vectors(secv);
  free_data_vectors(data, mcin);
  free_sample_vectors(mcdata);
#if COMPARE_IMPL == 1
  free_autocorrelation_vectors_sp(secv);
  free_data_vectors_sp(data, mcin);
  free_sample_vectors_sp(mcdata);
#endif
  free_rng(r);
  free(av);
  return differ ? 2 : 0;
}
