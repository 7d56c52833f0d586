/*
 * mcmc_host.c -- the four drivers, one source:
 *   -DHOST_IMPL=4  bin/mcmc_cpu   (src/mcmc_cpu/cpu_host.c:13-86; the CPU path's entry points, run on the GPU)
 *   -DHOST_IMPL=1  bin/mcmc_gpu   (src/mcmc_gpu/gpu_host.c:13-85)
 *   -DHOST_IMPL=2  bin/mcmc_sp    (src/mcmc_gpu_sp/gpu_sp_host.c)
 *   -DHOST_IMPL=3  bin/mcmc_mp    (src/mcmc_mp/mp_host.c:7-80)
 * Same flow as the reference's mains: parse flags, read the CSV column-major,
 * rwsd = 2.38/sqrt(d), tune, sample, ESS, print, write CSVs.  The reference
 * hard-codes its input / output paths from -d; here -in / -out override them.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mcmc_host_util.h"
#include "mcmc_samplers.h"

/* numeric on purpose: the enum names of mcmc_types.h are not visible to #if */
#define HOST_GPU 1
#define HOST_SP 2
#define HOST_MP 3
#define HOST_CPU 4
#ifndef HOST_IMPL
#define HOST_IMPL HOST_GPU
#endif

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
  char indir[4096], outdir[4096];
  memset(&data, 0, sizeof data);
  memset(&mcin, 0, sizeof mcin);
  memset(&mct, 0, sizeof mct);
  memset(&mcdata, 0, sizeof mcdata);
  memset(&sec, 0, sizeof sec);
  memset(&secv, 0, sizeof secv);
  memset(&gpu, 0, sizeof gpu);
  memset(&res, 0, sizeof res);

#if HOST_IMPL == HOST_CPU
  read_inputs(argc, argv, &mcin, &sec); /* cpu_host.c:32: no launch-geometry flags */
#else
  read_inputs_gpu(argc, argv, &mcin, &sec, &gpu);
#endif
  mcin.impl = HOST_IMPL == HOST_CPU ? CPU : HOST_IMPL == HOST_GPU ? GPU : HOST_IMPL == HOST_SP ? SP : MP;
  if (mcin.Nd <= 0 || mcin.ddata <= 0 || mcin.Ns <= 0 || mcin.burnin < 0) {
    fprintf(stderr, "usage: %s -d 1|2 -sz N -dim d -samp Ns -burn B -lag L -tune 0|1|2 [-in data.csv] [-out prefix]\n",
            argv[0]);
    return 1;
  }
  const char *tag = HOST_IMPL == HOST_CPU ? "cpu_" : HOST_IMPL == HOST_GPU ? "gpu_" : HOST_IMPL == HOST_SP ? "sp_" : "mp_";
  snprintf(indir, sizeof indir, "%s", sec.fdata == 2 ? "data/host/mnist.csv" : "data/host/synthetic.csv");
  snprintf(outdir, sizeof outdir, "out/host/%s/%s", sec.fdata == 2 ? "mnist" : "synthetic", tag);
  if (mcmc_amd_cli_input()) snprintf(indir, sizeof indir, "%s", mcmc_amd_cli_input());
  if (mcmc_amd_cli_output()) snprintf(outdir, sizeof outdir, "%s", mcmc_amd_cli_output());

  init_rng(&r);
  clock_t t0;

#if HOST_IMPL == HOST_SP
  malloc_data_vectors_sp(&data, mcin);
  malloc_sample_vectors_sp(&mcdata, mcin);
  malloc_autocorrelation_vectors_sp(&secv, sec);
  read_data_sp(indir, ColMajor, data, mcin);
  mct.rwsdf = 2.38 / sqrt(mcin.ddata);
  for (int i = 0; i < mcin.ddata; i++) mcdata.burnf[i] = 0;
  t0 = clock();
  if (mcin.tune == 1)
    tune_target_a_sp_v2(data, r, mcin, &mct, gpu, mcdata.burnf, 0.25, 40);
  else if (mcin.tune == 2)
    tune_ess_sp(data, r, mcin, &mct, gpu, mcdata.burnf, 5000);
  res.tuneTime = (clock() - t0) * 1000 / CLOCKS_PER_SEC;
  t0 = clock();
  sp_sampler(data, r, mcin, mct, mcdata, gpu, &res);
  res.samplerTime = (clock() - t0) * 1000 / CLOCKS_PER_SEC;
  res.ess = get_ess_sp(mcdata.samplesf, mcin.Ns, mcin.ddata, sec.lagidx, secv.circf);
  write_bandwidth_test_out(res);
  calculate_normalised_sample_means_sp(mcdata, mcin);
  print_normalised_sample_means_sp(mcdata, mcin);
  write_csv_outputs_sp(outdir, mcdata, mcin, sec, secv);
  free_autocorrelation_vectors_sp(secv);
  free_data_vectors_sp(data, mcin);
  free_sample_vectors_sp(mcdata);
#elif HOST_IMPL == HOST_MP
  mp_str mpVar;
  memset(&mpVar, 0, sizeof mpVar);
  malloc_data_vectors_mp(&data, mcin);
  malloc_sample_vectors(&mcdata, mcin);
  malloc_autocorrelation_vectors(&secv, sec);
  read_data(indir, ColMajor, data, mcin);
  mct.rwsd = 2.313297; /* mp_host.c:49 */
  for (int i = 0; i < mcin.ddata; i++) mcdata.burn[i] = 0;
  const double lowerBound = 0.00000017332527801894315; /* mp_host.c:55 */
  mpVar.dataBlockSz = 64;                               /* mp_host.c:57-59 */
  mpVar.dataBlocks = 16;
  mpVar.ResampFactor = 2;
  t0 = clock();
  mp_sampler(data, r, &mpVar, mcin, mct, mcdata, lowerBound, &res);
  res.samplerTime = (clock() - t0) * 1000 / CLOCKS_PER_SEC;
  res.ess = get_ess(mcdata.samples, mcin.Ns, mcin.ddata, sec.lagidx, secv.circ);
  write_bandwidth_test_out(res);
  calculate_normalised_sample_means(mcdata, mcin);
  print_normalised_sample_means(mcdata, mcin);
  write_csv_outputs(outdir, mcdata, mcin, sec, secv);
  free_autocorrelation_vectors(secv);
  free_data_vectors(data, mcin);
  free_sample_vectors(mcdata);
#elif HOST_IMPL == HOST_CPU
  malloc_data_vectors(&data, mcin);
  malloc_sample_vectors(&mcdata, mcin);
  malloc_autocorrelation_vectors(&secv, sec);
  read_data(indir, ColMajor, data, mcin);
  mct.rwsd = 2.38 / sqrt(mcin.ddata); /* cpu_host.c:49 */
  for (int i = 0; i < mcin.ddata; i++) mcdata.burn[i] = 0;
  t0 = clock();
  if (mcin.tune == 1)
    tune_target_a_cpu_v2(data, r, mcin, &mct, mcdata.burn, 0.25, 40);
  else if (mcin.tune == 2)
    tune_ess_cpu(data, r, mcin, &mct, mcdata.burn, 5000);
  res.tuneTime = (clock() - t0) * 1000 / CLOCKS_PER_SEC;
  t0 = clock();
  cpu_sampler(data, r, mcin, mct, mcdata, &res);
  res.samplerTime = (clock() - t0) * 1000 / CLOCKS_PER_SEC;
  res.ess = get_ess(mcdata.samples, mcin.Ns, mcin.ddata, sec.lagidx, secv.circ);
  write_test_out(res); /* cpu_host.c:71 */
  calculate_normalised_sample_means(mcdata, mcin);
  print_normalised_sample_means(mcdata, mcin);
  write_csv_outputs(outdir, mcdata, mcin, sec, secv);
  free_autocorrelation_vectors(secv);
  free_data_vectors(data, mcin);
  free_sample_vectors(mcdata);
#else
  malloc_data_vectors(&data, mcin);
  malloc_sample_vectors(&mcdata, mcin);
  malloc_autocorrelation_vectors(&secv, sec);
  read_data(indir, ColMajor, data, mcin);
  mct.rwsd = 2.38 / sqrt(mcin.ddata);
  for (int i = 0; i < mcin.ddata; i++) mcdata.burn[i] = 0;
  t0 = clock();
  if (mcin.tune == 1)
    tune_target_a_gpu_v2(data, r, mcin, &mct, gpu, mcdata.burn, 0.25, 40);
  else if (mcin.tune == 2)
    tune_ess_gpu(data, r, mcin, &mct, gpu, mcdata.burn, 5000);
  res.tuneTime = (clock() - t0) * 1000 / CLOCKS_PER_SEC;
  t0 = clock();
  gpu_sampler(data, r, mcin, mct, mcdata, gpu, &res);
  res.samplerTime = (clock() - t0) * 1000 / CLOCKS_PER_SEC;
  res.ess = get_ess(mcdata.samples, mcin.Ns, mcin.ddata, sec.lagidx, secv.circ);
  write_bandwidth_test_out(res);
  calculate_normalised_sample_means(mcdata, mcin);
  print_normalised_sample_means(mcdata, mcin);
  write_csv_outputs(outdir, mcdata, mcin, sec, secv);
  free_autocorrelation_vectors(secv);
  free_data_vectors(data, mcin);
  free_sample_vectors(mcdata);
#endif
  free_rng(r);
  return 0;
}
