/*
 * host_util.c -- allocation, command line, result printing and CSV in / out
 * for the drivers (see include/mcmc_host_util.h for the reference citations).
 *
 * Kept from the reference: flag names and meaning (src/util/io.c:6-176), the
 * label folding X[i,j] = -y_i * x_ij and the two storage orders
 * (src/util/data_util.c:6-45), every output format string (io.c:178-235,
 * data_util.c:166-309).
 * Not kept: an unknown flag loops forever there (no else-branch, io.c:9-70)
 * -- here it is an error; fields of mcmc_str / sec_str / gpu_v_str start
 * zeroed instead of uninitialised; indices are 64-bit (data_util.c:33
 * multiplies two ints); file names are built in buffers that cannot overflow
 * (data_util.c:86-112 strcat into char[50]).
 */
#include "mcmc_host_util.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ alloc */

static void *need(size_t n, size_t sz, const char *what)
{
  void *p = calloc(n ? n : 1, sz);
  if (!p) fprintf(stderr, "ERROR: %s Memory allocation did not complete successfully!\n", what);
  return p;
}

void malloc_data_vectors(data_str *data, mcmc_str mcin)
{
  data->data = (double *)need((size_t)mcin.ddata * mcin.Nd, sizeof(double), "Data");
  data->mvout = (double *)need((size_t)mcin.Nd, sizeof(double), "MVout Data");
}

void malloc_data_vectors_sp(data_str *data, mcmc_str mcin)
{
  data->dataf = (float *)need((size_t)mcin.ddata * mcin.Nd, sizeof(float), "Dataf");
  data->mvoutf = (float *)need((size_t)mcin.Nd, sizeof(float), "MVoutf Data");
}

void malloc_data_vectors_mp(data_str *data, mcmc_str mcin)
{
  data->data = (double *)need((size_t)mcin.ddata * mcin.Nd, sizeof(double), "Data");
  data->dataf = (float *)need((size_t)mcin.ddata * mcin.Nd, sizeof(float), "Single Precision Data");
}

void malloc_sample_vectors(mcmc_v_str *m, mcmc_str mcin)
{
  const size_t d = (size_t)mcin.ddata;
  m->samples = (double *)need(d * mcin.Ns, sizeof(double), "Parameters Data");
  m->burn = (double *)need(d * mcin.burnin, sizeof(double), "Burn Parameters Data");
  m->nsamples = (double *)need(d * mcin.Ns, sizeof(double), "Normalised Parameters Data");
  m->nburn = (double *)need(d * mcin.burnin, sizeof(double), "Normalised Burned Parameters Data");
  m->sample_means = (double *)need(d, sizeof(double), "Sample Means");
}

void malloc_sample_vectors_sp(mcmc_v_str *m, mcmc_str mcin)
{
  const size_t d = (size_t)mcin.ddata;
  m->samplesf = (float *)need(d * mcin.Ns, sizeof(float), "Parameters Data");
  m->burnf = (float *)need(d * mcin.burnin, sizeof(float), "Burn Parameters Data");
  m->nsamplesf = (float *)need(d * mcin.Ns, sizeof(float), "Normalised Parameters Data");
  m->nburnf = (float *)need(d * mcin.burnin, sizeof(float), "Normalised Burned Parameters Data");
  m->sample_meansf = (float *)need(d, sizeof(float), "Sample Means");
}

void malloc_autocorrelation_vectors(sec_v_str *secv, sec_str sec)
{
  secv->shift = (double *)need((size_t)sec.lagidx, sizeof(double), "Shift Autocorrelation Data");
  secv->circ = (double *)need((size_t)sec.lagidx, sizeof(double), "Circular Autocorrelation Data");
}

void malloc_autocorrelation_vectors_sp(sec_v_str *secv, sec_str sec)
{
  secv->circf = (float *)need((size_t)sec.lagidx, sizeof(float), "Circular Autocorrelation Data");
}

void malloc_mcmc_vectors(mcmc_int_v *v, mcmc_str mcin)
{
  v->proposed = (double *)need((size_t)mcin.ddata, sizeof(double), "Proposed Samples");
  v->current = (double *)need((size_t)mcin.ddata, sizeof(double), "Current Samples");
  if (mcin.impl == MP) {
    v->proposedf = (float *)need((size_t)mcin.ddata, sizeof(float), "Single Precision Proposed Samples");
    v->currentf = (float *)need((size_t)mcin.ddata, sizeof(float), "Single Precision Current Samples");
  }
}

void malloc_mcmc_vectors_sp(mcmc_int_v *v, mcmc_str mcin)
{
  v->proposedf = (float *)need((size_t)mcin.ddata, sizeof(float), "Proposedf Samples");
  v->currentf = (float *)need((size_t)mcin.ddata, sizeof(float), "Currentf Samples");
}

void free_data_vectors_sp(data_str data, mcmc_str mcin)
{
  (void)mcin;
  free(data.dataf);
  free(data.mvoutf);
}

void free_data_vectors(data_str data, mcmc_str mcin)
{
  if (mcin.impl == SP) {
    free(data.dataf);
    free(data.mvoutf);
  } else {
    free(data.data);
    if (mcin.impl == MP)
      free(data.dataf);
    else
      free(data.mvout);
  }
}

void free_sample_vectors(mcmc_v_str m)
{
  free(m.samples);
  free(m.burn);
  free(m.nsamples);
  free(m.nburn);
  free(m.sample_means);
}

void free_sample_vectors_sp(mcmc_v_str m)
{
  free(m.samplesf);
  free(m.burnf);
  free(m.nsamplesf);
  free(m.nburnf);
  free(m.sample_meansf);
}

void free_autocorrelation_vectors(sec_v_str secv)
{
  free(secv.shift);
  free(secv.circ);
}

void free_autocorrelation_vectors_sp(sec_v_str secv) { free(secv.circf); }

void free_mcmc_vectors(mcmc_int_v v, mcmc_str mcin)
{
  free(v.proposed);
  free(v.current);
  if (mcin.impl == MP) {
    free(v.proposedf);
    free(v.currentf);
  }
}

void free_mcmc_vectors_sp(mcmc_int_v v, mcmc_str mcin)
{
  (void)mcin;
  free(v.proposedf);
  free(v.currentf);
}

/* ------------------------------------------------------------ command line */

static const char *g_cli_in = NULL, *g_cli_out = NULL;
const char *mcmc_amd_cli_input(void) { return g_cli_in; }
const char *mcmc_amd_cli_output(void) { return g_cli_out; }

typedef struct {
  const char *flag;
  int *dst;  /* where the integer goes */
  int *dst2; /* optional second destination (-sz also sets gpu.size) */
  int lo, hi; /* accepted range; lo > hi means any */
} cli_opt;

static void parse_cli(int an, char *av[], const cli_opt *opts, int nopts)
{
  for (int ai = 1; ai < an;) {
    if (!strcmp(av[ai], "-in") || !strcmp(av[ai], "-out")) {
      if (ai + 1 >= an) {
        fprintf(stderr, "Missing argument to %s.", av[ai]);
        exit(1);
      }
      if (av[ai][1] == 'i')
        g_cli_in = av[ai + 1];
      else
        g_cli_out = av[ai + 1];
      ai += 2;
      continue;
    }
    const cli_opt *o = NULL;
    for (int k = 0; k < nopts; k++)
      if (!strcmp(av[ai], opts[k].flag)) o = &opts[k];
    if (!o) {
      fprintf(stderr, "Unknown option %s.\n", av[ai]);
      exit(1);
    }
    if (ai + 1 >= an) {
      fprintf(stderr, "Missing argument to %s.", o->flag);
      exit(1);
    }
    const int v = atoi(av[ai + 1]);
    if (o->lo <= o->hi && (v < o->lo || v > o->hi)) {
      fprintf(stderr, "Please enter a valid value.");
      exit(1);
    }
    *o->dst = v;
    if (o->dst2) *o->dst2 = v;
    ai += 2;
  }
}

#define ANY 1, 0
#define NONNEG 0, 2147483647

void read_inputs(int an, char *av[], mcmc_str *mcin, sec_str *sec)
{
  const cli_opt opts[] = {
      {"-d", &sec->fdata, NULL, ANY},      {"-sz", &mcin->Nd, NULL, ANY},     {"-dim", &mcin->ddata, NULL, ANY},
      {"-samp", &mcin->Ns, NULL, ANY},     {"-burn", &mcin->burnin, NULL, ANY}, {"-lag", &sec->lagidx, NULL, NONNEG},
      {"-tune", &mcin->tune, NULL, 0, 2},  {"-first", &sec->first, NULL, ANY},
  };
  parse_cli(an, av, opts, (int)(sizeof opts / sizeof opts[0]));
}

void read_inputs_gpu(int an, char *av[], mcmc_str *mcin, sec_str *sec, gpu_v_str *gpu)
{
  const cli_opt opts[] = {
      {"-d", &sec->fdata, NULL, ANY},
      {"-sz", &mcin->Nd, &gpu->size, ANY},
      {"-dim", &mcin->ddata, NULL, ANY},
      {"-samp", &mcin->Ns, NULL, ANY},
      {"-burn", &mcin->burnin, NULL, ANY},
      {"-lag", &sec->lagidx, NULL, NONNEG},
      {"-tune", &mcin->tune, NULL, 0, 2},
      {"-first", &sec->first, NULL, ANY},
      {"-maxThreads", &gpu->maxThreads, NULL, NONNEG},
      {"-kernel", &gpu->kernel, NULL, 0, 7},
      {"-cpuThresh", &gpu->cpuThresh, NULL, NONNEG},
      {"-maxBlocks", &gpu->maxBlocks, NULL, NONNEG},
  };
  parse_cli(an, av, opts, (int)(sizeof opts / sizeof opts[0]));
}

/* ---------------------------------------------------------------- printing */

static void print_times(out_str res)
{
  fprintf(stdout, "Tuning Time: %f ms\n", res.tuneTime);
  fprintf(stdout, "Burn In Time: %f ms\n", res.burnTime);
  fprintf(stdout, "Metropolis Time: %f ms\n", res.mcmcTime);
  fprintf(stdout, "Sampler Time: %f ms\n", res.samplerTime);
  fprintf(stdout, "Acceptance Ratio: %f\n", res.acceptance);
  fprintf(stdout, "Ess: %f\n", res.ess);
  fprintf(stdout, "Normalised Ess: %f samples/sec\n", res.ess * 1000 / res.mcmcTime);
}

void write_bandwidth_test_out(out_str res)
{
  fprintf(stdout, "Cublas Kernel:     cuTime = %f ms, cuBandwidth = %f GB/s\n", res.cuTime, res.cuBandwidth);
  fprintf(stdout, "Reduction Kernel:  kernelTime = %f ms, kernelBandwidth= %f GB/s\n", res.kernelTime,
          res.kernelBandwidth);
  fprintf(stdout, "Both Kernels:      LikelihoodTime = %f ms, Likelihood_Bandwidth = %f GB/s\n", res.gpuTime,
          res.gpuBandwidth);
  print_times(res);
}

void write_test_out(out_str res) { print_times(res); }

static void print_means(int d, const double *md, const float *mf)
{
  fprintf(stdout, "\nDimension: ");
  for (int k = 0; k < d; k++) fprintf(stdout, "%7d ", k);
  fprintf(stdout, "\nTheta:     ");
  for (int k = 0; k < d; k++) fprintf(stdout, "%+7.3f ", md ? md[k] : (double)mf[k]);
  fprintf(stdout, "\n\n");
}

void print_normalised_sample_means(mcmc_v_str mcdata, mcmc_str mcin) { print_means(mcin.ddata, mcdata.sample_means, NULL); }
void print_normalised_sample_means_sp(mcmc_v_str mcdata, mcmc_str mcin) { print_means(mcin.ddata, NULL, mcdata.sample_meansf); }

/* ------------------------------------------------------------------ CSV out */

static FILE *open_csv(const char *stem, const char *mode, const char *verb)
{
  char path[4096];
  fprintf(stdout, "%s %s.csv file. ", verb, stem);
  snprintf(path, sizeof path, "%s.csv", stem);
  FILE *fp = fopen(path, mode);
  if (!fp) {
    fprintf(stderr, "ERROR: cannot open %s\n", path);
    exit(1);
  }
  return fp;
}

#define DEFINE_WRITERS(T, SUFFIX)                                                    \
  void write_data##SUFFIX(char *filename, T *data, int sz, int dim)                  \
  {                                                                                  \
    FILE *fp = open_csv(filename, "w+", "Creating");                                 \
    for (long i = 0; i < sz; i++)                                                    \
      for (int j = 0; j < dim; j++) fprintf(fp, j < dim - 1 ? "%f, " : "%f\n", data[i * dim + j]); \
    fclose(fp);                                                                      \
    fprintf(stdout, "File created\n");                                               \
  }                                                                                  \
  void write_autocorr##SUFFIX(char *filename, T *autocorrelation, sec_str sec)       \
  {                                                                                  \
    FILE *fp = open_csv(filename, "w+", "Creating");                                 \
    fprintf(fp, "lag_k, autocorrelation\n");                                         \
    for (int i = 0; i < sec.lagidx; i++) fprintf(fp, "%d, %f\n", i, autocorrelation[i]); \
    fclose(fp);                                                                      \
    fprintf(stdout, "File created\n");                                               \
  }

DEFINE_WRITERS(double, )
DEFINE_WRITERS(float, _sp)

void write_performance_data(char *filename, out_str res, bool first)
{
  FILE *fp = open_csv(filename, "a", "Opening");
  if (first) {
    fprintf(fp, "Nd, dim, samples, burnSamples, sd, device, samplerTime, mcmcTime, tuneTime, burnTime, ");
    fprintf(fp, "cpuTime, kernel, blocksz, gpuTime, cuTime, kernelTime, ");
    fprintf(fp, "gpuBandwidth, cuBandwidth, kernelBandwidth, acceptance, ess\n");
  } else {
    fprintf(fp, "%d, %d, %d, %d, %f, %d, %f, %f, %f, %f, ", res.Nd, res.dim, res.samples, res.burnSamples, res.sd,
            res.device, res.samplerTime, res.mcmcTime, res.tuneTime, res.burnTime);
    fprintf(fp, "%f, %d, %d, %f, %f, %f, ", res.cpuTime, res.kernel, res.blocksz, res.gpuTime, res.cuTime,
            res.kernelTime);
    fprintf(fp, "%f, %f, %f, %f, %f\n", res.gpuBandwidth, res.cuBandwidth, res.kernelBandwidth, res.acceptance,
            res.ess);
  }
  fclose(fp);
  fprintf(stdout, "File appended.\n");
}

static char *stem(char *buf, size_t n, const char *dir, const char *name)
{
  snprintf(buf, n, "%s%s", dir, name);
  return buf;
}

void write_perf_out_csv(char *rootdir, out_str res, bool first)
{
  char f[4096];
  write_performance_data(stem(f, sizeof f, rootdir, "performance"), res, first);
}

void write_csv_outputs(char *rootdir, mcmc_v_str mcdata, mcmc_str mcin, sec_str sec, sec_v_str secv)
{
  char f[4096];
  fprintf(stdout, "*********************** File Output **********************\n");
  write_data(stem(f, sizeof f, rootdir, "burn_out"), mcdata.burn, mcin.burnin, mcin.ddata);
  write_data(stem(f, sizeof f, rootdir, "raw_out"), mcdata.samples, mcin.Ns, mcin.ddata);
  write_data(stem(f, sizeof f, rootdir, "norm_burn_out"), mcdata.nburn, mcin.burnin, mcin.ddata);
  write_data(stem(f, sizeof f, rootdir, "norm_raw_out"), mcdata.nsamples, mcin.Ns, mcin.ddata);
  write_data(stem(f, sizeof f, rootdir, "mean_out"), mcdata.sample_means, mcin.ddata, 1);
  fprintf(stdout, "**********************************************************\n");
  fprintf(stdout, "***************** Autocorrelation Output *****************\n");
  write_autocorr(stem(f, sizeof f, rootdir, "circular_autocorrelation"), secv.circ, sec);
  fprintf(stdout, "**********************************************************\n");
}

void write_csv_outputs_sp(char *rootdir, mcmc_v_str mcdata, mcmc_str mcin, sec_str sec, sec_v_str secv)
{
  char f[4096];
  fprintf(stdout, "*********************** File Output **********************\n");
  write_data_sp(stem(f, sizeof f, rootdir, "burn_out"), mcdata.burnf, mcin.burnin, mcin.ddata);
  write_data_sp(stem(f, sizeof f, rootdir, "raw_out"), mcdata.samplesf, mcin.Ns, mcin.ddata);
  write_data_sp(stem(f, sizeof f, rootdir, "norm_burn_out"), mcdata.nburnf, mcin.burnin, mcin.ddata);
  write_data_sp(stem(f, sizeof f, rootdir, "norm_raw_out"), mcdata.nsamplesf, mcin.Ns, mcin.ddata);
  write_data_sp(stem(f, sizeof f, rootdir, "mean_out"), mcdata.sample_meansf, mcin.ddata, 1);
  fprintf(stdout, "**********************************************************\n");
  fprintf(stdout, "***************** Autocorrelation Output *****************\n");
  write_autocorr_sp(stem(f, sizeof f, rootdir, "circular_autocorrelation"), secv.circf, sec);
  fprintf(stdout, "**********************************************************\n");
}

/* ------------------------------------------------------------------- CSV in */

/* One row per data point: label, x_1 .. x_d (scr/data_scr/synthetic_generation.py
 * writes exactly this).  The label's sign is folded into the stored value:
 * element = -label * x (data_util.c:22-35). */
static void load_csv(const char *path, int store, mcmc_str mcin, double *out_d, float *out_f)
{
  FILE *fp = fopen(path, "r");
  if (!fp) {
    fprintf(stderr, "ERROR: cannot open %s\n", path);
    exit(1);
  }
  fprintf(stdout, "Importing data. ");
  const long N = mcin.Nd;
  const int d = mcin.ddata;
  size_t cap = 1 << 16;
  char *line = (char *)malloc(cap);
  long row = 0;
  while (row < N) {
    size_t len = 0;
    int c;
    while ((c = fgetc(fp)) != EOF && c != '\n') {
      if (len + 2 > cap) line = (char *)realloc(line, cap *= 2);
      line[len++] = (char)c;
    }
    if (c == EOF && len == 0) break;
    line[len] = 0;
    if (len == 0) continue;
    char *p = line;
    int label = 1;
    for (int field = 0; field <= d && p && *p; field++) {
      char *end;
      const double v = strtod(p, &end);
      if (field == 0) {
        label = v > 0 ? 1 : -1;
      } else {
        const double x = -label * v;
        const long idx = (store == RowMajor) ? row * d + (field - 1) : (long)(field - 1) * N + row;
        if (out_d) out_d[idx] = x;
        if (out_f) out_f[idx] = (float)x;
      }
      p = strchr(end, ',');
      if (p) p++;
    }
    row++;
  }
  free(line);
  fclose(fp);
  fprintf(stdout, "Done\n");
}

void read_data(char *dir, int store, data_str data, mcmc_str mcin)
{
  load_csv(dir, store, mcin, data.data, mcin.impl == MP ? data.dataf : NULL);
}

void read_data_sp(char *dir, int store, data_str data, mcmc_str mcin) { load_csv(dir, store, mcin, NULL, data.dataf); }
