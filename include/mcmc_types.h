/*
 * mcmc_types.h -- the by-value argument structs of the sampler / tuner entry
 * points.  Field names, order and types follow the reference's
 * src/util/resources.h:20-214 exactly, because every sampler signature passes
 * these structs BY VALUE: a caller compiled against the reference's header
 * must be able to link against this library unchanged.
 * Layout is checked by the static asserts at the bottom and by
 * tests/test_abi.py.
 *
 * The reference pulls <gsl/gsl_rng.h> in through this header; here the RNG
 * type comes from mcmc_amd/csrc/gslcompat (same struct layout as GSL 2.x) or
 * from a real GSL when one is on the include path.
 */
#ifndef MCMC_AMD_TYPES_H
#define MCMC_AMD_TYPES_H

#include <stddef.h>
#include <stdint.h>

#include <gsl/gsl_randist.h>
#include <gsl/gsl_rng.h>

#ifdef __cplusplus
extern "C" {
#endif

/* resources.h:20-47 */
enum dataread { RowMajor, ColMajor };
enum implementation { CPU, GPU, SP, MP, CA };
enum kernel_selection {
  ModuloReduction,
  InterleavedReduction,
  SequentialReduction,
  FirstOnLoadReduction,
  UnrollLastReduction,
  UnrollAllReduction,
  MultipleThreadsReduction
};
enum reductionType { LogRegression, Reduction, LogReduction };

/* resources.h:49-59 -- host design matrix: data[j*Nd + i] = -y_i * x_ij */
typedef struct data_vectors {
  double *data;
  float *dataf;
  int8_t *labels;
  double *mvout;
  float *mvoutf;
  int8_t *zlabels;
  int *zidx;
} data_str;

/* resources.h:61-74 -- chains, row-major [iteration][dimension] */
typedef struct mcmc_vectors {
  double *samples;
  double *nsamples;
  double *burn;
  double *nburn;
  double *sample_means;
  float *samplesf;
  float *nsamplesf;
  float *burnf;
  float *nburnf;
  float *sample_meansf;
} mcmc_v_str;

/* resources.h:76-85 */
typedef struct mcmc_vars {
  int ddata;  /* columns of the design matrix, bias included */
  int Nd;     /* rows */
  int Ndmap;
  int Ns;     /* samples kept */
  int burnin; /* burn-in length */
  int tune;   /* 0 none, 1 target acceptance, 2 max ESS */
  int impl;   /* enum implementation */
} mcmc_str;

/* resources.h:87-95 */
typedef struct mcmc_mp_vars {
  int dataBlocks;
  int dataBlockSz;
  int brightBlocks;
  int darkBlocks;
  int ResampFactor;
  float ratio;
} mp_str;

/* resources.h:97-101 */
typedef struct tuning_par {
  double rwsd;
  float rwsdf;
} mcmc_tune_str;

/* resources.h:103-109 */
typedef struct mcmc_internal_vectors {
  double *proposed;
  double *current;
  float *proposedf;
  float *currentf;
} mcmc_int_v;

/* resources.h:111-122 */
typedef struct mcmc_internal {
  double cposterior;
  double pposterior;
  double acceptance;
  double u;
  float cposteriorf;
  float pposteriorf;
  float acceptancef;
  float uf;
} mcmc_int;

/* resources.h:150-159 -- launch knobs of the reference's CUDA path.  Kept so
 * the signatures match; the MI355X path ignores every field. */
typedef struct gpu_parameters {
  int size;
  int threads;
  int blocks;
  int maxThreads;
  int maxBlocks;
  int kernel;
  int cpuThresh;
} gpu_v_str;

/* resources.h:181-195 */
typedef struct secondary_vectors {
  double *shift;
  double *circ;
  float *shiftf;
  float *circf;
} sec_v_str;

typedef struct secondary_vars {
  int fdata;
  int lagidx;
  int first;
} sec_str;

/* resources.h:197-214 */
typedef struct outputs {
  float cuTime, cuBandwidth;
  float kernelTime, kernelBandwidth;
  float gpuTime, gpuBandwidth;
  float cpuTime;

  float tuneTime;
  float burnTime;
  float mcmcTime;
  float samplerTime;

  int Nd, dim, kernel, blocksz;
  float sd, acceptance, ess;
  int samples, burnSamples;
  int device;
} out_str;

#ifdef __cplusplus
}
#endif

#if defined(__cplusplus)
#define MCMC_AMD_SASSERT(c, m) static_assert(c, m)
#else
#define MCMC_AMD_SASSERT(c, m) _Static_assert(c, m)
#endif
MCMC_AMD_SASSERT(sizeof(mcmc_str) == 7 * sizeof(int), "mcmc_str layout");
MCMC_AMD_SASSERT(sizeof(gpu_v_str) == 7 * sizeof(int), "gpu_v_str layout");
MCMC_AMD_SASSERT(sizeof(mp_str) == 6 * 4, "mp_str layout");
MCMC_AMD_SASSERT(sizeof(mcmc_tune_str) == 16, "mcmc_tune_str layout");
MCMC_AMD_SASSERT(sizeof(out_str) == 21 * 4, "out_str layout");
MCMC_AMD_SASSERT(offsetof(out_str, Nd) == 11 * 4, "out_str.Nd offset");
MCMC_AMD_SASSERT(sizeof(data_str) == 7 * sizeof(void *), "data_str layout");
MCMC_AMD_SASSERT(sizeof(mcmc_v_str) == 10 * sizeof(void *), "mcmc_v_str layout");

#endif /* MCMC_AMD_TYPES_H */
