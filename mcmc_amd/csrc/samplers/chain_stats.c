/*
 * chain_stats.c -- autocorrelation, effective sample size and normalised
 * means of a chain: the helpers from src/util/processing_util.c that the
 * tuners (tune_ess_*) and the reference's drivers call
 * (circular_autocorrelation :206-260, _sp :262-316, get_mean/variance
 * :318-392, get_ess :394-414, normalise_samples :88-100, get_dim_mean
 * :114-124, calculate_normalised_sample_means :416-438).
 *
 * One deliberate difference: the circular shift wraps at `>= samples`; the
 * reference wraps at `> samples` (:234, :290) and so reads one row past the
 * end of the chain, which is undefined behaviour.
 */
#include "mcmc_samplers.h"

#include <math.h>
#include <stdlib.h>

#define DEFINE_STATS(T, SUFFIX, POW)                                                                       \
  static void col_mean##SUFFIX(int d, int n, const T *m, T *mu)                                            \
  {                                                                                                        \
    for (int k = 0; k < d; k++) mu[k] = 0;                                                                 \
    for (int s = 0; s < n; s++)                                                                            \
      for (int k = 0; k < d; k++) mu[k] += m[(long)s * d + k];                                             \
    for (int k = 0; k < d; k++) mu[k] /= n;                                                                \
  }                                                                                                        \
  static void col_var##SUFFIX(int d, int n, const T *m, T *var, const T *mu)                               \
  {                                                                                                        \
    for (int k = 0; k < d; k++) var[k] = 0;                                                                \
    for (int s = 0; s < n; s++)                                                                            \
      for (int k = 0; k < d; k++) var[k] += POW(m[(long)s * d + k] - mu[k], 2);                            \
    for (int k = 0; k < d; k++) var[k] /= n;                                                               \
  }                                                                                                        \
  T circular_autocorrelation##SUFFIX(T *out_v, T *in_m, int data_dim, int samples, int lag_idx)            \
  {                                                                                                        \
    T *mean = (T *)malloc(sizeof(T) * data_dim);                                                           \
    T *variance = (T *)malloc(sizeof(T) * data_dim);                                                       \
    T *acc = (T *)malloc(sizeof(T) * data_dim);                                                            \
    T sum = 0;                                                                                             \
    col_mean##SUFFIX(data_dim, samples, in_m, mean);                                                       \
    col_var##SUFFIX(data_dim, samples, in_m, variance, mean);                                              \
    for (int k = 0; k < lag_idx; k++) {                                                                    \
      for (int i = 0; i < data_dim; i++) acc[i] = 0;                                                       \
      for (int s = 0; s < samples; s++) {                                                                  \
        int shifted = s + k;                                                                               \
        if (shifted >= samples) shifted -= samples;                                                        \
        for (int i = 0; i < data_dim; i++)                                                                 \
          acc[i] += (in_m[(long)s * data_dim + i] - mean[i]) * (in_m[(long)shifted * data_dim + i] - mean[i]); \
      }                                                                                                    \
      T prod = 1;                                                                                          \
      for (int i = 0; i < data_dim; i++) prod *= acc[i] / (variance[i] * (samples - k));                   \
      out_v[k] = prod;                                                                                     \
      sum += out_v[k];                                                                                     \
    }                                                                                                      \
    sum -= out_v[0];                                                                                       \
    free(mean);                                                                                            \
    free(variance);                                                                                        \
    free(acc);                                                                                             \
    return sum;                                                                                            \
  }                                                                                                        \
  T get_ess##SUFFIX(T *samples, int sample_sz, int sample_dim, int max_lag, T *autocorrelation_v)          \
  {                                                                                                        \
    T s = circular_autocorrelation##SUFFIX(autocorrelation_v, samples, sample_dim, sample_sz, max_lag);    \
    return sample_sz / (1 + 2 * s);                                                                        \
  }                                                                                                        \
  static void normalise##SUFFIX(const T *raw, T *norm, int d, int n)                                       \
  {                                                                                                        \
    for (int s = 0; s < n; s++)                                                                            \
      for (int k = 0; k < d; k++) norm[(long)s * d + k] = raw[(long)s * d + k] / raw[(long)s * d];         \
  }                                                                                                        \
  static T dim_mean##SUFFIX(const T *m, int d, int k, int n)                                               \
  {                                                                                                        \
    T mu = 0;                                                                                              \
    for (int s = 0; s < n; s++) mu += m[(long)s * d + k];                                                  \
    return mu / n;                                                                                         \
  }

DEFINE_STATS(double, , pow)
DEFINE_STATS(float, _sp, pow)

void calculate_normalised_sample_means(mcmc_v_str mcdata, mcmc_str mcin)
{
  normalise(mcdata.burn, mcdata.nburn, mcin.ddata, mcin.burnin);
  normalise(mcdata.samples, mcdata.nsamples, mcin.ddata, mcin.Ns);
  for (int k = 0; k < mcin.ddata; k++) mcdata.sample_means[k] = dim_mean(mcdata.nsamples, mcin.ddata, k, mcin.Ns);
}

void calculate_normalised_sample_means_sp(mcmc_v_str mcdata, mcmc_str mcin)
{
  normalise_sp(mcdata.burnf, mcdata.nburnf, mcin.ddata, mcin.burnin);
  normalise_sp(mcdata.samplesf, mcdata.nsamplesf, mcin.ddata, mcin.Ns);
  for (int k = 0; k < mcin.ddata; k++)
    mcdata.sample_meansf[k] = dim_mean_sp(mcdata.nsamplesf, mcin.ddata, k, mcin.Ns);
}
