/*
 * oracle/ess.c -- restatement of the autocorrelation / ESS helpers the
 * tuners call (src/util/processing_util.c:206-260, 318-354, 394-403).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * One deliberate difference: the reference wraps the shifted index with
 * `shift_idx > samples` (processing_util.c:234), which reads row `samples`
 * -- one past the end of the chain -- before wrapping.  That read is
 * undefined behaviour and cannot be restated; the wrap here happens at
 * `shift_idx >= samples`, the evident intent of a circular shift.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>

static void column_mean(int d, int n, const double *m, double *mu)
{
  for (int k = 0; k < d; k++) mu[k] = 0;
  for (int s = 0; s < n; s++)
    for (int k = 0; k < d; k++) mu[k] += m[(long)s * d + k];
  for (int k = 0; k < d; k++) mu[k] /= n;
}

static void column_variance(int d, int n, const double *m, double *var, const double *mu)
{
  for (int k = 0; k < d; k++) var[k] = 0;
  for (int s = 0; s < n; s++)
    for (int k = 0; k < d; k++) var[k] += pow(m[(long)s * d + k] - mu[k], 2);
  for (int k = 0; k < d; k++) var[k] /= n;
}

double circular_autocorrelation(double *out_v, double *in_m, int data_dim, int samples, int lag_idx)
{
  double *mean = (double *)malloc(sizeof(double) * data_dim);
  double *variance = (double *)malloc(sizeof(double) * data_dim);
  double *acc = (double *)malloc(sizeof(double) * data_dim);
  double sum = 0;

  column_mean(data_dim, samples, in_m, mean);
  column_variance(data_dim, samples, in_m, variance, mean);

  for (int k = 0; k < lag_idx; k++) {
    for (int i = 0; i < data_dim; i++) acc[i] = 0;
    for (int s = 0; s < samples; s++) {
      int shifted = s + k;
      if (shifted >= samples) shifted -= samples; /* circular shift */
      for (int i = 0; i < data_dim; i++)
        acc[i] += (in_m[(long)s * data_dim + i] - mean[i]) * (in_m[(long)shifted * data_dim + i] - mean[i]);
    }
    double prod = 1;
    for (int i = 0; i < data_dim; i++) prod *= acc[i] / (variance[i] * (samples - k));
    out_v[k] = prod;
    sum += out_v[k];
  }
  sum -= out_v[0]; /* lag 0 does not count */

  free(mean);
  free(variance);
  free(acc);
  return sum;
}

double get_ess(double *samples, int sample_sz, int sample_dim, int max_lag, double *autocorrelation_v)
{
  double s = circular_autocorrelation(autocorrelation_v, samples, sample_dim, sample_sz, max_lag);
  return sample_sz / (1 + 2 * s);
}
