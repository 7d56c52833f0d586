/*
 * oracle/extras.c -- helpers that have no counterpart in the reference:
 * a threaded variant of log_likelihood for the CPU-baseline timing, and a
 * batched wrapper used to check the K-chain GPU mode.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

double oracle_log_likelihood_mt(const double *theta, const double *X, long N, int d, long lda, int threads)
{
  if (threads < 1) threads = 1;
  double *part = (double *)calloc((size_t)threads, sizeof(double));
  const long blk = 4096; /* rows handled at a time: y stays in L1/L2 */
#pragma omp parallel num_threads(threads)
  {
    int t = 0, nt = 1;
#ifdef _OPENMP
    t = omp_get_thread_num();
    nt = omp_get_num_threads();
#endif
    long lo = N * t / nt, hi = N * (t + 1) / nt;
    double y[4096];
    double s = 0;
    for (long r0 = lo; r0 < hi; r0 += blk) {
      long m = (hi - r0 < blk) ? hi - r0 : blk;
      oracle_dgemv_colmajor_n(m, d, 1.0, X + r0, lda, theta, 0.0, y);
      for (long i = 0; i < m; i++) s -= log(1 + exp(y[i]));
    }
    part[t] = s;
  }
  double total = 0;
  for (int t = 0; t < threads; t++) total += part[t];
  free(part);
  return total;
}

void oracle_log_likelihood_batched(const double *Theta, int K, const double *X, long N, int d, long lda,
                                   double *mvout, double *out)
{
  for (int k = 0; k < K; k++) {
    double s = 0;
    oracle_dgemv_colmajor_n(N, d, 1.0, X, lda, Theta + (long)k * d, 0.0, mvout);
    for (long i = 0; i < N; i++) s -= log(1 + exp(mvout[i]));
    out[k] = s;
  }
}
