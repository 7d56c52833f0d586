/*
 * oracle/gpu_order.c -- the reference's GPU likelihood in its own summation
 * order, on the CPU.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows src/mcmc_gpu/mcmc_gpu.cu:177-257 (gpu_likelihood_d, reduction_d),
 * src/mcmc_gpu_sp/mcmc_gpu_sp.cu:179-264 (float twins),
 * src/reduction/reduction_kernel.cu:140-167 (reduceRegr2: transform, then a
 * sequential-addressing shared-memory tree), :253-280 (reduce2: the same tree
 * on plain values) and src/util/processing_util.c:43-86 (launch geometry for
 * kernel < 3).  cuBLAS's gemv order is unspecified; the gslcblas order is
 * used for the row dots.  This is what the SP tolerance (1e-4 relative) is
 * measured against besides the DP oracle: the float transform and the float
 * tree are the reference's, a serial float sum over 1e7 terms would not be.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>

/* processing_util.c:43-52 -- strictly greater power of two for exact powers */
static int next_pow2_ref(int n)
{
  int ctr = 0;
  while (n != 0) {
    n /= 2;
    ctr++;
  }
  return 1 << ctr;
}

/* processing_util.c:67-86, kernel < 3 branch */
static void geometry(long n, int max_threads, long *blocks, int *threads)
{
  int t = (n < max_threads) ? next_pow2_ref((int)n) : max_threads;
  *threads = t;
  *blocks = (n + t - 1) / t;
}

#define DEFINE_TREE(NAME, T, EXP, LOG)                                                          \
  T NAME(const T *v, long n, int max_threads, int cpu_thresh, int transform)                     \
  {                                                                                              \
    long blocks;                                                                                 \
    int threads;                                                                                 \
    geometry(n, max_threads, &blocks, &threads);                                                 \
    T *in = (T *)malloc(sizeof(T) * (n > 0 ? n : 1));                                            \
    T *out = (T *)malloc(sizeof(T) * (blocks > 0 ? blocks : 1));                                 \
    T *sdata = (T *)malloc(sizeof(T) * threads);                                                 \
    for (long i = 0; i < n; i++) in[i] = v[i];                                                   \
    long count = n;                                                                              \
    int first = 1;                                                                               \
    for (;;) {                                                                                   \
      for (long b = 0; b < blocks; b++) {                                                        \
        for (int t = 0; t < threads; t++) {                                                      \
          long i = b * threads + t;                                                              \
          T x = (T)0;                                                                            \
          if (i < count) x = (first && transform) ? -LOG((T)1 + EXP(in[i])) : in[i];             \
          sdata[t] = x;                                                                          \
        }                                                                                        \
        for (int s = threads / 2; s > 0; s >>= 1)                                                \
          for (int t = 0; t < s; t++) sdata[t] += sdata[t + s];                                  \
        out[b] = sdata[0];                                                                       \
      }                                                                                          \
      first = 0;                                                                                 \
      count = blocks;                                                                            \
      if (count < cpu_thresh) break;                                                             \
      /* next pass: partials become the input (the reference's D2D memcpy) */                    \
      for (long i = 0; i < count; i++) in[i] = out[i];                                           \
      free(sdata);                                                                               \
      geometry(count, max_threads, &blocks, &threads);                                           \
      sdata = (T *)malloc(sizeof(T) * threads);                                                  \
    }                                                                                            \
    T total = (T)0;                                                                              \
    for (long i = 0; i < count; i++) total += out[i]; /* serial CPU tail */                      \
    free(in);                                                                                    \
    free(out);                                                                                   \
    free(sdata);                                                                                 \
    return total;                                                                                \
  }

DEFINE_TREE(oracle_tree_reduce_d, double, exp, log)
DEFINE_TREE(oracle_tree_reduce_f, float, expf, logf)

double oracle_gpu_likelihood_d(const double *X, long N, int d, long lda, const double *theta, double *scratch,
                               int max_threads, int cpu_thresh)
{
  oracle_dgemv_colmajor_n(N, d, 1.0, X, lda, theta, 0.0, scratch);
  return oracle_tree_reduce_d(scratch, N, max_threads, cpu_thresh, 1);
}

float oracle_gpu_likelihood_f(const float *X, long N, int d, long lda, const float *theta, float *scratch,
                              int max_threads, int cpu_thresh)
{
  oracle_sgemv_colmajor_n(N, d, 1.0f, X, lda, theta, 0.0f, scratch);
  return oracle_tree_reduce_f(scratch, N, max_threads, cpu_thresh, 1);
}
