/*
 * oracle/ref_blas.c -- restatement of the one CBLAS call on the path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference calls cblas_dgemv(CblasColMajor, CblasNoTrans, m=Nd, n=d,
 * alpha=1, A, lda=Nd, x, 1, beta=0, y, 1) from libgslcblas
 * (src/mcmc_cpu/mcmc_cpu.c:436).  libgslcblas (GSL 2.x, cblas/source_gemv_r.h)
 * is absent here; its published reference loop for this case is:
 *     y := beta*y            (beta == 0 stores zeros)
 *     for j in 0..n-1:  t = alpha*x[j];  if t != 0:  for i: y[i] += t*A[lda*j+i]
 * so every row's dot product is a left-to-right sum of individually rounded
 * products.  Built with -ffp-contract=off so no product is fused into an FMA.
 */
#include "oracle.h"

#define DEFINE_GEMV(NAME, T)                                                                     \
  void NAME(long m, long n, T alpha, const T *A, long lda, const T *x, T beta, T *y)             \
  {                                                                                              \
    long i, j;                                                                                   \
    if (beta == (T)0) {                                                                          \
      for (i = 0; i < m; i++) y[i] = (T)0;                                                       \
    } else if (beta != (T)1) {                                                                   \
      for (i = 0; i < m; i++) y[i] *= beta;                                                      \
    }                                                                                            \
    if (alpha == (T)0) return;                                                                   \
    for (j = 0; j < n; j++) {                                                                    \
      const T t = alpha * x[j];                                                                  \
      if (t != (T)0) {                                                                           \
        const T *col = A + lda * j;                                                              \
        for (i = 0; i < m; i++) y[i] += t * col[i];                                              \
      }                                                                                          \
    }                                                                                            \
  }

DEFINE_GEMV(oracle_dgemv_colmajor_n, double)
DEFINE_GEMV(oracle_sgemv_colmajor_n, float)
