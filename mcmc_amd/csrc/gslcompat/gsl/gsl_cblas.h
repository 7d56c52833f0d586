/*
 * gslcompat: <gsl/gsl_cblas.h> as far as the reference needs it to compile.
 *
 * src/util/resources.h:14 of the reference includes this header for every
 * translation unit; the only call site is cblas_dgemv in the CPU likelihood
 * (src/mcmc_cpu/mcmc_cpu.c:436), which libmcmc_amd.so replaces with the HIP
 * pass -- so the enums and the two prototypes are declared here and NOT
 * defined by libmcmc_amd.so (the product has no CPU likelihood).  A program
 * that calls them links a real CBLAS (-lgslcblas).  The oracle has its own
 * restatement of the gslcblas loop (oracle/ref_blas.c).
 */
#ifndef MCMC_AMD_GSLCOMPAT_CBLAS_H
#define MCMC_AMD_GSLCOMPAT_CBLAS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum CBLAS_ORDER { CblasRowMajor = 101, CblasColMajor = 102 };
enum CBLAS_TRANSPOSE { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 };
typedef enum CBLAS_ORDER CBLAS_ORDER_t;
typedef enum CBLAS_TRANSPOSE CBLAS_TRANSPOSE_t;

void cblas_dgemv(const enum CBLAS_ORDER order, const enum CBLAS_TRANSPOSE TransA, const int M, const int N,
                 const double alpha, const double *A, const int lda, const double *X, const int incX,
                 const double beta, double *Y, const int incY);
void cblas_sgemv(const enum CBLAS_ORDER order, const enum CBLAS_TRANSPOSE TransA, const int M, const int N,
                 const float alpha, const float *A, const int lda, const float *X, const int incX,
                 const float beta, float *Y, const int incY);

#ifdef __cplusplus
}
#endif
#endif
