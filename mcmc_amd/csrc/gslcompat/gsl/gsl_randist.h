/*
 * gslcompat: Gaussian variate + density, the two gsl_randist entry points the
 * samplers call (src/mcmc_cpu/mcmc_cpu.c:53 and :413 in the reference).
 */
#ifndef MCMC_AMD_GSLCOMPAT_RANDIST_H
#define MCMC_AMD_GSLCOMPAT_RANDIST_H

#include <gsl/gsl_rng.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Marsaglia-Tsang ziggurat, 128 levels, one 32-bit draw split 8/24. */
double gsl_ran_gaussian_ziggurat(const gsl_rng *r, double sigma);

/* exp(-x^2 / 2 sigma^2) / (sqrt(2 pi) |sigma|) */
double gsl_ran_gaussian_pdf(double x, double sigma);

#ifdef __cplusplus
}
#endif

#endif
