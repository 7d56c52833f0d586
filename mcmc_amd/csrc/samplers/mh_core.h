/*
 * mh_core.h -- shared pieces of the C samplers: the one MH iteration that the
 * reference spells out eight times per precision, context acquisition, the
 * print-and-exit error convention, tracing.
 */
#ifndef MCMC_AMD_MH_CORE_H
#define MCMC_AMD_MH_CORE_H

#include "mcmc_samplers.h"

#include <stdio.h>

extern int mcmc_amd_verbose;
#define MH_SAY(...)                                      \
  do {                                                   \
    if (mcmc_amd_verbose) fprintf(stdout, __VA_ARGS__);  \
  } while (0)

/* gpuErrchk of the reference (mcmc_gpu.cu:9-17): print, then exit(code). */
void mh_check(int status, mhlik_ctx *lik, const char *file, int line);
#define MH_CHECK(status, lik) mh_check((status), (lik), __FILE__, __LINE__)

/* A context for (Nd, ddata, precision): the attached one when it matches,
 * else a fresh one with the host matrix uploaded.  *owned tells whether the
 * caller must destroy it. */
mhlik_ctx *mh_acquire(data_str data, mcmc_str mcin, int precision, int *owned);
void mh_release(mhlik_ctx *lik, int owned);

void mh_trace_push(double lik, double ratio, double logu, int accept);

/* work vectors (malloc_mcmc_vectors / _sp, src/util/alloc_util.c:126-176) */
void mh_alloc_vectors(mcmc_int_v *v, int d, int want_double, int want_float);
void mh_free_vectors(mcmc_int_v v);

/* One double-precision MH iteration: d ziggurat draws, likelihood, prior,
 * one uniform, accept test.  Returns 1 if accepted.  prior_sd and the accept
 * rule differ between the DP sampler (10, log(u) <= ratio) and the MP one
 * (5, u <= exp(ratio)). */
typedef double (*mh_lik_d)(void *env, mcmc_str mcin, mcmc_int_v v, out_str *res);
int mh_step_d(mh_lik_d lik, void *env, gsl_rng *r, mcmc_str mcin, double sd, mcmc_int_v v, mcmc_int *loc,
              out_str *res, int prior_sd, int exp_rule);
double mh_log_prior_d(const double *sample, int d, int prior_sd);

/* n iterations of mh_step_d (log(u) <= ratio rule) with the chain kept on the
 * device: the host draws the batch's random numbers in the same order, the
 * device runs the iterations back to back (mhlik_chain_submit_f64 / _collect_f64, two batches in flight).  rows_out:
 * n rows of d, the state after each iteration (NULL: not kept).  v.current /
 * loc->cposterior are the state on entry and on return. */
int mh_device_chain_enabled(void);
int mh_device_stats_enabled(void);
long mh_chain_batch_size(long k, long left); /* size of the k-th submitted batch when `left` iterations remain */
void mh_chain_steps_d(mhlik_ctx *lik, gsl_rng *r, mcmc_str mcin, double sd, mcmc_int_v v, mcmc_int *loc, long n,
                      double *rows_out, int *accepted, out_str *res, int prior_sd);

#endif
