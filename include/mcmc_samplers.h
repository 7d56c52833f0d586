/*
 * mcmc_samplers.h -- the sampler / tuner entry points of libmcmc_amd.so.
 *
 * Same names, argument lists and by-value structs as the reference
 * (src/mcmc_gpu/mcmc_gpu.h:16-49, src/mcmc_gpu_sp/mcmc_gpu_sp.h:16-51,
 * src/mcmc_mp/mcmc_mp.h:18-52), so its host drivers (src/(all)/(all)_host.c,
 * tests/(all).c) link against this library unchanged.  They stay in C; the
 * per-iteration likelihood goes through the C-ABI in mhlik.h to hand-written
 * HIP kernels.
 *
 * The reference's inner helpers carry CUDA types in their signatures
 * (cublasHandle_t, curandGenerator_t, dev_v_str: mcmc_gpu.h:19-42).  Those are
 * implementation detail there and do not exist here; their roles are taken by
 * the functions at the bottom, which take an mhlik_ctx* instead.
 */
#ifndef MCMC_AMD_SAMPLERS_H
#define MCMC_AMD_SAMPLERS_H

#include "mcmc_types.h"
#include "mhlik.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the CPU path's entry points: src/mcmc_cpu/mcmc_cpu.h ----------------
 * Same signatures; the chain is the same (draws, decisions, samples), the
 * likelihood of every iteration runs on the MI355X (samplers/mcmc_cpu.c). */
void cpu_sampler(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata, out_str *res); /* :12 */
void tune_ess_cpu(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, double *initCond, int length); /* :22 */
void tune_target_a_cpu_v2(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, double *initCond,
                          double ratio, int max_reps);                                       /* :23 */
void tune_target_a_cpu(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, double *initCond); /* :24 */
double acceptance_ratio(mcmc_int_v mclocv, mcmc_int *mcloc, mcmc_str mcin, double lhood);  /* :32 */
double log_prior(double *sample, mcmc_str mcin);                                            /* :33 */
double log_likelihood(double *sample, data_str data, mcmc_str mcin, out_str *res);          /* :34 */

/* ---- double precision: src/mcmc_gpu/mcmc_gpu.h -------------------------- */
void gpu_sampler(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata, gpu_v_str gpu,
                 out_str *res);                                                             /* :16 */
void tune_ess_gpu(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, gpu_v_str gpu, double *initCond,
                  int length);                                                              /* :32 */
void tune_target_a_gpu_v2(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, gpu_v_str gpu,
                          double *initCond, double ratio, int max_reps);                    /* :33 */
void tune_target_a_gpu(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, gpu_v_str gpu,
                       double *initCond, int length);                                       /* :35 */
double acceptance_ratio_gpu(mcmc_int_v mclocv, mcmc_int *mcloc, mcmc_str mcin, double plhood); /* :45 */
double log_prior_gpu(double *sample, mcmc_str mcin);                                        /* :47 */
void print_gpu_info(void);                                                                  /* :49 */

/* ---- single precision: src/mcmc_gpu_sp/mcmc_gpu_sp.h -------------------- */
void sp_sampler(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata, gpu_v_str gpu,
                out_str *res);                                                              /* :16 */
void tune_ess_sp(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, gpu_v_str gpu, float *initCond,
                 int length);                                                               /* :35 */
void tune_target_a_sp_v2(data_str data, gsl_rng *r, mcmc_str mcin, mcmc_tune_str *mct, gpu_v_str gpu,
                         float *initCond, float ratio, int max_reps);                       /* :37 */
float acceptance_ratio_sp(mcmc_int_v mclocv, mcmc_int *mcloc, mcmc_str mcin, float plhood); /* :49 */
float log_prior_sp(float *sample, mcmc_str mcin);                                           /* :51 */

/* ---- mixed precision: src/mcmc_mp/mcmc_mp.h ------------------------------ */
void mp_sampler(data_str data, gsl_rng *r, mp_str *mpVar, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata,
                double lhoodBound, out_str *res);                                           /* :18 */
double acceptance_ratio_mp(mcmc_int_v mclocv, mcmc_int *mcloc, mcmc_str mcin, double plhood); /* :37 */
double log_prior_mp(double *sample, mcmc_str mcin);                                         /* :38 */
double get_lowerbound(data_str data, double *initialCond, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct); /* :41 */
void get_blockSz(data_str data, gsl_rng *r, mp_str *mpVar, mcmc_str mcin, mcmc_tune_str mct, double *initialCond,
                 double lhoodBound);                                                        /* :48 */

/* ---- chain statistics used by the tuners and drivers:
 *      src/util/processing_util.h (circular_autocorrelation, get_ess, ...) ---- */
double circular_autocorrelation(double *out_v, double *in_m, int data_dim, int samples, int lag_idx);
float circular_autocorrelation_sp(float *out_v, float *in_m, int data_dim, int samples, int lag_idx);
double get_ess(double *samples, int sample_sz, int sample_dim, int max_lag, double *autocorrelation_v);
float get_ess_sp(float *samples, int sample_sz, int sample_dim, int max_lag, float *autocorrelation_v);
void calculate_normalised_sample_means(mcmc_v_str mcdata, mcmc_str mcin);
void calculate_normalised_sample_means_sp(mcmc_v_str mcdata, mcmc_str mcin);

/* ---- RNG set-up: src/util/alloc_util.h ----------------------------------- */
void init_rng(gsl_rng **r);
void init_rng_seed(gsl_rng **r, unsigned long int seed);
void free_rng(gsl_rng *r);

/* ======================================================================== *
 *  Not in the reference                                                    *
 * ======================================================================== */

/* Inner loops with the CUDA handles replaced by the likelihood context
 * (roles of metropolis_gpu / burn_in_metropolis_gpu / gpu_likelihood_d,
 * mcmc_gpu.h:19-29). */
void metropolis_gpu(mhlik_ctx *lik, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata,
                    mcmc_int_v mclocv, mcmc_int *mcloc, int *accepted_samples, out_str *res);
void burn_in_metropolis_gpu(mhlik_ctx *lik, gsl_rng *r, mcmc_str mcin, mcmc_tune_str mct, mcmc_v_str mcdata,
                            mcmc_int_v mclocv, mcmc_int *mcloc);
double gpu_likelihood_d(mhlik_ctx *lik, mcmc_str mcin, double *samples, out_str *res);
float gpu_likelihood_f(mhlik_ctx *lik, mcmc_str mcin, float *samplesf, out_str *res);

/* K independent DP chains in lock step, one batched (MFMA) likelihood pass per
 * iteration.  r: K generators; samples [K][Ns][d]; burn [K][burnin][d] with
 * row 0 of each chain its start; accepted [K]; use_f32 != 0 evaluates on the
 * float matrix.  The context must hold the matching precision. */
void mcmc_amd_multichain_sampler(mhlik_ctx *lik, int K, gsl_rng **r, mcmc_str mcin, mcmc_tune_str mct,
                                 double *samples, double *burn, int *accepted, int use_f32);

/* K mixed-precision chains in lock step, one mhlik_mp_loglik_batched call per
 * iteration (float X.Theta pass on MFMA + double correction of bright /
 * re-examined blocks).  Chain k is mp_sampler run with generator r[k].
 * samples [K][Ns][d]; burn [K][burnin][d]; accepted [K]; bright_blocks [K] or
 * NULL.  shared_phase != 0: all chains re-examine the phase chain 0 drew (same
 * random streams, 1/ResampFactor of the double matrix read per iteration).
 * theta0: d values every chain starts from, or NULL for the reference's
 * theta = 1 (mcmc_mp.cu:433). */
void mcmc_amd_mp_multichain_sampler(mhlik_ctx *lik, int K, gsl_rng **r, mp_str *mpVar, mcmc_str mcin,
                                    mcmc_tune_str mct, double *samples, double *burn, int *accepted,
                                    double lhoodBound, int shared_phase, int *bright_blocks, const double *theta0);

/* By default every sampler / tuner call uploads data.data (or data.dataf) to
 * a fresh context and frees it on return, as the reference does
 * (mcmc_gpu.cu:53-58, 78-81).  A context attached here is used instead by
 * every following call whose (Nd, ddata) match it: upload once, share between
 * tuner and sampler, use device-generated data, or run row-sharded over
 * several GPUs.  Pass NULL to detach.  The caller keeps ownership. */
void mcmc_amd_attach_context(mhlik_ctx *lik);
mhlik_ctx *mcmc_amd_attached_context(void);

/* log_likelihood (mcmc_cpu.h:34) without an attached context keeps its device buffers between calls of one shape;
 * this releases them.  Nothing is released from an exit handler: what is still held at exit is left to the OS. */
void mcmc_amd_release_cached(void);

/* Per-iteration record of the MH loops (every loop appends while tracing). */
typedef struct {
  double lik;   /* L(theta') */
  double ratio; /* log acceptance ratio (MP: the ratio itself) */
  double logu;  /* log(u) (MP: u) */
  int accept;
} mcmc_amd_trace_rec;
void mcmc_amd_trace_begin(mcmc_amd_trace_rec *buf, long capacity);
long mcmc_amd_trace_end(void);

/* Progress lines on stdout like the reference's (default on = 1). */
void mcmc_amd_set_verbose(int on);

#ifdef __cplusplus
}
#endif

#endif /* MCMC_AMD_SAMPLERS_H */
