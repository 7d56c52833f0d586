/*
 * mhlik.h -- C-ABI of the MI355X Metropolis-Hastings likelihood library
 * (libmhlik.so).  Plain C types only: no HIP, torch or C++ types cross this
 * boundary.
 *
 * What it replaces in the reference (cstyl/mcmc), call for call:
 *
 *   cudaMalloc x4 + cudaMemcpy of X   src/mcmc_gpu/mcmc_gpu.cu:53-58, 409-414
 *                                     -> mhlik_create + mhlik_upload_f64/_f32
 *                                        (once per dataset, shared by tuner
 *                                        and sampler)
 *   gpu_likelihood_d + reduction_d    src/mcmc_gpu/mcmc_gpu.cu:214-257,177-212
 *     (cublasDgemv, reduceRegr2,      src/reduction/reduction_kernel.cu:140-167,
 *      reduce2 passes, D2H + CPU sum)   253-280, 387-458
 *                                     -> mhlik_loglik_f64
 *   gpu_likelihood_f + reduction_f    src/mcmc_gpu_sp/mcmc_gpu_sp.cu:216-264,179-214
 *                                     -> mhlik_loglik_f32
 *   mp_likelihood + mcmc_mp_kernel    src/mcmc_mp/mcmc_mp.cu:475-552, 28-200
 *     + reduction_mp_d/_f               554-627
 *                                     -> mhlik_mp_loglik
 *   likelihood_bound/call_boundKernel src/mcmc_mp/mcmc_mp.cu:746-821, 202-280
 *                                     -> mhlik_mp_bound
 *   cudaFree x4                       src/mcmc_gpu/mcmc_gpu.cu:78-81
 *                                     -> mhlik_destroy
 *
 * The log-likelihood is  L(theta) = - sum_i log(1 + exp(sum_j X[i,j]*theta[j]))
 * with X[i,j] = -y_i * x_ij (label folded in, src/util/data_util.c:33).
 *
 * Threading: a context is not re-entrant; one host thread drives it.  Every
 * evaluation call blocks until its result is on the host, like the
 * reference's (the value feeds the host accept test, mcmc_gpu.cu:107-114).
 * All sizes that can exceed 2^31 are int64_t (the reference overflows int at
 * N=1e8, d=64: mcmc_gpu.cu:48).
 */
#ifndef MHLIK_H
#define MHLIK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mhlik_ctx mhlik_ctx;

/* status codes (0 = ok); text via mhlik_last_error */
enum {
  MHLIK_OK = 0,
  MHLIK_EINVAL = 1, /* bad argument */
  MHLIK_ENOMEM = 2, /* host or device allocation failed */
  MHLIK_EHIP = 3,   /* a HIP runtime call failed */
  MHLIK_ESTATE = 4, /* call not valid in this state (e.g. no data uploaded) */
  MHLIK_ECOMM = 5   /* RCCL not available / collective failed */
};

/* which copies of the design matrix the context keeps in HBM (bit mask) */
enum { MHLIK_F64 = 1, MHLIK_F32 = 2 };

/* HBM layout of the design matrix */
enum {
  MHLIK_LAYOUT_DEFAULT = 0,
  MHLIK_LAYOUT_TILED = 1,   /* [tile][column][row-in-tile], tile contiguous */
  MHLIK_LAYOUT_COLMAJOR = 2 /* the reference's column-major, padded lda */
};

#define MHLIK_MAX_SEGMENTS 64
#define MHLIK_MAX_BATCH 64
#define MHLIK_COMM_ID_BYTES 128

typedef struct mhlik_config {
  int64_t n_rows;    /* N: rows of the whole (global) design matrix */
  int32_t n_cols;    /* d: columns, bias included */
  int32_t precision; /* MHLIK_F64 | MHLIK_F32 */
  int32_t device;    /* HIP device ordinal; -1 = current device */
  int32_t rank;      /* this context's shard, 0 <= rank < world */
  int32_t world;     /* number of shards (processes, one GPU each); 1 = no sharding */
  int32_t segments;  /* fixed row segments the sum is grouped by; 0 = 8.
                        Must be a multiple of world.  The result is bitwise
                        identical for every world that divides it. */
  int32_t layout;    /* MHLIK_LAYOUT_* */
  int32_t variant;   /* kernel variant; 0 = default (tuning knob) */
  int32_t timing;    /* !=0: bracket the likelihood kernel with events */
  int32_t segment_align; /* 0/1: segments of ceil(N/segments) rows.  > 1: that count rounded up to a multiple of this
                            (trailing segments may then be empty).  Mixed precision over several segments / GPUs
                            needs a multiple of 1024, so that data blocks and tiles never straddle a shard. */
  int32_t reserved[5];
} mhlik_config;

/* Create a context on config->device.  Allocates nothing large until data is
 * uploaded or filled.  On failure *out is NULL and mhlik_last_error(NULL)
 * holds the reason. */
int mhlik_create(const mhlik_config *config, mhlik_ctx **out);
void mhlik_destroy(mhlik_ctx *ctx);

/* Last error text of this context (ctx == NULL: of the last failed create). */
const char *mhlik_last_error(const mhlik_ctx *ctx);

/* Global row range [begin, begin+count) this shard holds. */
int mhlik_local_rows(const mhlik_ctx *ctx, int64_t *begin, int64_t *count);

/* Upload the shard from a host column-major matrix: global row g, column j
 * is X[j*ld + (g - row0)].  Pass the whole matrix with row0 = 0, or only this
 * shard's rows with row0 = its first global row.  Replaces the cudaMemcpy of
 * data.data / data.dataf (mcmc_gpu.cu:58, mcmc_gpu_sp.cu:58, mcmc_mp.cu:320-321). */
int mhlik_upload_f64(mhlik_ctx *ctx, const double *X, int64_t ld, int64_t row0);
int mhlik_upload_f32(mhlik_ctx *ctx, const float *X, int64_t ld, int64_t row0);

/* Fill the shard on the device with the seeded synthetic logistic-regression
 * data set (distribution of scr/data_scr/synthetic_generation.py:33-65: bias
 * column 1, other features U(0,1), integer beta in [-15,15) with beta_0 = 1,
 * y = sign(x.beta), stored as -y*x).  Counter-based: any shard of any N is
 * produced identically on any number of GPUs.  Fills every precision the
 * context holds (the float copy is the rounded double, data_util.c:72). */
int mhlik_fill_synthetic(mhlik_ctx *ctx, uint64_t seed);

/* The same generator on the host: rows [row0, row0+rows) of the N x d data
 * set, column-major with leading dimension ld.  For building the CPU copy
 * that the oracle / CPU baseline reads. */
int mhlik_synthetic_host_f64(uint64_t seed, int64_t n_rows, int32_t n_cols, int64_t row0, int64_t rows,
                             double *X, int64_t ld);
int mhlik_synthetic_beta(uint64_t seed, int32_t n_cols, int32_t *beta);

/* Copy the shard back to the host as column-major (debug / tests). */
int mhlik_download_f64(mhlik_ctx *ctx, double *X, int64_t ld);
int mhlik_download_f32(mhlik_ctx *ctx, float *X, int64_t ld);

/* L(theta) over ALL rows (all shards: with world > 1 every rank must call and
 * every rank gets the same bits).  theta is a host array of n_cols values. */
int mhlik_loglik_f64(mhlik_ctx *ctx, const double *theta, double *out);
int mhlik_loglik_f32(mhlik_ctx *ctx, const float *theta, float *out);

/* mhlik_loglik_f64 for callers that cannot allocate (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has NO
 * counterpart in the reference.  Same kernels and the same bits as
 * mhlik_loglik_f64, but the per-tile partial sums live in a static device
 * array of MHLIK_NOALLOC_MAX_TILES doubles (2 MiB, held by every GPU the
 * library runs on until the process ends) instead of the context's buffer,
 * and the call makes no hipMalloc / hipFree / allocating library call.
 * Limits (MHLIK_EINVAL / MHLIK_ESTATE with a message otherwise): at most
 * MHLIK_NOALLOC_MAX_TILES row tiles (512 rows each by default: N <= 134M),
 * n_cols <= 256 (theta travels in the kernarg), no RCCL communicator.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (two contexts driven from two threads) race.  Nothing needs clearing
 * between calls: every slot the sum reads is written by the same call. */
#define MHLIK_NOALLOC_MAX_TILES (1 << 18)
int mhlik_loglik_f64_noalloc(mhlik_ctx *ctx, const double *theta, double *out);

/* K proposals in one pass over X (X.Theta on the matrix cores).  Theta is
 * host row-major [K][n_cols] (chain k contiguous); out[k] = L(Theta[k]).
 * 1 <= K <= MHLIK_MAX_BATCH.  No counterpart in the reference (one chain,
 * one theta per call: mcmc_gpu.cu:101-107); BASELINE.json config 5. */
int mhlik_loglik_batched_f64(mhlik_ctx *ctx, const double *Theta, int32_t K, double *out);
int mhlik_loglik_batched_f32(mhlik_ctx *ctx, const float *Theta, int32_t K, double *out);

/* The same evaluation in two halves, for a caller that has host work to do while the device runs (the K-chain
 * samplers draw the next iteration's random numbers meanwhile; no counterpart in the reference, whose likelihood
 * call blocks, mcmc_gpu.cu:107).  _begin stages Theta and queues the kernels; _end waits and writes the K results.
 * Between the two the context accepts no other evaluation (MHLIK_ESTATE), and Theta may be reused at once (it has
 * been copied). */
int mhlik_loglik_batched_begin_f64(mhlik_ctx *ctx, const double *Theta, int32_t K);
int mhlik_loglik_batched_begin_f32(mhlik_ctx *ctx, const float *Theta, int32_t K);
int mhlik_loglik_batched_end(mhlik_ctx *ctx, double *out);

/* How mhlik_loglik_batched_f32 forms the float dots x_i.theta_k (no counterpart in the reference):
 *   MHLIK_BATCHED_F32_EXACT  f32 MFMA: bit-for-bit the column-ordered fmaf chain of mhlik_loglik_f32's kernel;
 *                            bound by the f32 matrix rate.
 *   MHLIK_BATCHED_F32_SPLIT  (default) every float is split exactly into three bfloat16 pieces and the six
 *                            leading piece products run on the 16x faster bf16 MFMA; the dropped products are
 *                            below 2^-23 of |x.theta|, i.e. the dots are as close to the exact dot as the fmaf
 *                            chain's, but not the same bits.
 * The default can also be chosen with the environment variable MHLIK_BATCHED_F32=exact|split. */
#define MHLIK_BATCHED_F32_EXACT 0
#define MHLIK_BATCHED_F32_SPLIT 1
int mhlik_set_batched_f32_mode(mhlik_ctx *ctx, int32_t mode);

/* How mhlik_loglik_batched_f64 turns the row dots into -log(1 + e^x) (no counterpart in the reference):
 *   MHLIK_BATCHED_F64_FAST   (default) product form: a lane adds max(x, 0) and multiplies the factors 1 + e^-|x| of its
 *                            32 rows, a chain's 16 lane products of a tile are multiplied and ONE log is taken per
 *                            (tile, chain) (mhlik_softplus.h) -- 19 FP64 instructions per element instead of 45,
 *                            within 2e-15 relative on a lane's sum, rounding errors only (no bias);
 *   MHLIK_BATCHED_F64_EXACT  -log(1 + e^x) per element, 6e-16 each, by full-length series.
 * Both meet the 1e-10 contract of the summed log-likelihood with five orders of magnitude to spare; the row dots are
 * the same.  The default can also be chosen with the environment variable MHLIK_BATCHED_F64=exact|fast. */
#define MHLIK_BATCHED_F64_EXACT 0
#define MHLIK_BATCHED_F64_FAST 1
int mhlik_set_batched_f64_mode(mhlik_ctx *ctx, int32_t mode);

/* Per-row dot products x_i.theta of this shard's rows, to a host array of
 * local-row-count doubles: the reference's data.mvout / d.cuLhood
 * (mcmc_cpu.c:436, mcmc_gpu.cu:233).  Test / debug entry point. */
int mhlik_rowdots_f64(mhlik_ctx *ctx, const double *theta, double *out);

/* ---- mixed precision (src/mcmc_mp/mcmc_mp.cu) ------------------------- */

typedef struct mhlik_mp_state {
  int32_t block_rows;    /* mp_str.dataBlockSz */
  int32_t blocks;        /* mp_str.dataBlocks (global) */
  int32_t resamp_factor; /* mp_str.ResampFactor */
  int32_t resamp_choose; /* gsl_rng_uniform_int(r, ResampFactor), mcmc_mp.cu:504 */
  uint64_t u_seed;       /* seed of the per-block uniforms (reference: cuRAND, 1234) */
  uint64_t u_call;       /* call counter: selects the uniforms of this call */
  double bound_eps;      /* lhoodBound / `e` */
} mhlik_mp_state;

/* (Re)initialise the bright/dark indicators: every block bright (mcmc_mp.cu:323). */
int mhlik_mp_reset(mhlik_ctx *ctx, int32_t block_rows, int32_t blocks);

/* One mixed-precision likelihood evaluation (mp_likelihood, mcmc_mp.cu:475-552):
 * float dots for all rows, double dots for bright / resampled blocks, z
 * resampling.  Returns red_d + red_f and the new bright / dark block counts. */
int mhlik_mp_loglik(mhlik_ctx *ctx, const mhlik_mp_state *st, const double *theta, double *out,
                    int32_t *bright_blocks, int32_t *dark_blocks);

/* likelihood_bound (mcmc_mp.cu:746-821): sum of double sigmoids and the
 * extreme of epsilon_i = sigma_D - sigma_S over all rows. */
int mhlik_mp_bound(mhlik_ctx *ctx, const double *theta, double *sum_lb, double *epsilon);

/* Row shards (no counterpart: the reference is single-GPU, mcmc_mp.cu:286).
 * mhlik_mp_loglik / mhlik_mp_bound work on contexts with several segments and
 * world > 1 when config.segment_align is a multiple of 1024: every rank calls
 * with the same arguments, each evaluates the data blocks of its own rows,
 * the per-segment sums (double sum, float sum, bright count) are all-reduced
 * like the DP path's and added in segment order -- the same bits for every
 * world.  Without a communicator (other transports, tests) the call returns
 * this rank's share and mhlik_mp_last_slots hands out its 3*segments slots
 * [double sums | float sums | bright counts] (mhlik_mp_bound: [sums | minima])
 * to be summed over the ranks by the caller. */
int mhlik_mp_last_slots(const mhlik_ctx *ctx, double *slots);

/* ---- mixed precision, K lock-step chains (BASELINE.json config 5) -------
 *
 * No counterpart in the reference (one chain per mp_likelihood call,
 * src/mcmc_mp/mcmc_mp.cu:475-552).  Chain k is exactly the single-chain
 * definition above with its own indicators, its own re-examination phase
 * resamp_choose[k] and its own per-block uniforms u_seed[k]; all chains share
 * block_rows, blocks, resamp_factor, the call counter and the bound.  The
 * float dots of ALL rows and chains (the reference's cublasSgemv over the
 * whole matrix, :515) run as one X.Theta pass on the matrix cores; double
 * arithmetic touches only the (block, chain) pairs that are bright or
 * re-examined in this call.  Needs both matrices and, as the single-chain
 * path, segments=1 or config.segment_align a multiple of 1024 (row shards:
 * see mhlik_mp_last_slots_batched below). */
typedef struct mhlik_mp_batch {
  int32_t block_rows;           /* mp_str.dataBlockSz */
  int32_t blocks;               /* mp_str.dataBlocks */
  int32_t resamp_factor;        /* mp_str.ResampFactor */
  int32_t K;                    /* chains, 1..MHLIK_MAX_BATCH */
  const int32_t *resamp_choose; /* host [K]: gsl_rng_uniform_int(r_k, ResampFactor), mcmc_mp.cu:504 */
  const uint64_t *u_seed;       /* host [K]: seed of chain k's per-block uniforms */
  uint64_t u_call;              /* call counter */
  double bound_eps;             /* lhoodBound */
} mhlik_mp_batch;

/* (Re)initialise K sets of indicators: every block bright for every chain. */
int mhlik_mp_reset_batched(mhlik_ctx *ctx, int32_t block_rows, int32_t blocks, int32_t K);

/* One mixed-precision evaluation of K proposals.  Theta is host row-major
 * [K][n_cols]; out[k] = red_d + red_f of chain k; bright_blocks / dark_blocks
 * (host [K], may be NULL) are the counts after resampling. */
int mhlik_mp_loglik_batched(mhlik_ctx *ctx, const mhlik_mp_batch *st, const double *Theta, double *out,
                            int32_t *bright_blocks, int32_t *dark_blocks);

/* The same evaluation in two halves (see mhlik_loglik_batched_begin_f64): _begin copies st and Theta and queues
 * the kernels, _end waits and writes the results.  No other evaluation on the context in between. */
int mhlik_mp_loglik_batched_begin(mhlik_ctx *ctx, const mhlik_mp_batch *st, const double *Theta);
int mhlik_mp_loglik_batched_end(mhlik_ctx *ctx, double *out, int32_t *bright_blocks, int32_t *dark_blocks);

/* Row shards for the K-chain mixed-precision calls (no counterpart in the reference): as for mhlik_mp_loglik, a
 * context with several segments (config.segment_align a multiple of 1024) and world > 1 evaluates the data blocks
 * of its own rows; per chain and segment the three sums (double sum over blocks bright at entry, float sum over dark
 * rows, bright count) are all-reduced in the stream and added in segment order on the device -- bit-identical for
 * every world that divides `segments` (with segments = 1 the bits of the unsharded path).  The indicators a rank
 * keeps (mhlik_mp_indicators_batched) are those of its own blocks; the others stay at their reset value.  Without
 * a communicator the call returns this rank's share, and mhlik_mp_last_slots_batched hands out the 4*K*segments
 * slots of the last evaluation, slots[(k*4 + i)*segments + g], i = 0 double sum, 1 float sum, 2 bright count
 * (3 unused), to be summed over the ranks by the caller. */
int mhlik_mp_last_slots_batched(mhlik_ctx *ctx, int32_t K, double *slots);

/* Copy the indicators to the host as [K][blocks] (tests / diagnostics). */
int mhlik_mp_indicators_batched(mhlik_ctx *ctx, int8_t *z);

/* ---- MH chain kept on the device -----------
 *
 * Replaces the per-iteration host round trip of the reference's loop
 * (metropolis_gpu, src/mcmc_gpu/mcmc_gpu.cu:101-128: H2D theta :231, D2H
 * partials :205, CPU tail sum :207-209, prior + ratio :639-658, accept test
 * :112-114; float twin src/mcmc_gpu_sp/mcmc_gpu_sp.cu:101-128).  The random
 * numbers an iteration consumes do not depend on the chain (d Gaussians, then
 * one uniform), so the C sampler draws them for a batch of iterations in the
 * reference's order and the device runs the whole batch back to back:
 *   proposed = current + increments[i]
 *   L = likelihood(proposed);  pposterior = sum_j log N(proposed_j; 0, prior_sd) + L
 *   accept iff log_u[i] <= pposterior - cposterior
 * With world > 1 every rank calls with the same numbers; every rank ends with
 * the same bits.  The _f32 twins keep state, prior sum, posterior and accept
 * test in float like sp_sampler does. */
typedef struct mhlik_chain_batch_f64 {
  int64_t n;                /* iterations in this batch */
  const double *increments; /* host [n][n_cols] */
  const double *log_u;      /* host [n] */
  double *samples;          /* host out [n][n_cols]: state after each iteration (NULL: skip) */
  double *lik;              /* host out [n]: L(proposed) (NULL: skip) */
  double *ratio;            /* host out [n]: pposterior - cposterior (NULL: skip) */
  uint8_t *accepted;        /* host out [n]: 1 = accepted (NULL: skip) */
  int64_t n_accepted;       /* out: accepted proposals of this batch */
  float kernel_ms;          /* out: mean device time of the sampled likelihood kernels (config.timing) */
  int32_t kernel_samples;   /* out: how many launches were bracketed with events */
} mhlik_chain_batch_f64;

typedef struct mhlik_chain_batch_f32 {
  int64_t n;
  const float *increments;
  const float *log_u;
  float *samples;
  double *lik;
  double *ratio;
  uint8_t *accepted;
  int64_t n_accepted;
  float kernel_ms;
  int32_t kernel_samples;
} mhlik_chain_batch_f32;

/* Start a chain at theta0: one likelihood pass, cposterior = prior + L(theta0)
 * (start of burn_in_metropolis_gpu, mcmc_gpu.cu:141-147). */
int mhlik_chain_begin_f64(mhlik_ctx *ctx, const double *theta0, double prior_sd, double *cposterior);
int mhlik_chain_begin_f32(mhlik_ctx *ctx, const float *theta0, double prior_sd, float *cposterior);
/* Continue from a state the caller holds (no likelihood pass, and no device
 * operation: the state is uploaded with the next batch).  begin and resume
 * also make sure that the buffers of a following run of up to 256 iterations
 * exist, so that such a run allocates nothing; buffers never shrink. */
int mhlik_chain_resume_f64(mhlik_ctx *ctx, const double *current, double cposterior, double prior_sd);
int mhlik_chain_resume_f32(mhlik_ctx *ctx, const float *current, float cposterior, double prior_sd);
/* Run batch->n iterations from the device-side state; blocks until the
 * outputs are on the host. */
int mhlik_chain_run_f64(mhlik_ctx *ctx, mhlik_chain_batch_f64 *batch);
int mhlik_chain_run_f32(mhlik_ctx *ctx, mhlik_chain_batch_f32 *batch);
/* The same in two halves, so that the caller can draw the next batch's random
 * numbers and file the previous batch's rows while the device runs (no
 * counterpart in the reference, whose loop is synchronous).  _submit copies
 * increments and log_u (1 <= n <= MHLIK_CHAIN_MAX_SUBMIT; both arrays may be
 * reused on return), queues the batch behind whatever the context's stream
 * holds and returns; at most two batches may be in flight.  _collect waits for
 * the OLDEST batch in flight and fills that batch's outputs: pass the same n;
 * samples may only be asked for if the submitted batch had a samples pointer.
 * mhlik_chain_begin / _resume / _run / _state refuse while a batch is in flight. */
#define MHLIK_CHAIN_MAX_SUBMIT 4096
int mhlik_chain_submit_f64(mhlik_ctx *ctx, const mhlik_chain_batch_f64 *batch);
int mhlik_chain_submit_f32(mhlik_ctx *ctx, const mhlik_chain_batch_f32 *batch);
int mhlik_chain_collect_f64(mhlik_ctx *ctx, mhlik_chain_batch_f64 *batch);
int mhlik_chain_collect_f32(mhlik_ctx *ctx, mhlik_chain_batch_f32 *batch);
/* Read the device-side state back. */
int mhlik_chain_state_f64(mhlik_ctx *ctx, double *current, double *cposterior);
int mhlik_chain_state_f32(mhlik_ctx *ctx, float *current, float *cposterior);

/* ---- K lock-step MH chains kept on the device (SURVEY.md section 8 row f3, K chains) ----
 *
 * The K-chain twin of mhlik_chain_*: no counterpart in the reference (one
 * chain, src/mcmc_gpu/mcmc_gpu.cu:101-128).  Chain k is the reference's chain
 * driven by its own generator: the host draws, per iteration and per chain,
 * d Gaussians and one uniform in that order (mixed precision: d Gaussians, the
 * re-examination phase gsl_rng_uniform_int(r_k, ResampFactor), one uniform;
 * src/mcmc_mp/mcmc_mp.cu:389,504,400), hands a whole batch over, and the device
 * runs   proposed_k = current_k + increments[i][k]
 *        L_k = batched likelihood (X.Theta on the matrix cores)
 *        pposterior_k = sum_j log N(proposed_kj; 0, prior_sd) + L_k
 *        accept iff log_u[i][k] <= pposterior_k - cposterior_k
 *        (MHLIK_KCHAIN_MP: iff u[i][k] <= exp(pposterior_k - cposterior_k), mcmc_mp.cu:402)
 * for i = 0..n-1 back to back with no host round trip.  State is double in
 * every mode (MHLIK_KCHAIN_F32 rounds the proposals to float for the pass, like
 * samplers/multichain.c does).  With world > 1 every rank calls with the same
 * numbers (all three modes). */
enum { MHLIK_KCHAIN_F64 = 1, MHLIK_KCHAIN_F32 = 2, MHLIK_KCHAIN_MP = 3 };

typedef struct mhlik_kchain_mp {
  int32_t resamp_factor;        /* mp_str.ResampFactor */
  int32_t reserved;
  const int32_t *resamp_choose; /* host: the phases; [K] for mhlik_kchain_begin, [n][K] for mhlik_kchain_run */
  const uint64_t *u_seed;       /* host [K]: seed of chain k's per-block uniforms */
  uint64_t u_call;              /* call counter of the FIRST evaluation of this call (then +1 per iteration) */
  double bound_eps;             /* lhoodBound */
} mhlik_kchain_mp;

typedef struct mhlik_kchain_batch {
  int64_t n;                 /* iterations in this batch */
  const double *increments;  /* host [n][K][n_cols] */
  const double *log_u;       /* host [n][K]: log(u); MHLIK_KCHAIN_MP: u itself */
  const mhlik_kchain_mp *mp; /* MHLIK_KCHAIN_MP only (block geometry: mhlik_mp_reset_batched) */
  double *samples;           /* host out [n][K][n_cols]: state after each iteration (NULL: skip) */
  double *lik;               /* host out [n][K]: L(proposed) (NULL: skip) */
  double *ratio;             /* host out [n][K]: pposterior - cposterior (MP: its exp) (NULL: skip) */
  uint8_t *accepted;         /* host out [n][K]: 1 = accepted (NULL: skip) */
  int32_t *bright_blocks;    /* host out [K], MP: bright blocks after the last iteration (NULL: skip) */
  int32_t *dark_blocks;      /* host out [K] (NULL: skip) */
  float kernel_ms;           /* out: mean device time of the sampled batched passes (config.timing) */
  int32_t kernel_samples;    /* out */
} mhlik_kchain_batch;

/* Page-locked host memory for the batch arrays (increments, samples, ...):
 * copies to and from it run at the full PCIe rate and never stage through a
 * bounce buffer.  Optional -- every entry point takes ordinary memory too. */
int mhlik_host_alloc(size_t bytes, void **out);
void mhlik_host_free(void *p);

/* Start K chains at theta0 (host [K][n_cols]): one batched pass,
 * cposterior[k] = prior + L(theta0_k) (host [K], may be NULL).  mp: the phases,
 * seeds, counter and bound of that first evaluation (MHLIK_KCHAIN_MP only). */
int mhlik_kchain_begin(mhlik_ctx *ctx, int32_t mode, int32_t K, const double *theta0, double prior_sd,
                       const mhlik_kchain_mp *mp, double *cposterior);
/* Continue from states the caller holds (no likelihood pass). */
int mhlik_kchain_resume(mhlik_ctx *ctx, int32_t mode, int32_t K, const double *current, const double *cposterior,
                        double prior_sd);
/* Run batch->n lock-step iterations; blocks until the outputs are on the host. */
int mhlik_kchain_run(mhlik_ctx *ctx, mhlik_kchain_batch *batch);
/* Read the device-side states back: current [K][n_cols], cposterior [K]. */
int mhlik_kchain_state(mhlik_ctx *ctx, double *current, double *cposterior);

/* ---- chain statistics on the device ---------
 *
 * circular_autocorrelation (src/util/processing_util.c:206-260, float twin
 * :262-316) as tune_ess_gpu / tune_ess_sp call it after every trial chain
 * (src/mcmc_gpu/mcmc_gpu.cu:260-377): samples is a host array [n][d]
 * (row-major: one chain state per row); out_v[k], k < lags, gets
 * prod_i acc_i(k) / (var_i * (n - k)); *sum (may be NULL) the sum of out_v[1..].
 * Needs no design matrix; uses the context's device and stream. */
int mhlik_autocorrelation_f64(mhlik_ctx *ctx, const double *samples, int64_t n, int32_t d, int32_t lags,
                              double *out_v, double *sum);
int mhlik_autocorrelation_f32(mhlik_ctx *ctx, const float *samples, int64_t n, int32_t d, int32_t lags,
                              float *out_v, float *sum);

/* ---- multi-GPU (no counterpart: the reference is single-GPU, mcmc_gpu.cu:28) */

/* Rank 0 makes an id, the host layer ships the bytes to every rank (e.g. a
 * torch.distributed broadcast), every rank calls mhlik_comm_init.  After it,
 * each evaluation all-reduces the per-segment partial sums over RCCL/xGMI. */
int mhlik_comm_make_id(uint8_t id[MHLIK_COMM_ID_BYTES]);
int mhlik_comm_init(mhlik_ctx *ctx, const uint8_t id[MHLIK_COMM_ID_BYTES]);

/* Tear the communicator down again (evaluations go back to this rank's rows
 * only) / number of ranks of the live communicator, 0 when there is none. */
int mhlik_comm_destroy(mhlik_ctx *ctx);
int mhlik_comm_ranks(const mhlik_ctx *ctx, int32_t *ranks);

/* Without RCCL (tests, other transports): get this rank's per-segment sums
 * of the LAST evaluation / set the combined ones.  seg_sums has
 * config.segments entries, zero where this rank holds no rows. */
int mhlik_last_segment_sums(const mhlik_ctx *ctx, double *seg_sums);

/* ---- instrumentation --------------------------------------------------- */

/* Device time of the last likelihood kernel in ms (config.timing != 0) and
 * the algorithmic bytes it read (local rows * n_cols * sizeof(T)). */
int mhlik_last_kernel_ms(const mhlik_ctx *ctx, float *ms, int64_t *bytes);

/* Which instantiation of the streaming kernel the last evaluation launched,
 * spelled as rocprofv3 prints it ("lik_tile_kernel<double, 1, 16, true, false,
 * false>"; "" before the first launch).  bench.py matches it against the
 * kernel named in the committed counter passes. */
const char *mhlik_last_kernel_name(const mhlik_ctx *ctx);

/* Geometry actually used (for DESIGN.md / bench reporting). */
typedef struct mhlik_info {
  int64_t n_rows; /* global rows, columns and precision mask of the config */
  int32_t n_cols;
  int32_t precision;
  int32_t rank, world;
  int64_t local_rows;
  int64_t tiles_f64, tiles_f32;
  int32_t tile_rows_f64, tile_rows_f32;
  int32_t layout, variant;
  int32_t compute_units;
  int32_t reserved;
  int64_t hbm_bytes; /* device bytes held by the context */
} mhlik_info;
int mhlik_get_info(const mhlik_ctx *ctx, mhlik_info *info);

/* Library version string. */
const char *mhlik_version(void);

#ifdef __cplusplus
}
#endif

#endif /* MHLIK_H */
