/*
 * tests/doubles/mhlik_double.c -- TEST DOUBLE of the few mhlik_* entry points
 * that samplers/multichain.c and samplers/mh_core.c call.  It exists for one
 * purpose: to run multichain.c's chunk pipeline (kc_run: helper thread that
 * draws chunk i+1 and files chunk i-1 while "the device" runs chunk i, the
 * buffer cache behind a mutex) under -fsanitize=thread / address on a machine
 * without a GPU.  It is never linked into the product and computes no
 * likelihood: mhlik_kchain_run copies the caller's buffers through the "device"
 * side (reads every increment and uniform, writes every output row) and
 * accepts a proposal when log u < -1.  See tests/test_host_sanitizers.py.
 */
#include "mhlik.h"

#include <stdlib.h>
#include <string.h>

struct mhlik_ctx {
  int64_t n_rows;
  int32_t n_cols;
  int K, mode;
  double *cur; /* [K][d] */
};

const char *mhlik_last_error(const mhlik_ctx *c) { (void)c; return "test double"; }

int mhlik_create(const mhlik_config *cfg, mhlik_ctx **out)
{
  mhlik_ctx *c = (mhlik_ctx *)calloc(1, sizeof *c);
  c->n_rows = cfg->n_rows;
  c->n_cols = cfg->n_cols;
  *out = c;
  return MHLIK_OK;
}
void mhlik_destroy(mhlik_ctx *c)
{
  if (!c) return;
  free(c->cur);
  free(c);
}
int mhlik_get_info(const mhlik_ctx *c, mhlik_info *info)
{
  memset(info, 0, sizeof *info);
  info->n_rows = c->n_rows;
  info->n_cols = c->n_cols;
  info->precision = MHLIK_F64 | MHLIK_F32;
  info->world = 1;
  return MHLIK_OK;
}
int mhlik_host_alloc(size_t bytes, void **out) { *out = malloc(bytes ? bytes : 1); return *out ? MHLIK_OK : MHLIK_ENOMEM; }
void mhlik_host_free(void *p) { free(p); }
int mhlik_last_kernel_ms(const mhlik_ctx *c, float *ms, int64_t *bytes) { (void)c; if (ms) *ms = 0; if (bytes) *bytes = 0; return MHLIK_OK; }

static int kc_state(mhlik_ctx *c, int mode, int K, const double *cur)
{
  free(c->cur);
  c->cur = (double *)malloc(sizeof(double) * (size_t)K * c->n_cols);
  memcpy(c->cur, cur, sizeof(double) * (size_t)K * c->n_cols);
  c->K = K;
  c->mode = mode;
  return MHLIK_OK;
}
int mhlik_kchain_begin(mhlik_ctx *c, int32_t mode, int32_t K, const double *theta0, double prior_sd, const mhlik_kchain_mp *mp,
                       double *cposterior)
{
  (void)prior_sd; (void)mp;
  if (cposterior) memset(cposterior, 0, sizeof(double) * (size_t)K);
  return kc_state(c, mode, K, theta0);
}
int mhlik_kchain_resume(mhlik_ctx *c, int32_t mode, int32_t K, const double *current, const double *cposterior, double prior_sd)
{
  (void)cposterior; (void)prior_sd;
  return kc_state(c, mode, K, current);
}
int mhlik_kchain_run(mhlik_ctx *c, mhlik_kchain_batch *b)
{
  const int K = c->K, d = c->n_cols;
  for (int64_t it = 0; it < b->n; it++)
    for (int k = 0; k < K; k++) {
      const double lu = b->log_u[it * K + k];
      const int acc = c->mode == MHLIK_KCHAIN_MP ? lu < 0.3 : lu < -1.0;
      double *cur = c->cur + (size_t)k * d;
      const double *inc = b->increments + ((size_t)it * K + k) * d;
      if (b->mp && (b->mp->resamp_choose[it * K + k] < 0 || b->mp->resamp_choose[it * K + k] >= b->mp->resamp_factor)) return MHLIK_EINVAL;
      if (acc)
        for (int j = 0; j < d; j++) cur[j] += inc[j];
      if (b->samples) memcpy(b->samples + ((size_t)it * K + k) * d, cur, sizeof(double) * d);
      if (b->accepted) b->accepted[it * K + k] = (uint8_t)acc;
      if (b->lik) b->lik[it * K + k] = -1.0;
      if (b->ratio) b->ratio[it * K + k] = lu;
    }
  if (b->bright_blocks) for (int k = 0; k < K; k++) b->bright_blocks[k] = 1;
  if (b->dark_blocks) for (int k = 0; k < K; k++) b->dark_blocks[k] = 2;
  return MHLIK_OK;
}
int mhlik_mp_reset_batched(mhlik_ctx *c, int32_t block_rows, int32_t blocks, int32_t K) { (void)c; (void)block_rows; (void)blocks; (void)K; return MHLIK_OK; }

/* referenced by the same objects, never reached by the test */
#define UNUSED_ENTRY(name, ...) int name(__VA_ARGS__) { abort(); }
UNUSED_ENTRY(mhlik_upload_f64, mhlik_ctx *c, const double *X, int64_t ld, int64_t row0)
UNUSED_ENTRY(mhlik_upload_f32, mhlik_ctx *c, const float *X, int64_t ld, int64_t row0)
UNUSED_ENTRY(mhlik_chain_resume_f64, mhlik_ctx *c, const double *cur, double cp, double sd)
UNUSED_ENTRY(mhlik_chain_state_f64, mhlik_ctx *c, double *cur, double *cp)
UNUSED_ENTRY(mhlik_chain_submit_f64, mhlik_ctx *c, const mhlik_chain_batch_f64 *b)
UNUSED_ENTRY(mhlik_chain_collect_f64, mhlik_ctx *c, mhlik_chain_batch_f64 *b)
UNUSED_ENTRY(mhlik_loglik_batched_begin_f64, mhlik_ctx *c, const double *T, int32_t K)
UNUSED_ENTRY(mhlik_loglik_batched_begin_f32, mhlik_ctx *c, const float *T, int32_t K)
UNUSED_ENTRY(mhlik_loglik_batched_end, mhlik_ctx *c, double *out)
UNUSED_ENTRY(mhlik_mp_loglik_batched_begin, mhlik_ctx *c, const mhlik_mp_batch *st, const double *T)
UNUSED_ENTRY(mhlik_mp_loglik_batched_end, mhlik_ctx *c, double *out, int32_t *nb, int32_t *nd)
