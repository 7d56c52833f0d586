/*
 * gslcompat.c -- MT19937 + Gaussian ziggurat behind GSL's API names.
 *
 * Restated from the published algorithms:
 *   - MT19937 (Matsumoto & Nishimura, 2002 initialisation), pinned by the
 *     public known-answer vectors in tests/test_rng.py;
 *   - GSL conventions on top of it: seed 0 means 4357, uniform =
 *     get / 2^32 in [0,1), uniform_int by rejection on get / (range / n);
 *   - Gaussian ziggurat with tables from tools/gen_ziggurat_tables.py
 *     (stream not pinned against a real libgsl: "parity unpinned").
 */
#include <gsl/gsl_rng.h>
#include <gsl/gsl_randist.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zig_tables.h"

/* ---------------------------------------------------------------- MT19937 */

enum { MT_N = 624, MT_M = 397 };

typedef struct {
  uint32_t mt[MT_N];
  int mti;
} mt_state;

static void mt_set(void *vstate, unsigned long int seed)
{
  mt_state *s = (mt_state *)vstate;
  uint32_t x = (uint32_t)(seed & 0xffffffffUL);
  if (seed == 0) x = 4357u;
  s->mt[0] = x;
  for (int i = 1; i < MT_N; i++) {
    x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
    s->mt[i] = x;
  }
  s->mti = MT_N;
}

static inline uint32_t mt_twist(uint32_t hi, uint32_t lo, uint32_t far)
{
  uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
  return far ^ (y >> 1) ^ ((lo & 1u) ? 0x9908b0dfu : 0u);
}

static unsigned long int mt_get(void *vstate)
{
  mt_state *s = (mt_state *)vstate;
  uint32_t *mt = s->mt;
  if (s->mti >= MT_N) {
    int k;
    for (k = 0; k < MT_N - MT_M; k++) mt[k] = mt_twist(mt[k], mt[k + 1], mt[k + MT_M]);
    for (; k < MT_N - 1; k++) mt[k] = mt_twist(mt[k], mt[k + 1], mt[k + (MT_M - MT_N)]);
    mt[MT_N - 1] = mt_twist(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
    s->mti = 0;
  }
  uint32_t k = mt[s->mti++];
  k ^= (k >> 11);
  k ^= (k << 7) & 0x9d2c5680u;
  k ^= (k << 15) & 0xefc60000u;
  k ^= (k >> 18);
  return (unsigned long int)k;
}

static double mt_get_double(void *vstate)
{
  return (double)mt_get(vstate) / 4294967296.0;
}

static const gsl_rng_type mt_type = {
  "mt19937", 0xffffffffUL, 0, sizeof(mt_state), &mt_set, &mt_get, &mt_get_double
};

const gsl_rng_type *gsl_rng_mt19937 = &mt_type;
const gsl_rng_type *gsl_rng_default = &mt_type;
unsigned long int gsl_rng_default_seed = 0;

/* ------------------------------------------------------------ gsl_rng API */

const gsl_rng_type *gsl_rng_env_setup(void)
{
  const char *p = getenv("GSL_RNG_TYPE");
  if (p && *p) {
    if (strcmp(p, "mt19937") != 0) {
      fprintf(stderr, "gslcompat: GSL_RNG_TYPE=%s not available, only mt19937\n", p);
      exit(EXIT_FAILURE);
    }
    fprintf(stderr, "GSL_RNG_TYPE=%s\n", p);
  }
  gsl_rng_default = &mt_type;

  p = getenv("GSL_RNG_SEED");
  if (p && *p) {
    gsl_rng_default_seed = strtoul(p, 0, 0);
    fprintf(stderr, "GSL_RNG_SEED=%lu\n", gsl_rng_default_seed);
  }
  return gsl_rng_default;
}

gsl_rng *gsl_rng_alloc(const gsl_rng_type *T)
{
  gsl_rng *r = (gsl_rng *)malloc(sizeof(gsl_rng));
  if (!r) return NULL;
  r->state = calloc(1, T->size);
  if (!r->state) {
    free(r);
    return NULL;
  }
  r->type = T;
  gsl_rng_set(r, gsl_rng_default_seed);
  return r;
}

void gsl_rng_free(gsl_rng *r)
{
  if (!r) return;
  free(r->state);
  free(r);
}

void gsl_rng_set(const gsl_rng *r, unsigned long int seed) { r->type->set(r->state, seed); }

unsigned long int gsl_rng_get(const gsl_rng *r) { return r->type->get(r->state); }

double gsl_rng_uniform(const gsl_rng *r) { return r->type->get_double(r->state); }

const char *gsl_rng_name(const gsl_rng *r) { return r->type->name; }

unsigned long int gsl_rng_uniform_int(const gsl_rng *r, unsigned long int n)
{
  unsigned long int offset = r->type->min;
  unsigned long int range = r->type->max - offset;
  unsigned long int scale, k;

  if (n > range || n == 0) {
    fprintf(stderr, "gslcompat: gsl_rng_uniform_int: invalid n, either 0 or exceeds maximum value of generator\n");
    return 0;
  }
  scale = range / n;
  do {
    k = (r->type->get(r->state) - offset) / scale;
  } while (k >= n);
  return k;
}

/* -------------------------------------------------------------- Gaussians */

double gsl_ran_gaussian_ziggurat(const gsl_rng *r, double sigma)
{
  unsigned long int i, j;
  int sign;
  double x, y;
  const unsigned long int range = r->type->max - r->type->min;
  const unsigned long int offset = r->type->min;

  for (;;) {
    if (range >= 0xFFFFFFFFUL) {
      unsigned long int k = gsl_rng_get(r) - offset;
      i = (k & 0xFF);
      j = (k >> 8) & 0xFFFFFF;
    } else if (range >= 0x00FFFFFFUL) {
      unsigned long int k1 = gsl_rng_get(r) - offset;
      unsigned long int k2 = gsl_rng_get(r) - offset;
      i = (k1 & 0xFF);
      j = (k2 & 0x00FFFFFF);
    } else {
      i = gsl_rng_uniform_int(r, 256);
      j = gsl_rng_uniform_int(r, 16777216);
    }

    sign = (i & 0x80) ? +1 : -1;
    i &= 0x7f;

    x = j * zig_wtab[i];

    if (j < zig_ktab[i]) break; /* inside the rectangle: no floating-point test */

    if (i < 127) {
      /* wedge between level i and i+1 */
      double y0 = zig_ytab[i], y1 = zig_ytab[i + 1];
      double U1 = gsl_rng_uniform(r);
      y = y1 + (y0 - y1) * U1;
    } else {
      /* tail beyond R: exponential envelope */
      double U1 = 1.0 - gsl_rng_uniform(r);
      double U2 = gsl_rng_uniform(r);
      x = ZIG_PARAM_R - log(U1) / ZIG_PARAM_R;
      y = exp(-ZIG_PARAM_R * (x - 0.5 * ZIG_PARAM_R)) * U2;
    }

    if (y < exp(-0.5 * x * x)) break;
  }

  return sign * sigma * x;
}

double gsl_ran_gaussian_pdf(double x, double sigma)
{
  double u = x / fabs(sigma);
  double p = (1.0 / (sqrt(2.0 * M_PI) * fabs(sigma))) * exp(-u * u / 2.0);
  return p;
}
