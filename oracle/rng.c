/*
 * oracle/rng.c -- the oracle's OWN random numbers behind GSL's names.
 *
 * TEST INFRASTRUCTURE ONLY.  Until round 3 liboracle.so compiled the product's
 * mcmc_amd/csrc/gslcompat/gslcompat.c, so a sampler trace of the product was
 * checked against proposals drawn by the very same code.  This file is a
 * second, separately written statement of the same published algorithms; it
 * shares with the product only the declarations of GSL's public API
 * (gsl/gsl_rng.h, gsl/gsl_randist.h: the struct layout a caller sees):
 *   - MT19937 as in Matsumoto & Nishimura's mt19937ar.c (init_genrand /
 *     genrand_int32 with the mag01 table), plus GSL's conventions: seed 0
 *     stands for 4357, uniform = get / 2^32, uniform_int by rejection
 *     (reference call sites: src/util/alloc_util.c:249-277,
 *     src/mcmc_cpu/mcmc_cpu.c:53,58, src/mcmc_mp/mcmc_mp.cu:504);
 *   - the 128-level Gaussian ziggurat (Marsaglia & Tsang 2000 in J. Voss's
 *     layout for GSL's gauss-zig): the tables are BUILT HERE at first use
 *     from the construction (equal-area layers closed at y = 1, R found by
 *     bisection) and rounded to the 12 significant digits GSL prints its
 *     literals with -- not read from the product's zig_tables.h.
 * tests/test_rng.py runs the same known-answer checks on both libraries and
 * compares their streams with each other, with numpy's MT19937 and with a
 * Python restatement of the ziggurat.  GSL itself is absent from the image:
 * "parity unpinned" against a real libgsl.
 */
#include <gsl/gsl_rng.h>
#include <gsl/gsl_randist.h>

#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------- generator */

#define ORC_NN 624
#define ORC_MM 397

struct orc_mt {
  unsigned long v[ORC_NN]; /* 32 significant bits each, as in mt19937ar.c */
  int next;
};

static void orc_seed(void *st, unsigned long int s)
{
  struct orc_mt *g = st;
  if (s == 0) s = 4357; /* GSL: the default seed of the 2002 generator */
  g->v[0] = s & 0xffffffffUL;
  for (int j = 1; j < ORC_NN; j++) {
    const unsigned long prev = g->v[j - 1];
    g->v[j] = (1812433253UL * (prev ^ (prev >> 30)) + (unsigned long)j) & 0xffffffffUL;
  }
  g->next = ORC_NN;
}

static unsigned long int orc_next(void *st)
{
  static const unsigned long mag01[2] = {0x0UL, 0x9908b0dfUL};
  struct orc_mt *g = st;
  if (g->next >= ORC_NN) { /* regenerate the whole block */
    for (int kk = 0; kk < ORC_NN; kk++) {
      const unsigned long y = (g->v[kk] & 0x80000000UL) | (g->v[(kk + 1) % ORC_NN] & 0x7fffffffUL);
      g->v[kk] = g->v[(kk + ORC_MM) % ORC_NN] ^ (y >> 1) ^ mag01[y & 0x1UL];
    }
    g->next = 0;
  }
  unsigned long y = g->v[g->next++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680UL;
  y ^= (y << 15) & 0xefc60000UL;
  y ^= y >> 18;
  return y & 0xffffffffUL;
}

static double orc_next_double(void *st) { return orc_next(st) * (1.0 / 4294967296.0); }

static const gsl_rng_type orc_type = {"mt19937", 0xffffffffUL, 0, sizeof(struct orc_mt), orc_seed, orc_next, orc_next_double};

const gsl_rng_type *gsl_rng_mt19937 = &orc_type;
const gsl_rng_type *gsl_rng_default = &orc_type;
unsigned long int gsl_rng_default_seed = 0;

const gsl_rng_type *gsl_rng_env_setup(void)
{
  const char *t = getenv("GSL_RNG_TYPE"), *s = getenv("GSL_RNG_SEED");
  if (t && *t && strcmp(t, "mt19937")) {
    fprintf(stderr, "oracle rng: GSL_RNG_TYPE=%s unknown (mt19937 only)\n", t);
    exit(EXIT_FAILURE);
  }
  if (s && *s) gsl_rng_default_seed = strtoul(s, NULL, 0);
  return gsl_rng_default;
}

gsl_rng *gsl_rng_alloc(const gsl_rng_type *T)
{
  gsl_rng *r = malloc(sizeof *r);
  void *st = r ? calloc(1, T->size) : NULL;
  if (!st) {
    free(r);
    return NULL;
  }
  r->type = T;
  r->state = st;
  T->set(st, gsl_rng_default_seed);
  return r;
}

void gsl_rng_free(gsl_rng *r)
{
  if (r) {
    free(r->state);
    free(r);
  }
}

void gsl_rng_set(const gsl_rng *r, unsigned long int seed) { r->type->set(r->state, seed); }
unsigned long int gsl_rng_get(const gsl_rng *r) { return r->type->get(r->state); }
double gsl_rng_uniform(const gsl_rng *r) { return r->type->get_double(r->state); }
const char *gsl_rng_name(const gsl_rng *r) { return r->type->name; }

unsigned long int gsl_rng_uniform_int(const gsl_rng *r, unsigned long int n)
{
  const unsigned long span = r->type->max - r->type->min;
  if (n == 0 || n > span) {
    fprintf(stderr, "oracle rng: gsl_rng_uniform_int: n = %lu out of range\n", n);
    return 0;
  }
  const unsigned long bucket = span / n;
  for (;;) {
    const unsigned long k = (r->type->get(r->state) - r->type->min) / bucket;
    if (k < n) return k;
  }
}

/* -------------------------------------------------------------- ziggurat */

static double zg_R, zg_y[128], zg_w[128];
static unsigned long zg_k[128];
static pthread_once_t zg_once = PTHREAD_ONCE_INIT;

static double zg_density(double x) { return exp(-0.5 * x * x); }

/* layers for a trial R: edges x[0..128] and heights y[0..127]; returns y[0] (1 when R closes the stack) */
static double zg_stack(double R, double *x, double *y)
{
  const double area = zg_density(R) * (R + 1.0 / R);
  x[128] = R + 1.0 / R;
  x[127] = R;
  y[127] = zg_density(R);
  for (int i = 126; i >= 0; i--) {
    y[i] = y[i + 1] + area / x[i + 1];
    x[i] = y[i] < 1.0 ? sqrt(-2.0 * log(y[i])) : 0.0;
  }
  return y[0];
}

static double zg_print12(double v) /* the value of v's 12-significant-digit decimal, as a literal would be read */
{
  char buf[40];
  snprintf(buf, sizeof buf, "%.12g", v);
  return strtod(buf, NULL);
}

static void zg_build(void)
{
  double x[129], y[128];
  double lo = 3.4442, hi = 3.4443;
  for (int it = 0; it < 200; it++) {
    const double mid = 0.5 * (lo + hi);
    if (zg_stack(mid, x, y) > 1.0)
      lo = mid;
    else
      hi = mid;
  }
  const double R = 0.5 * (lo + hi);
  zg_stack(R, x, y);
  y[0] = 1.0;
  x[0] = 0.0;
  zg_R = zg_print12(R);
  for (int i = 0; i < 128; i++) {
    zg_y[i] = zg_print12(y[i]);
    zg_k[i] = (unsigned long)(16777216.0 * x[i] / x[i + 1]);
    zg_w[i] = zg_print12(x[i + 1] / 16777216.0);
  }
}

double gsl_ran_gaussian_ziggurat(const gsl_rng *r, double sigma)
{
  pthread_once(&zg_once, zg_build);
  const unsigned long lo = r->type->min, span = r->type->max - lo;
  double x;
  int negative;
  for (;;) {
    unsigned long level, frac;
    if (span >= 0xFFFFFFFFUL) { /* one 32-bit draw: low byte picks the layer, the top 24 bits the abscissa */
      const unsigned long bits = gsl_rng_get(r) - lo;
      level = bits & 0xFF;
      frac = (bits >> 8) & 0xFFFFFF;
    } else if (span >= 0x00FFFFFFUL) {
      level = (gsl_rng_get(r) - lo) & 0xFF;
      frac = (gsl_rng_get(r) - lo) & 0x00FFFFFF;
    } else {
      level = gsl_rng_uniform_int(r, 256);
      frac = gsl_rng_uniform_int(r, 16777216);
    }
    negative = !(level & 0x80);
    level &= 0x7F;
    x = frac * zg_w[level];
    if (frac < zg_k[level]) break; /* strictly inside the layer's rectangle */
    double height;
    if (level < 127) { /* the sliver between two layers: a uniform height between them against the density */
      const double top = zg_y[level], bottom = zg_y[level + 1];
      height = bottom + (top - bottom) * gsl_rng_uniform(r);
    } else { /* beyond R: exponential proposal */
      const double u1 = 1.0 - gsl_rng_uniform(r);
      const double u2 = gsl_rng_uniform(r);
      x = zg_R - log(u1) / zg_R;
      height = exp(-zg_R * (x - 0.5 * zg_R)) * u2;
    }
    if (height < exp(-0.5 * x * x)) break;
  }
  return (negative ? -1 : +1) * sigma * x;
}

double gsl_ran_gaussian_pdf(double x, double sigma)
{
  const double s = fabs(sigma), u = x / s;
  return (1.0 / (sqrt(2.0 * M_PI) * s)) * exp(-u * u / 2.0);
}
