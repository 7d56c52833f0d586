/*
 * gslcompat: the slice of GSL's random-number API that the MH samplers use.
 *
 * GSL is not present in the build image, so mcmc_amd ships its own MT19937
 * generator behind GSL's names and struct layout (gsl_rng / gsl_rng_type have
 * the same members in the same order as GSL 2.x, so a generator allocated by
 * a real libgsl can be handed to the samplers as well).  Reference call
 * sites: src/util/alloc_util.c:249-277 (init_rng, init_rng_seed, free_rng),
 * src/mcmc_cpu/mcmc_cpu.c:53,58 (ziggurat draw, uniform draw),
 * src/mcmc_mp/mcmc_mp.cu:504 (gsl_rng_uniform_int).
 */
#ifndef MCMC_AMD_GSLCOMPAT_RNG_H
#define MCMC_AMD_GSLCOMPAT_RNG_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  const char *name;
  unsigned long int max;
  unsigned long int min;
  size_t size;
  void (*set)(void *state, unsigned long int seed);
  unsigned long int (*get)(void *state);
  double (*get_double)(void *state);
} gsl_rng_type;

typedef struct {
  const gsl_rng_type *type;
  void *state;
} gsl_rng;

extern const gsl_rng_type *gsl_rng_mt19937;
extern const gsl_rng_type *gsl_rng_default;
extern unsigned long int gsl_rng_default_seed;

/* Reads GSL_RNG_TYPE / GSL_RNG_SEED like GSL does; only mt19937 is known. */
const gsl_rng_type *gsl_rng_env_setup(void);

gsl_rng *gsl_rng_alloc(const gsl_rng_type *T);
void gsl_rng_free(gsl_rng *r);
void gsl_rng_set(const gsl_rng *r, unsigned long int seed);
unsigned long int gsl_rng_get(const gsl_rng *r);
double gsl_rng_uniform(const gsl_rng *r);
unsigned long int gsl_rng_uniform_int(const gsl_rng *r, unsigned long int n);
const char *gsl_rng_name(const gsl_rng *r);

#ifdef __cplusplus
}
#endif

#endif
