/*
 * mhlik_synth.h -- counter-based synthetic logistic-regression data, usable
 * from host C/C++ and from HIP device code.
 *
 * Distribution of the reference's generator (scr/data_scr/
 * synthetic_generation.py:33-65): feature 0 is the bias (1), the others are
 * U(0,1); beta_j is an integer in [-15, 15), beta_0 = 1; the label is
 * sign(x.beta) with 0 counted as +1; the stored value is -y*x
 * (src/util/data_util.c:33).  The reference draws from numpy's global,
 * unseeded generator; here every value is a pure function of
 * (seed, row, column) so any shard of any N can be produced anywhere.
 */
#ifndef MHLIK_SYNTH_H
#define MHLIK_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MHLIK_HD __host__ __device__ static inline
#else
#define MHLIK_HD static inline
#endif

/* splitmix64 finaliser */
MHLIK_HD uint64_t mhlik_mix64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* raw feature x_ij in [0,1) (j >= 1), 53 random bits; x_i0 = 1 */
MHLIK_HD double mhlik_synth_feature(uint64_t seed, int64_t row, int32_t col)
{
  if (col == 0) return 1.0;
  uint64_t h = mhlik_mix64(mhlik_mix64(seed ^ (0xD1B54A32D192ED03ULL * (uint64_t)(col + 1))) + (uint64_t)row);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

MHLIK_HD int32_t mhlik_synth_beta_j(uint64_t seed, int32_t col)
{
  if (col == 0) return 1;
  uint64_t h = mhlik_mix64(seed * 0x2545F4914F6CDD1DULL + 0x5851F42D4C957F2DULL * (uint64_t)(col + 1));
  return (int32_t)(h % 30u) - 15;
}

/* label of a row: sign of sum_j x_ij * beta_j, products rounded before the
 * adds, columns in order (so host and device agree bit for bit) */
MHLIK_HD int32_t mhlik_synth_label(uint64_t seed, int64_t row, int32_t d)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  double acc = 0.0;
  for (int32_t j = 0; j < d; j++) {
    const double p = mhlik_synth_feature(seed, row, j) * (double)mhlik_synth_beta_j(seed, j);
    acc = acc + p;
  }
  return acc < 0.0 ? -1 : 1;
}

/* uniform in [0,1) for (seed, call, block): the per-data-block uniforms of the
 * mixed-precision path (reference: a cuRAND stream, mcmc_mp.cu:507) */
MHLIK_HD double mhlik_mp_uniform(uint64_t seed, uint64_t call, uint64_t block)
{
  uint64_t h = mhlik_mix64(mhlik_mix64(seed ^ 0xA0761D6478BD642FULL) + call);
  h = mhlik_mix64(h + block);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

#endif
