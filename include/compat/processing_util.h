/* include/compat/processing_util.h -- stands in for the reference's src/util/processing_util.h: get_ess,
 * circular_autocorrelation, calculate_normalised_sample_means (+ _sp twins) are declared in include/mcmc_samplers.h and
 * exported by libmcmc_amd.so.  The CUDA launch-geometry helpers (getBlocksAndThreads, nextPow2, getSizes_mp) have no
 * counterpart: the HIP path picks its own geometry. */
#ifndef MCMC_AMD_COMPAT_PROCESSING_UTIL_H
#define MCMC_AMD_COMPAT_PROCESSING_UTIL_H
#include "resources.h"
#include "../mcmc_samplers.h"
#endif
