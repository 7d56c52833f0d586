/* include/compat/alloc_util.h -- stands in for the reference's src/util/alloc_util.h: the same host utilities, exported by
 * libmcmc_amd.so (include/mcmc_host_util.h). */
#ifndef MCMC_AMD_COMPAT_ALLOC_UTIL_H
#define MCMC_AMD_COMPAT_ALLOC_UTIL_H
#include "resources.h"
#include "../mcmc_host_util.h"
#include "../mcmc_samplers.h"
#endif
