/* include/compat/io.h -- stands in for the reference's src/util/io.h: the same host utilities, exported by
 * libmcmc_amd.so (include/mcmc_host_util.h). */
#ifndef MCMC_AMD_COMPAT_IO_H
#define MCMC_AMD_COMPAT_IO_H
#include "resources.h"
#include "../mcmc_host_util.h"
#include "../mcmc_samplers.h"
#endif
