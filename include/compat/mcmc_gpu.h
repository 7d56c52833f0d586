/* include/compat/mcmc_gpu.h -- stands in for the reference's src/mcmc_gpu/mcmc_gpu.h: the sampler / tuner entry points with the
 * reference's signatures (include/mcmc_samplers.h), none of its cublas / cuda_runtime / curand includes and none of its
 * inner functions that carry CUDA types. */
#ifndef MCMC_AMD_COMPAT_MCMC_GPU_H
#define MCMC_AMD_COMPAT_MCMC_GPU_H
#include "resources.h"
#include "../mcmc_samplers.h"
#include "../mcmc_host_util.h"
#endif
