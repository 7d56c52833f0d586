/* include/compat/resources.h -- stands in for the reference's src/util/resources.h (:1-220): the same structs and
 * enums (include/mcmc_types.h), without the <gsl/gsl_cblas.h> dependency on a CUDA/GSL build.  See INTEGRATION.md
 * section 2 for where this file has to sit (the reference includes it with quotes). */
#ifndef MCMC_AMD_COMPAT_RESOURCES_H
#define MCMC_AMD_COMPAT_RESOURCES_H
#include <ctype.h>
#include <math.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../mcmc_types.h"
#endif
