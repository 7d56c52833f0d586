// mhlik_mp.hpp -- mixed-precision (FlyMC-style bright/dark) likelihood kernels.
#pragma once
#include "mhlik_kernels.hpp"
