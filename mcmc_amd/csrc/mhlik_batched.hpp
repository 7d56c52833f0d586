// mhlik_batched.hpp -- K-chain batched likelihood (X.Theta on the matrix cores).
#pragma once
#include "mhlik_kernels.hpp"
