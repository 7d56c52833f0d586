// mhlik_stats.hpp -- circular autocorrelation of a chain on the device
// (SURVEY.md section 8 row f4).
//
// What tune_ess_gpu / tune_ess_sp evaluate after every trial chain
// (src/util/processing_util.c:206-260 circular_autocorrelation, :318-392 mean
// and variance, :394-403 get_ess; callers src/mcmc_gpu/mcmc_gpu.cu:260-377):
//   mean_i, var_i over the n samples of component i
//   acc_i(k) = sum_s (x[s][i] - mean_i) * (x[(s+k) mod n][i] - mean_i)
//   out[k]   = prod_i acc_i(k) / (var_i * (n - k)),   k = 0 .. lags-1
// n * d * lags multiply-adds (5000 x 64 x 500 = 1.6e8) that the reference does
// in one scalar loop on the host between trial chains.  Here: one workgroup
// per component for the moments, one workgroup per lag for acc / out; sums are
// taken in double in a fixed order (tolerance vs the sequential host loop:
// summation order only).  The wrap is at (s+k) >= n, see chain_stats.c.
#pragma once

#include "mhlik_kernels.hpp"

namespace mhlik {

// grid = d; m: [n][d] row-major
template <typename T>
__global__ __launch_bounds__(kBlock) void stats_moments_kernel(const T *__restrict__ m, int n, int d,
                                                               double *__restrict__ mean, double *__restrict__ var)
{
  __shared__ double lds[kBlock / kWave];
  __shared__ double mu_s;
  const int i = blockIdx.x;
  double s = 0.0;
  for (int r = threadIdx.x; r < n; r += kBlock) s += (double)m[(int64_t)r * d + i];
  s = block_sum(s, lds);
  if (threadIdx.x == 0) mu_s = s / n;
  __syncthreads();
  const double mu = mu_s;
  double v = 0.0;
  for (int r = threadIdx.x; r < n; r += kBlock) {
    const double c = (double)m[(int64_t)r * d + i] - mu;
    v += c * c;
  }
  v = block_sum(v, lds);
  if (threadIdx.x == 0) {
    mean[i] = mu;
    var[i] = v / n;
  }
}

// grid = lags; dpad = power of two >= min(d, 256): thread t owns component
// (t mod dpad) [+ chunk] and sample lane t / dpad
template <typename T>
__global__ __launch_bounds__(kBlock) void stats_autocorr_kernel(const T *__restrict__ m, int n, int d, int dpad,
                                                                const double *__restrict__ mean,
                                                                const double *__restrict__ var,
                                                                double *__restrict__ out)
{
  __shared__ double accs[kBlock];
  __shared__ double factor[kBlock];
  const int k = blockIdx.x;
  const int ci = threadIdx.x & (dpad - 1), sl = threadIdx.x / dpad, nsl = kBlock / dpad;
  double prod = 1.0; // thread 0 only
  for (int i0 = 0; i0 < d; i0 += dpad) {
    const int i = i0 + ci;
    double a = 0.0;
    if (i < d) {
      const double mu = mean[i];
      for (int s = sl; s < n; s += nsl) {
        int sh = s + k;
        if (sh >= n) sh -= n;
        a += ((double)m[(int64_t)s * d + i] - mu) * ((double)m[(int64_t)sh * d + i] - mu);
      }
    }
    accs[threadIdx.x] = a;
    __syncthreads();
    if ((int)threadIdx.x < dpad && i0 + (int)threadIdx.x < d) {
      double t = 0.0;
      for (int q = 0; q < nsl; q++) t += accs[threadIdx.x + q * dpad];
      factor[threadIdx.x] = t / (var[i0 + threadIdx.x] * (double)(n - k));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int m_here = d - i0 < dpad ? d - i0 : dpad;
      for (int q = 0; q < m_here; q++) prod *= factor[q]; // component order, like the host loop
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[k] = prod;
}

} // namespace mhlik
