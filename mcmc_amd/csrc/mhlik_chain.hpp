// mhlik_chain.hpp -- the MH chain kept on the device between iterations.
//
// The reference's loop (src/mcmc_gpu/mcmc_gpu.cu:101-128, float twin
// src/mcmc_gpu_sp/mcmc_gpu_sp.cu:101-128) goes back to the host after every
// likelihood: D2H of the partials, CPU tail sum, prior, accept test, next
// proposal, H2D of theta.  Which random numbers an iteration consumes does not
// depend on the chain (d Gaussians, then one uniform), so the host can draw
// them for a whole batch of iterations in the reference's order, hand them
// over once, and let the device run the batch without a host round trip:
//
//   per iteration:  lik_tile_kernel   (theta = the device-side proposal)
//                   chain_step_kernel (segment sums -> L, prior, ratio, accept
//                                      test, state update, sample row, next
//                                      proposal)
//
// The step kernel repeats the host's arithmetic operation for operation (the
// slot sum in segment order, the prior as a sequential sum of log(pdf) terms,
// pposterior = prior + L, ratio = pposterior - cposterior, log(u) <= ratio);
// proposals and samples are bit-identical to the host loop's, the prior can
// differ from glibc's in the last bit of exp/log.
#pragma once

#include "mhlik_kernels.hpp"

namespace mhlik {

constexpr int kMaxSegmentsDev = 64; // == MHLIK_MAX_SEGMENTS
constexpr int kPriorChunk = 256;

struct ChainHdr {
  double cpost;       // posterior of the current state (a float value in the SP chain)
  long long accepted; // accepted proposals since the last begin / resume
  long long pad[2];
};

struct ChainDev {
  ChainHdr *hdr;
  void *cur, *prop;    // T[d]
  const void *inc;     // T[n][d]  proposal increments, drawn by the host
  const void *logu;    // T[n]     log of the uniforms
  void *samples;       // T[n][d]  state after each iteration
  double *lik, *ratio; // [n]
  unsigned char *acc;  // [n]
  int d, n;
  double pdf_norm;     // 1 / (sqrt(2 pi) |sigma|)
  double abs_sigma;
  // A batch travels as ONE upload image and ONE download image (mhlik_chain_abi.inc): the state a caller resumes
  // from rides at the head of the upload image and is taken over by the batch's first kernel (load_state != 0);
  // the batch's last step kernel leaves a copy of the state at the head of the download image.
  const ChainHdr *hdr_in;
  const void *cur_in; // T[d]
  ChainHdr *hdr_out;
  void *cur_out;      // T[d]
  int load_state;
};

// log(gsl_ran_gaussian_pdf(x, sigma)): u = x/|sigma|; p = norm * exp(-u*u/2)
// (mcmc_gpu.cu:648-658 calls it per component)
__device__ __forceinline__ double log_gauss_pdf(double x, double norm, double abs_sigma)
{
  const double u = x / abs_sigma;
  const double p = norm * exp(-u * u / 2.0);
  return log(p);
}

// prop = cur + inc[it]   (first proposal of a batch; later ones are made by
// the step kernel)
template <typename T> __global__ __launch_bounds__(kBlock) void chain_propose_kernel(ChainDev c, int it)
{
  const T *inc = (const T *)c.inc;
  T *cur = (T *)c.cur, *prop = (T *)c.prop;
  if (c.load_state) { // the state mhlik_chain_resume handed over, uploaded with this batch
    const T *cur_in = (const T *)c.cur_in;
    if (threadIdx.x == 0) *c.hdr = *c.hdr_in;
    for (int k = threadIdx.x; k < c.d; k += kBlock) {
      const T x = cur_in[k];
      cur[k] = x;
      prop[k] = x + inc[(int64_t)it * c.d + k];
    }
    return;
  }
  for (int k = threadIdx.x; k < c.d; k += kBlock) prop[k] = cur[k] + inc[(int64_t)it * c.d + k];
}

// mode 0: MH iteration `it` (0 <= it < c.n) of the batch
// mode 1: chain start -- the pass was over prop = theta0: cur = prop,
//         cposterior = prior + L; nothing is written to the batch outputs
//
// slots_in != nullptr: the per-segment sums are there already (after the
// segment kernel and, with several GPUs, the all-reduce).  Otherwise this
// kernel adds the tile partials itself, one wave per segment, with the same
// code and therefore the same bits as segment_sum_kernel (needs every segment
// local and nseg <= waves of the workgroup).
template <typename T>
__global__ __launch_bounds__(kSegBlock) void chain_step_kernel(ChainDev c, int it, int mode,
                                                               const double *__restrict__ slots_in, int nseg,
                                                               const double *__restrict__ partials,
                                                               const SegDesc *__restrict__ segs)
{
  __shared__ double slots[kMaxSegmentsDev];
  __shared__ double terms[kPriorChunk];
  __shared__ int s_accept;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (!slots_in) {
    if (wave < nseg) {
      const SegDesc sd = segs[wave];
      const double v = wave_sum_lane63(strided_sum<kWave>(partials + sd.tile_begin, lane, sd.tile_count));
      if (lane == kWave - 1) slots[sd.global_index] = v;
    }
  } else if ((int)threadIdx.x < nseg) {
    slots[threadIdx.x] = slots_in[threadIdx.x];
  }

  T *cur = (T *)c.cur, *prop = (T *)c.prop;
  const int d = c.d;
  // prior: the terms in parallel, then added in column order like the host
  // loop does (float chain: a float accumulator += a double term)
  T prior = (T)0;
  for (int k0 = 0; k0 < d; k0 += kPriorChunk) {
    const int m = d - k0 < kPriorChunk ? d - k0 : kPriorChunk;
    __syncthreads();
    for (int k = threadIdx.x; k < m; k += blockDim.x)
      terms[k] = log_gauss_pdf((double)prop[k0 + k], c.pdf_norm, c.abs_sigma);
    __syncthreads();
    // eight LDS reads in flight at a time; the additions keep the host loop's order
    int k = 0;
    for (; k + 8 <= m; k += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = terms[k + u];
#pragma unroll
      for (int u = 0; u < 8; u++) prior = (T)((double)prior + t[u]);
    }
    for (; k < m; k++) prior = (T)((double)prior + terms[k]);
  }

  if (threadIdx.x == 0) {
    double total = 0.0;
    for (int g = 0; g < nseg; g++) total += slots[g];
    const T L = (T)total;
    const T ppost = prior + L;
    ChainHdr *h = c.hdr;
    if (mode == 1) {
      h->cpost = (double)ppost;
      h->accepted = 0;
      s_accept = 1;
    } else {
      const T ratio = ppost - (T)h->cpost;
      const T logu = ((const T *)c.logu)[it];
      const int a = logu <= ratio ? 1 : 0;
      if (a) {
        h->cpost = (double)ppost;
        h->accepted += 1;
      }
      c.lik[it] = (double)L;
      c.ratio[it] = (double)ratio;
      c.acc[it] = (unsigned char)a;
      s_accept = a;
    }
  }
  __syncthreads();
  const bool accept = s_accept != 0;
  const T *inc = (const T *)c.inc;
  T *samples = (T *)c.samples;
  const bool last = mode == 0 && it + 1 == c.n && c.cur_out != nullptr;
  for (int k = threadIdx.x; k < d; k += blockDim.x) {
    const T x = accept ? prop[k] : cur[k];
    if (accept) cur[k] = x;
    if (mode == 0) {
      if (samples) samples[(int64_t)it * d + k] = x;
      if (it + 1 < c.n) prop[k] = x + inc[(int64_t)(it + 1) * d + k];
      if (last) ((T *)c.cur_out)[k] = x;
    }
  }
  if (last && threadIdx.x == 0) *c.hdr_out = *c.hdr; // thread 0 wrote the header above
}

} // namespace mhlik
