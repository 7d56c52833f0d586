// mhlik_kchain.hpp -- K lock-step MH chains kept on the device between iterations.
//
// The K-chain twin of mhlik_chain.hpp.  The reference runs one chain and goes back to the host after every
// likelihood (metropolis_gpu, src/mcmc_gpu/mcmc_gpu.cu:101-128); the K-chain host loop of samplers/multichain.c did the
// same for K proposals.  Which random numbers an iteration consumes does not depend on any chain's state (per chain:
// d Gaussians, then one uniform; mixed precision: d Gaussians, the re-examination phase, one uniform), so the host
// draws them for a whole batch of iterations, each chain from its own generator in the reference's order, hands them
// over once, and the device runs the batch back to back:
//
//   per iteration:  Theta image kernel   (the K device-side proposals in the batched pass's operand order)
//                   batched pass          (X.Theta on the matrix cores, mhlik_batched*.hpp / mhlik_mp_batched.hpp)
//                   segment sums          (+ the in-stream RCCL all-reduce with several GPUs)
//                   kchain_step_kernel    (per chain: L, prior, ratio, accept test, state update, sample row, next
//                                          proposal)
//
// The step kernel repeats multichain.c's host arithmetic operation for operation: the slot sum in segment order, the
// prior as a sequential sum of log(pdf) terms in column order, pposterior = prior + L, ratio = pposterior - cposterior,
// log(u) <= ratio (mixed precision: u <= exp(ratio), mcmc_mp.cu:402,:635).
#pragma once

#include "mhlik_batched_x3.hpp"
#include "mhlik_chain.hpp"

namespace mhlik {

struct KChainDev {
  double *cpost;       // [K] posterior of each chain's current state
  double *cur, *prop;  // [K][d]
  const double *inc;   // [n][K][d] proposal increments, drawn by the host
  const double *logu;  // [n][K]    log of the uniforms (exp rule: the uniforms themselves)
  double *samples;     // [n][K][d] state after each iteration
  double *lik, *ratio; // [n][K]
  unsigned char *acc;  // [n][K]
  int d, K, n;
  int exp_rule;        // 0: accept iff log u <= ratio (mcmc_gpu.cu:112-114); 1: iff u <= exp(ratio) (mcmc_mp.cu:402)
  double pdf_norm;     // 1 / (sqrt(2 pi) |sigma|)
  double abs_sigma;
};

// The proposals in the B-operand order of lik_batched_f32_kernel / lik_batched_f64_kernel (what batched_impl stages on
// the host): image[(j*NBLK + c)*per + n] = Theta[chain n*NBLK + c][column j], per = 64/NBLK, zero outside K chains and
// d columns.  The float image is the rounded double, like the host loop's (float) cast.
template <typename T>
__global__ __launch_bounds__(kBlock) void kchain_image_kernel(const double *__restrict__ prop, int d, int K, int d_pad,
                                                              int nblk, T *__restrict__ image)
{
  const int per = 64 / nblk;
  const int total = d_pad * 64;
  for (int o = blockIdx.x * kBlock + threadIdx.x; o < total; o += gridDim.x * kBlock) {
    const int j = o / 64, rem = o % 64;
    const int c = rem / per, n = rem % per;
    const int chain = n * nblk + c;
    image[o] = (j < d && chain < K) ? (T)prop[(int64_t)chain * d + j] : (T)0;
  }
}

// grid = K workgroups of one wave: chain k = blockIdx.x.
// mode 0: MH iteration `it` of the batch; mode 1: chain start (the pass was over prop = theta0).
// L of chain k: mp_res == nullptr: sum over g < nseg of slots[k*nseg + g], in that order (combine_wait's host loop);
//               else mp_res[k*4 + 0] + (float)mp_res[k*4 + 1]  (mhlik_mp_loglik_batched_end: red_d + red_f).
__global__ __launch_bounds__(kWave) void kchain_step_kernel(KChainDev c, int it, int mode,
                                                            const double *__restrict__ slots, int nseg,
                                                            const double *__restrict__ mp_res)
{
  __shared__ double terms[kPriorChunk];
  const int k = blockIdx.x, lane = threadIdx.x;
  const int d = c.d;
  double *cur = c.cur + (int64_t)k * d, *prop = c.prop + (int64_t)k * d;

  // prior: the terms in parallel, then every lane adds them in column order (the host loop's order)
  double prior = 0.0;
  for (int k0 = 0; k0 < d; k0 += kPriorChunk) {
    const int m = d - k0 < kPriorChunk ? d - k0 : kPriorChunk;
    __syncthreads();
    for (int j = lane; j < m; j += kWave) terms[j] = log_gauss_pdf(prop[k0 + j], c.pdf_norm, c.abs_sigma);
    __syncthreads();
    for (int j = 0; j < m; j++) prior += terms[j];
  }
  double L;
  if (mp_res) {
    const float red_f = (float)mp_res[k * 4 + 1];
    L = mp_res[k * 4 + 0] + red_f;
  } else {
    L = 0.0;
    for (int g = 0; g < nseg; g++) L += slots[(int64_t)k * nseg + g];
  }
  const double ppost = prior + L;
  bool accept;
  if (mode == 1) {
    accept = true;
    if (lane == 0) c.cpost[k] = ppost;
  } else {
    const double cpost = c.cpost[k];
    const double u = c.logu[(int64_t)it * c.K + k];
    const double ratio = c.exp_rule ? exp(ppost - cpost) : ppost - cpost;
    accept = u <= ratio;
    if (lane == 0) {
      if (accept) c.cpost[k] = ppost;
      c.lik[(int64_t)it * c.K + k] = L;
      c.ratio[(int64_t)it * c.K + k] = ratio;
      c.acc[(int64_t)it * c.K + k] = accept ? 1 : 0;
    }
  }
  const double *inc = c.inc + ((int64_t)(it + 1) * c.K + k) * d;
  double *srow = c.samples + ((int64_t)it * c.K + k) * d;
  for (int j = lane; j < d; j += kWave) {
    const double x = accept ? prop[j] : cur[j];
    if (accept) cur[j] = x;
    if (mode == 0) {
      srow[j] = x;
      if (it + 1 < c.n) prop[j] = x + inc[j];
    }
  }
}

// Everything between two batched passes in ONE launch (single GPU, <= 8 local segments): grid = K workgroups of 8
// waves, chain k = blockIdx.x.  Wave w adds the tile partials of segment w for this chain -- the code, and therefore
// the bits, of segment_sum_kernel -- then the step of kchain_step_kernel, then this chain's entries of the NEXT
// proposal's Theta image (every entry of an image depends on one chain only; entries of chains >= K and columns >= d
// were zeroed when the chain was set up).  IMG: 0 = double B-operand order, 1 = float B-operand order, 2 = the
// three-piece bfloat16 image of the split pass (x3_theta_image's layout).
constexpr int kPostBlock = 512;
template <int IMG>
__global__ __launch_bounds__(kPostBlock) void kchain_post_kernel(KChainDev c, int it, int mode,
                                                                 const double *__restrict__ partials, int64_t pstride,
                                                                 const SegDesc *__restrict__ segs, int nseg,
                                                                 void *__restrict__ image, int d_pad, int NB)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char post_smem[];
  double *newp = reinterpret_cast<double *>(post_smem); // [d_pad + 1] the next proposal of this chain, zero padded
  __shared__ double slots[kMaxSegmentsDev];
  __shared__ double terms[kPriorChunk];
  __shared__ int s_accept;
  const int k = blockIdx.x, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int d = c.d;
  double *cur = c.cur + (int64_t)k * d, *prop = c.prop + (int64_t)k * d;

  if (wave < nseg) {
    const SegDesc sd = segs[wave];
    const double v = wave_sum_lane63(strided_sum<kWave>(partials + (int64_t)k * pstride + sd.tile_begin, lane, sd.tile_count));
    if (lane == kWave - 1) slots[sd.global_index] = v;
  }
  double prior = 0.0;
  for (int k0 = 0; k0 < d; k0 += kPriorChunk) {
    const int m = d - k0 < kPriorChunk ? d - k0 : kPriorChunk;
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += kPostBlock) terms[j] = log_gauss_pdf(prop[k0 + j], c.pdf_norm, c.abs_sigma);
    __syncthreads();
    if (threadIdx.x == 0)
      for (int j = 0; j < m; j++) prior += terms[j];
  }
  if (threadIdx.x == 0) {
    double L = 0.0;
    for (int g = 0; g < nseg; g++) L += slots[g];
    const double ppost = prior + L;
    if (mode == 1) {
      c.cpost[k] = ppost;
      s_accept = 1;
    } else {
      const double ratio = ppost - c.cpost[k];
      const int a = c.logu[(int64_t)it * c.K + k] <= ratio ? 1 : 0;
      if (a) c.cpost[k] = ppost;
      c.lik[(int64_t)it * c.K + k] = L;
      c.ratio[(int64_t)it * c.K + k] = ratio;
      c.acc[(int64_t)it * c.K + k] = (unsigned char)a;
      s_accept = a;
    }
  }
  __syncthreads();
  const bool accept = s_accept != 0;
  const bool next = mode == 0 && it + 1 < c.n;
  const double *inc = c.inc + ((int64_t)(it + 1) * c.K + k) * d;
  double *srow = c.samples + ((int64_t)it * c.K + k) * d;
  for (int j = threadIdx.x; j <= d_pad; j += kPostBlock) {
    double np_ = 0.0;
    if (j < d) {
      const double x = accept ? prop[j] : cur[j];
      if (accept) cur[j] = x;
      if (mode == 0) srow[j] = x;
      if (next) {
        np_ = x + inc[j];
        prop[j] = np_;
      }
    }
    newp[j] = np_;
  }
  if (!next) return;
  __syncthreads();
  if constexpr (IMG == 0) {
    double *img = reinterpret_cast<double *>(image);
    const int n = k / 16, cc = k % 16;
    for (int j = threadIdx.x; j < d_pad; j += kPostBlock) img[(int64_t)j * 64 + cc * 4 + n] = newp[j];
  } else if constexpr (IMG == 1) {
    float *img = reinterpret_cast<float *>(image);
    const int n = k / 32, cc = k % 32;
    for (int j = threadIdx.x; j < d_pad; j += kPostBlock) img[(int64_t)j * 64 + cc * 2 + n] = (float)newp[j];
  } else {
    unsigned *img = reinterpret_cast<unsigned *>(image);
    const int n = k / 16, cc = k % 16;
    const int words = d_pad / kX3KBlock * 3 * 16; // (kb, t, g, i)
    for (int w = threadIdx.x; w < words; w += kPostBlock) {
      const int i = w & 3, g = (w >> 2) & 3;
      const int t = (w >> 4) % 3, kb = (w >> 4) / 3;
      const int j0 = kb * kX3KBlock + 8 * g + 2 * i;
      const unsigned lo = x3_piece((float)newp[j0], t), hi = x3_piece((float)newp[j0 + 1], t);
      img[((((int64_t)kb * NB + n) * 3 + t) * 64 + (g * 16 + cc)) * 4 + i] = lo | (hi << 16);
    }
  }
}

// prop = cur + inc[0]: the first proposal of a batch (later ones are made by the step kernel)
__global__ __launch_bounds__(kBlock) void kchain_propose_kernel(KChainDev c)
{
  const int64_t total = (int64_t)c.K * c.d;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock)
    c.prop[i] = c.cur[i] + c.inc[i];
}

} // namespace mhlik
