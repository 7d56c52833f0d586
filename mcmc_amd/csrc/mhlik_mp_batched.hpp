// mhlik_mp_batched.hpp -- mixed precision for K lock-step chains (BASELINE.json
// config 5: "mcmc_mp with 64 batched chains, X.Theta on MFMA").
//
// Per chain the definition is the one of mhlik_mp.hpp / oracle/mp_path.c
// (src/mcmc_mp/mcmc_mp.cu:28-200,475-552): every chain k has its own bright /
// dark indicators z[block][k], its own re-examination phase resamp_choose[k]
// and its own per-block uniforms (seed[k]).  The work is split by what it
// needs, not by data block:
//
//   mpb_dark_kernel  the float pass over ALL rows for ALL chains -- the
//                    reference's cublasSgemv over the whole matrix
//                    (mcmc_mp.cu:515) -- as X.Theta on v_mfma_f32_32x32x2_f32
//                    (the pass of mhlik_batched.hpp, bit-for-bit the k-ordered
//                    fmaf chain of the single-chain kernel).  Epilogue per
//                    (row, chain): B = sigma_S - eps, log B on the hardware
//                    exp2/rcp/log2 units, summed per chain over the rows whose
//                    data block is DARK for that chain.  MFMA-bound.
//   mpb_fix_kernel   the double-precision correction, only for the (block,
//                    chain) pairs that are bright or re-examined this call:
//                    a work item is one data block x up to 4 chains that need
//                    it, run by one wave, one row per lane: the block's double
//                    and float rows are dotted against those chains' theta
//                    (broadcast from LDS); the items of a block run side by
//                    side in one workgroup so its rows come from HBM once.  Row terms are exactly
//                    the single-chain kernel's (same madd / fmaf chains, the
//                    shared deterministic float sigmoid); z is resampled with
//                    P(bright) = 1 - exp(sum(log B - log L)).  Lane k of a wave
//                    keeps chain k's running bright sum and bright count over
//                    a fixed subset of the blocks, so there are no atomics and
//                    the result does not depend on scheduling.
//   mpb_final_kernel one workgroup per (chain, local segment) adds the segment's
//                    tile partials of the dark pass and its per-wave bright sums /
//                    counts; with several segments or GPUs mpb_combine_kernel adds
//                    the segments' values in order (after the in-stream all-reduce).
//
// Dark rows use fast float math (they are a float sum in the reference too);
// everything that decides z or enters log(L - B) is the exact per-row
// arithmetic of the single-chain path, so chain k reproduces mhlik_mp_loglik
// run with (theta_k, z_k, resamp_choose_k, seed_k) to float-sum tolerance with
// identical indicators (tests/test_gpu_mp_batched.py).
#pragma once

#include "mhlik_batched.hpp"
#include "mhlik_batched_x3.hpp"
#include "mhlik_mp.hpp"

namespace mhlik {

constexpr int kMpbFixGroups = 256;                 // workgroups of the fix kernel (fixed: it shapes the sums)
constexpr int kMpbFixBlock = 1024;                 // 16 waves share one LDS copy of Theta
constexpr int kMpbFixWaves = kMpbFixGroups * (kMpbFixBlock / kWave); // each wave visits a contiguous run of blocks
constexpr int kMpbGroup = 4;                       // chains dotted against one load of a block's rows
constexpr int kMpbU = 4;                           // columns per load batch of the fix kernel

struct MpbParams {
  int32_t block_rows;   // 64 .. 1024, power of two
  int32_t block_shift;  // log2(block_rows)
  int32_t blocks;
  int32_t resamp_factor;
  int32_t K;
  int32_t pad;
  uint64_t u_call;
  double eps;
  int64_t n_rows;
  int64_t row_base;     // global row of the first row this context holds (a multiple of 1024; 0 without shards)
  int32_t seg_groups;   // workgroups of the correction kernels per local segment: kMpbFixGroups / segments
  int32_t nseg;         // global segment count
};

// Data blocks of one local segment: [blk_lo, blk_hi) in global block numbers (blocks never straddle segments: a
// segment is a multiple of 1024 rows; blocks beyond the last row belong to the last segment), and the segment's
// global index.  One entry per local segment, in device memory.
struct MpbSeg {
  int32_t blk_lo, blk_hi, global, pad;
};

struct MpbChainArgs {
  int32_t choose[kBatchCols];
  uint64_t seed[kBatchCols];
};

// log(sigmoid - eps) with sigmoid = 1/(1+e^x) on v_exp_f32 / v_rcp_f32 / v_log_f32
__device__ __forceinline__ float mpb_dark_term(float x, float eps)
{
  const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
  const float sig = __builtin_amdgcn_rcpf(1.0f + e);
  return 0.69314718055994531f * __builtin_amdgcn_logf(sig - eps);
}

// grid = f32 tiles (1024 rows); z: [block][64 chains]; partials[chain*pstride + tile]
template <int NB, int KU>
__global__ __launch_bounds__(kBlock, 2) void mpb_dark_kernel(const float *__restrict__ X, TileGeom g,
                                                             const float *__restrict__ theta_dev, int d_pad,
                                                             MpbParams p, const int8_t *__restrict__ z,
                                                             double *__restrict__ partials, int64_t pstride)
{
  typedef Vec16<float>::type V4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *thetaL = reinterpret_cast<float *>(smem);
  double *red = reinterpret_cast<double *>(smem + (size_t)d_pad * kBatchCols * sizeof(float)); // [4 waves][NB][64 lanes]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile = blockIdx.x;
  const int d = g.d;
  for (int i = threadIdx.x * 4; i < d_pad * kBatchCols; i += kBlock * 4)
    *reinterpret_cast<V4 *>(thetaL + i) = *reinterpret_cast<const V4 *>(theta_dev + i);
  __syncthreads();

  const int arow = lane & 31, kk = lane >> 5;
  const float *tbase = X + tile * g.tile_stride;
  const int64_t all = (int64_t)p.blocks << p.block_shift;
  const int64_t covered = all < p.n_rows ? all : p.n_rows; // rows that belong to a considered block
  const float epsf = (float)p.eps;
  double sum[NB];
#pragma unroll
  for (int n = 0; n < NB; n++) sum[n] = 0.0;

#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
    const int row0 = wave * 256 + pass * 128;
    const int64_t g0 = p.row_base + tile * kMpTileRows + row0; // global row of the pass's first row
    if (g0 >= covered) break;                                  // wave-uniform
    f32x16 acc[4][NB];
    batched_f32_pass<NB, KU>(tbase + row0, g.col_stride, d, thetaL, arow, kk, acc);
    // indicators of the two 64-row groups of this pass (a group never straddles a data block)
    int8_t zc[2][NB];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int64_t gb = g0 + 64 * h;
      const int64_t blk = gb >> p.block_shift;
#pragma unroll
      for (int n = 0; n < NB; n++) zc[h][n] = gb < covered ? z[blk * kBatchCols + n * 32 + arow] : (int8_t)1;
    }

    // accumulator (m, n, r) of lane l: tile row row0 + 4*i + m, i = (r&3) + 8*(r>>2) + 4*(l>>5);
    // r < 8 is the first 64-row group of the pass, r >= 8 the second
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int64_t left = covered - (g0 + 64 * h); // rows of this group inside the considered blocks
      const int lim = left >= 64 ? 64 : (left > 0 ? (int)left : 0);
      const int lbase = 16 * kk;                     // group row of (rr = 0, m = 0) in this lane
#pragma unroll
      for (int n = 0; n < NB; n++) {
        float s = 0.f;
        if (lim == 64) { // wave-uniform
#pragma unroll
          for (int rr = 0; rr < 8; rr++)
#pragma unroll
            for (int m = 0; m < 4; m++) s += mpb_dark_term(acc[m][n][h * 8 + rr], epsf);
        } else {
#pragma unroll
          for (int rr = 0; rr < 8; rr++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
              const float t = mpb_dark_term(acc[m][n][h * 8 + rr], epsf);
              s += (lbase + 4 * ((rr & 3) + 8 * (rr >> 2)) + m < lim) ? t : 0.f;
            }
        }
        sum[n] += zc[h][n] == 0 ? (double)s : 0.0;
      }
    }
  }

#pragma unroll
  for (int n = 0; n < NB; n++) red[(wave * NB + n) * 64 + lane] = sum[n];
  __syncthreads();
  if ((int)threadIdx.x < NB * 32) {
    const int n = threadIdx.x >> 5, c = threadIdx.x & 31;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < 4; w++) s += red[(w * NB + n) * 64 + c] + red[(w * NB + n) * 64 + c + 32];
    const int chain = n * 32 + c;
    if (chain < p.K) partials[(int64_t)chain * pstride + tile] = s;
  }
}

// The same pass with the float dots formed from three bfloat16 pieces (mhlik_batched_x3.hpp; chosen by
// mhlik_set_batched_f32_mode, like the plain K-chain float pass).  Only DARK rows take their value from this pass,
// and a dark row's term is a float sum in the reference too; everything that decides an indicator or enters
// log(L - B) is recomputed by mpb_fix_kernel in the single-chain kernel's exact order either way, so the indicators
// do not depend on the mode.  Persistent workgroups (grid = what fits the chip), a wave's pass is 64 rows = one
// 64-row group of one data block.  Accumulator (m, n, r) of lane l: tile row row0 + 4*(4*(l>>4) + r) + m,
// chain 16n + (l&15).  z: [block][64 chains]; partials[chain*pstride + tile].
template <int NB>
__global__ __launch_bounds__(kBlock, 2) void mpb_dark_x3_kernel(const float *__restrict__ X, TileGeom g,
                                                                const u32x4 *__restrict__ theta_dev, int d_pad,
                                                                MpbParams p, const int8_t *__restrict__ z,
                                                                double *__restrict__ partials, int64_t pstride,
                                                                int64_t ntiles)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4 *thetaL = reinterpret_cast<u32x4 *>(smem);
  const int theta_vecs = d_pad / kX3KBlock * NB * 3 * 64;
  double *red = reinterpret_cast<double *>(smem + (size_t)theta_vecs * sizeof(u32x4)); // [4 waves][NB][64 lanes]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < theta_vecs; i += kBlock) thetaL[i] = theta_dev[i];
  __syncthreads();

  const int grp = lane >> 4, ccol = lane & 15;
  const int64_t all = (int64_t)p.blocks << p.block_shift;
  const int64_t covered = all < p.n_rows ? all : p.n_rows; // rows that belong to a considered block
  const float epsf = (float)p.eps;
  // product form of a lane's dark terms (see epi): c = eps / (1 - eps), 16 log(1 - eps), and the largest dot it takes:
  // c e^x <= 2e-4 (first-order log, error 2e-8) and x <= 7 (eight factors 1 + e^x stay far below the float range)
  const float cfac = (float)(p.eps / (1.0 - p.eps)), l1e16 = (float)(16.0 * log1p(-p.eps));
  const float xlim = fminf(7.0f, logf(2e-4f / cfac));
  double sum[NB];
#pragma unroll
  for (int n = 0; n < NB; n++) sum[n] = 0.0;

  auto epi = [&](int64_t tile, int row0, const f32x4 (&acc)[4][NB]) {
    const int64_t g0 = p.row_base + tile * kMpTileRows + row0; // global row of the pass's first row
    const int64_t left = covered - g0;
    if (left <= 0) return; // wave-uniform
    const int lim = left >= 64 ? 64 : (int)left;
    const int64_t blk = g0 >> p.block_shift;
#pragma unroll
    for (int n = 0; n < NB; n++) {
      const int8_t zc = z[blk * kBatchCols + n * 16 + ccol];
      float s = 0.f;
      if (lim == 64) { // wave-uniform
        // A lane's 16 terms in product form: log(sigma - eps) = log(1 - eps) + log(1 - c e^x) - log(1 + e^x) with
        // c = eps / (1 - eps); while c e^x <= 2e-4 its log is -c e^x to 2e-8 (below the float sum's own rounding),
        // and the 1 + e^x multiply up eight at a time (< 1.1e3^8 for x <= 7): one v_exp_f32 per element and two
        // v_log_f32 per 16 instead of exp + rcp + log each.  A group that holds an x > xlim (or +inf) anywhere in
        // the wave takes the per-element form; NaN travels through the sums.
        float sq0 = 0.f, sq1 = 0.f, pd0 = 1.f, pd1 = 1.f, xm = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int m = 0; m < 4; m++) {
            const float x = acc[m][n][r];
            const float q = __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
            __asm__("v_max_f32 %0, %1, %2" : "=v"(xm) : "v"(xm), "v"(x)); // no canonicalisation of x first
            if (r < 2) {
              sq0 += q;
              pd0 = fmaf(pd0, q, pd0);
            } else {
              sq1 += q;
              pd1 = fmaf(pd1, q, pd1);
            }
          }
        s = fmaf(-cfac, sq0 + sq1, l1e16) -
            0.69314718055994531f * (__builtin_amdgcn_logf(pd0) + __builtin_amdgcn_logf(pd1));
        if (__builtin_amdgcn_ballot_w64(!(xm <= xlim)) != 0) { // wave-uniform, rare
          s = 0.f;
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int m = 0; m < 4; m++) s += mpb_dark_term(acc[m][n][r], epsf);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int m = 0; m < 4; m++) {
            const float t = mpb_dark_term(acc[m][n][r], epsf);
            s += (4 * (4 * grp + r) + m < lim) ? t : 0.f;
          }
      }
      sum[n] += zc == 0 ? (double)s : 0.0;
    }
  };
  auto tile_done = [&](int64_t tile) {
#pragma unroll
    for (int n = 0; n < NB; n++) {
      red[(wave * NB + n) * 64 + lane] = sum[n];
      sum[n] = 0.0;
    }
    __syncthreads();
    if ((int)threadIdx.x < NB * 16) {
      const int n = threadIdx.x >> 4, c = threadIdx.x & 15;
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < 4; w++)
#pragma unroll
        for (int q = 0; q < 4; q++) s += red[(w * NB + n) * 64 + c + 16 * q];
      const int chain = n * 16 + c;
      if (chain < p.K) partials[(int64_t)chain * pstride + tile] = s;
    }
    __syncthreads();
  };
  batched_x3_stream<NB, 6>(X, g.tile_stride, (int64_t)blockIdx.x, (int64_t)gridDim.x, ntiles, wave, g.col_stride, g.d,
                              d_pad, thetaL, lane, epi, tile_done);
}

// Theta three ways, on the device, from the caller's row-major double Theta (raw[chain*d + j]):
//   pd / pf  [K][d_fix] doubles / floats, zero padded: the scalar loads of mpb_fix_kernel ((float)theta[j] as the
//            single-chain kernel rounds it);
//   image    the float Theta of the dark pass: the three-piece image of mhlik_batched_x3.hpp (split != 0), else the
//            B-operand order of the f32 MFMA pass, [column j][chain-in-block 32][block 2], d_pad columns.
__global__ __launch_bounds__(kBlock) void mpb_theta_prep_kernel(const double *__restrict__ raw, int d, int K, int d_fix,
                                                                double *__restrict__ pd, float *__restrict__ pf,
                                                                int split, int d_pad, int NB3, void *__restrict__ image)
{
  const int64_t first = (int64_t)blockIdx.x * kBlock + threadIdx.x, step = (int64_t)gridDim.x * kBlock;
  for (int64_t i = first; i < (int64_t)K * d_fix; i += step) {
    const int k = (int)(i / d_fix), j = (int)(i % d_fix);
    const double v = j < d ? raw[(int64_t)k * d + j] : 0.0;
    pd[i] = v;
    pf[i] = (float)v;
  }
  if (split) {
    x3_theta_image(raw, d, K, d_pad, NB3, reinterpret_cast<unsigned *>(image), first, step);
  } else {
    float *img = reinterpret_cast<float *>(image);
    for (int64_t o = first; o < (int64_t)d_pad * kBatchCols; o += step) {
      const int n = (int)(o & 1), cc = (int)((o >> 1) & 31), j = (int)(o >> 6);
      const int chain = n * 32 + cc;
      img[o] = (j < d && chain < K) ? (float)raw[(int64_t)chain * d + j] : 0.f;
    }
  }
}

__device__ __forceinline__ double wave_total(double v)
{
  v = wave_sum_lane63(v);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// One row of a (block, chain) pair that needs double precision: exactly the
// per-row arithmetic of mp_tile_kernel / oracle/mp_path.c.  Kept out of line:
// four inlined copies of exp + three logs cost more registers than the call.
struct MpbTerms {
  double rs, lb; // this row's log B - log L, and log(L - B) (0 unless bright)
};
__device__ __noinline__ MpbTerms mpb_pair_terms(float af, double ad, double eps, bool bright)
{
  const float sig_s = mhlik_sigmoidf_det(af);
  const double Bd = (double)sig_s - eps; // mcmc_mp.cu:63,146
  MpbTerms t;
  if (bright) {
    const double L = 1.0 / (1.0 + exp(ad));
    const float Bf = (float)Bd;
    t.rs = (double)logf(Bf) - log(L);
    t.lb = log(L - (double)Bf);
  } else {
    // A dark (re-examined) pair needs log B - log L = log1p(B/L - 1) only, and B/L - 1 = B(1 + e^x) - 1 is of
    // the order of the float sigmoid's error (~1e-7): one exp, one fma and a four-term series (next term
    // < 2e-16 relative) replace exp, a division and two logs; the absolute error (~1e-16) is that of the two
    // logs.  Anything else (B <= 0 at the far tails, overflow of e^x, NaN) takes the definition itself.
    const double u = fma(Bd, exp(ad), Bd - 1.0);
    if (fabs(u) < 0x1p-10)
      t.rs = u * (1.0 - u * (0.5 - u * (1.0 / 3.0 - u * 0.25)));
    else
      t.rs = log(Bd) - mhlik_neg_softplus(ad); // log(1/(1+e^x)) = -log(1+e^x), mhlik_softplus.h
    t.lb = 0.0;
  }
  return t;
}

// The common case inline: a dark pair under re-examination whose u = B(1 + e^x) - 1 is small (always, away from the
// far tails).  Same float sigmoid, same series as mpb_pair_terms; e^x from mhlik_exp_fast (5e-16 relative, i.e. at
// most 5e-16 absolute on u, which is of order 1e-7).  Returns false when the pair needs mpb_pair_terms instead.
__device__ __forceinline__ bool mpb_reexamined_term(float af, double ad, double eps, double &rs)
{
  const float sig_s = mhlik_sigmoidf_det(af);
  const double Bd = (double)sig_s - eps;
  const double u = fma(Bd, mhlik_exp_fast(ad), Bd - 1.0);
  rs = u * (1.0 - u * (0.5 - u * (1.0 / 3.0 - u * 0.25)));
  return fabs(u) < 0x1p-10; // false for NaN
}

// The rows of data block b against the theta of G chains c[0..G): per chain
// the wave totals  lb = sum log(L - B)  (bright at entry)  and
// rs = sum (log B - log L).  thd / thf: Theta as [chain][d_pad] doubles /
// floats, zero padded to a multiple of kMpbU columns (LDS, or global when it
// does not fit); every lane reads the same address (broadcast).  The loads of
// column chunk i+1 are in flight while chunk i is accumulated.
template <int G>
__device__ __forceinline__ void mpb_fix_group(const float *__restrict__ Xf, const TileGeom &gf,
                                              const double *__restrict__ Xd, const TileGeom &gd, const double *thd,
                                              const float *thf, int d_pad, const MpbParams &p, int64_t covered,
                                              int64_t b, const int (&c)[kMpbGroup], uint64_t bright_mask, int lane,
                                              double (&lb)[kMpbGroup], double (&rs)[kMpbGroup])
{
  constexpr int U = kMpbU;
  const int d = gf.d;
  double s_lb[G], s_rs[G];
#pragma unroll
  for (int q = 0; q < G; q++) s_lb[q] = s_rs[q] = 0.0;
  const double *td[G];
  const float *tf[G];
#pragma unroll
  for (int q = 0; q < G; q++) {
    td[q] = thd + c[q] * d_pad;
    tf[q] = thf + c[q] * d_pad;
  }

  const int segs = p.block_rows >> 6;
  for (int seg = 0; seg < segs; seg++) {
    const int64_t first = (b << p.block_shift) + (int64_t)seg * 64;
    if (first >= covered) break; // wave-uniform
    const int64_t row = first + lane;
    const int64_t lrow = row - p.row_base; // row of this context's store
    const float *pf = Xf + (lrow >> 10) * gf.tile_stride + (lrow & 1023);
    const double *pd = Xd + (lrow >> 9) * gd.tile_stride + (lrow & 511);
    float af[G];
    double ad[G];
#pragma unroll
    for (int q = 0; q < G; q++) {
      af[q] = 0.f;
      ad[q] = 0.0;
    }
    double xd[U], nd[U];
    float xf[U], nf[U];
    const int csd = (int)gd.col_stride, csf = (int)gf.col_stride; // offsets inside one tile fit 32 bits
    const int nfull = d / U;
    int j0 = 0;
    if (nfull > 0) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        nd[u] = pd[u * csd];
        nf[u] = pf[u * csf];
      }
    }
    for (int bt = 0; bt < nfull; bt++, j0 += U) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        xd[u] = nd[u];
        xf[u] = nf[u];
      }
      if (bt + 1 < nfull) { // wave-uniform: the whole next batch or nothing
#pragma unroll
        for (int u = 0; u < U; u++) {
          nd[u] = pd[(j0 + U + u) * csd];
          nf[u] = pf[(j0 + U + u) * csf];
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int q = 0; q < G; q++) {
          ad[q] = madd(ad[q], td[q][j0 + u], xd[u]);
          af[q] = fmaf(tf[q][j0 + u], xf[u], af[q]);
        }
    }
    for (; j0 < d; j0++) { // d % U leftover columns
      const double x1 = pd[j0 * csd];
      const float f1 = pf[j0 * csf];
#pragma unroll
      for (int q = 0; q < G; q++) {
        ad[q] = madd(ad[q], td[q][j0], x1);
        af[q] = fmaf(tf[q][j0], f1, af[q]);
      }
    }
    if (row < covered) {
#pragma unroll
      for (int q = 0; q < G; q++) {
        const bool br = (bright_mask >> c[q]) & 1ull;
        const MpbTerms t = mpb_pair_terms(af[q], ad[q], p.eps, br);
        s_rs[q] += t.rs;
        if (br) s_lb[q] += t.lb; // a dark pair's sum stays +0.0 exactly as before
      }
    }
  }
#pragma unroll
  for (int q = 0; q < G; q++) {
    lb[q] = wave_total(s_lb[q]);
    rs[q] = wave_total(s_rs[q]);
  }
}

// grid = kMpbFixGroups x kMpbFixBlock.  acc_out / cnt_out: [kMpbFixWaves][64].
// Theta: [K][d_pad] doubles then [K][d_pad] floats, zero padded; LDS_THETA: the
// workgroup stages both into LDS (dynamic, K*d_pad*12 bytes).
//
// Each workgroup owns a fixed contiguous run of data blocks and walks it 16
// blocks at a time (wave w <-> block b0 + w):
//   1. lane k of wave w reads z[b0+w][k] and votes: which chains need block
//      b0+w in double precision (bright, or dark and due for re-examination);
//   2. the (block, up-to-4-chains) work items of the 16 blocks are dealt to the
//      16 waves in order, so the items of one block run side by side and share
//      its rows in L1/L2 (a block all 64 chains need is 16 items = the whole
//      workgroup; 16 blocks one chain each need are one item per wave);
//   3. wave w, lane k resamples z[b0+w][k] from the item results and adds to
//      chain k's running bright sum / count.
// Every sum is taken in an order fixed by (blocks, block_rows) alone.
template <bool LDS_THETA>
__global__ __launch_bounds__(kMpbFixBlock) void mpb_fix_kernel(const float *__restrict__ Xf, TileGeom gf,
                                                               const double *__restrict__ Xd, TileGeom gd,
                                                               const double *__restrict__ Theta,
                                                               const float *__restrict__ Thetaf, int d_pad,
                                                               MpbParams p, const MpbChainArgs *__restrict__ ca,
                                                               const MpbSeg *__restrict__ segtab,
                                                               int8_t *__restrict__ z, double *__restrict__ acc_out,
                                                               int32_t *__restrict__ cnt_out)
{
  constexpr int NW = kMpbFixBlock / kWave; // 16 waves = 16 blocks per round
  __shared__ unsigned long long need_mask[NW], bright_mask_s[NW];
  __shared__ int item_begin[NW + 1];
  __shared__ double res_rs[NW][kBatchCols], res_lb[NW][kBatchCols];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const double *thd = Theta;
  const float *thf = Thetaf;
  if constexpr (LDS_THETA) {
    double *ld = reinterpret_cast<double *>(smem);
    float *lf = reinterpret_cast<float *>(smem + sizeof(double) * (size_t)p.K * d_pad);
    for (int i = threadIdx.x; i < p.K * d_pad; i += kMpbFixBlock) {
      ld[i] = Theta[i];
      lf[i] = Thetaf[i];
    }
    thd = ld;
    thf = lf;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = blockIdx.x * NW + wave;
  const int64_t all = (int64_t)p.blocks << p.block_shift;
  const int64_t covered = all < p.n_rows ? all : p.n_rows;
  const int my_choose = ca->choose[lane];
  const uint64_t my_seed = ca->seed[lane];
  double acc_b = 0.0;
  int32_t cnt = 0;

  // workgroup (local segment, group): the segment's blocks in seg_groups contiguous runs -- a shape that depends on
  // the segment alone, so the sums are the same however the segments are dealt to GPUs
  const MpbSeg sg = segtab[blockIdx.x / p.seg_groups];
  const int64_t chunk = ((int64_t)(sg.blk_hi - sg.blk_lo) + p.seg_groups - 1) / p.seg_groups;
  const int64_t wg_begin = sg.blk_lo + (int64_t)(blockIdx.x % p.seg_groups) * chunk;
  const int64_t wg_end = wg_begin + chunk < sg.blk_hi ? wg_begin + chunk : sg.blk_hi;
  for (int64_t b0 = wg_begin; b0 < wg_end; b0 += NW) { // workgroup-uniform trip count
    // ---- 1. votes -------------------------------------------------------------
    const int64_t b = b0 + wave;
    const bool live = b < wg_end;
    int8_t zc = (live && lane < p.K) ? z[b * kBatchCols + lane] : (int8_t)0;
    const bool has_rows = live && (b << p.block_shift) < p.n_rows; // a block without rows keeps its indicator
    const bool bright = zc != 0;
    const bool re = !bright && ((int)(b % p.resamp_factor) == my_choose);
    const bool need = lane < p.K && has_rows && (bright || re);
    const uint64_t mask = __ballot(need), bmask = __ballot(bright);
    if (lane == 0) {
      need_mask[wave] = mask;
      bright_mask_s[wave] = bmask;
    }
    __syncthreads(); // also orders the LDS copy of Theta before its first use
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < NW; w++) {
        item_begin[w] = t;
        t += (__builtin_popcountll(need_mask[w]) + kMpbGroup - 1) / kMpbGroup;
      }
      item_begin[NW] = t;
    }
    __syncthreads();
    // ---- 2. work items ----------------------------------------------------------
    const int items = item_begin[NW];
    for (int it = wave; it < items; it += NW) {
      int w = 0;
      while (item_begin[w + 1] <= it) w++; // wave-uniform
      w = __builtin_amdgcn_readfirstlane(w);
      int skip = (it - item_begin[w]) * kMpbGroup;
      uint64_t m = need_mask[w];
      while (skip-- > 0) m &= m - 1;
      int c[kMpbGroup];
      int ng = 0;
#pragma unroll
      for (int q = 0; q < kMpbGroup; q++) {
        if (m) {
          c[q] = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(m));
          m &= m - 1;
          ng++;
        } else {
          c[q] = c[0];
        }
      }
      const uint64_t bm = bright_mask_s[w];
      double lb[kMpbGroup], rs[kMpbGroup];
      if (ng <= 1)
        mpb_fix_group<1>(Xf, gf, Xd, gd, thd, thf, d_pad, p, covered, b0 + w, c, bm, lane, lb, rs);
      else if (ng <= 2)
        mpb_fix_group<2>(Xf, gf, Xd, gd, thd, thf, d_pad, p, covered, b0 + w, c, bm, lane, lb, rs);
      else
        mpb_fix_group<4>(Xf, gf, Xd, gd, thd, thf, d_pad, p, covered, b0 + w, c, bm, lane, lb, rs);
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < kMpbGroup; q++)
          if (q < ng) {
            res_rs[w][c[q]] = rs[q];
            res_lb[w][c[q]] = lb[q];
          }
      }
    }
    __syncthreads();
    // ---- 3. resample, accumulate -------------------------------------------------
    if (need) {
      const double u = mp_uniform(my_seed, p.u_call, (uint64_t)b);
      const double p_bright = 1.0 - exp(res_rs[wave][lane]);
      const int8_t zn = (p_bright <= u) ? 0 : 1; // mcmc_mp.cu:185-193
      if (bright) acc_b += res_lb[wave][lane];
      if (zn != zc) z[b * kBatchCols + lane] = zn;
      zc = zn;
    }
    cnt += zc != 0 ? 1 : 0;
    // the next round's writes to the LDS tables come after its first barrier,
    // which every wave reaches only after this point
  }
  acc_out[(int64_t)wave_global * kBatchCols + lane] = acc_b;
  cnt_out[(int64_t)wave_global * kBatchCols + lane] = cnt;
}

// ---------------------------------------------------------------------------------------------------------
// The same correction when (nearly) every chain needs the same blocks -- all chains re-examine one phase
// (mcmc_amd_mp_multichain_sampler's shared_phase), or everything is bright as in the first calls.  Then a block's
// rows against the 64 thetas is a dense contraction and runs on the matrix cores instead of one row per lane
// against four thetas broadcast from LDS (which is bound by the LDS return path: 12 bytes per lane per
// multiply-add pair):
//   float dots   v_mfma_f32_16x16x4_f32 -- bit-for-bit the column-ordered fmaf chain of the single-chain kernel
//                (tools/check_mfma_order.hip), which log(L - B) needs: it is the log of a one-float-ulp difference;
//   double dots  v_mfma_f64_16x16x4_f64 -- a fused chain, so within 1 ulp of the unfused mul+add chain of the
//                single-chain kernel, not the same bits; it enters through L = 1/(1+e^x) only.
// The two MFMAs lay their results out differently (float: A-row 4g+r, double: A-row g+4r in lane (c, g),
// register r); the double rows are loaded in the permutation that makes element r of both the same row.
// Per (row, chain) the same mpb_pair_terms as above.
// Rounds 1-2 gave a wave 64 rows x 16 chains (16-byte loads, four row sets): the four waves of a block then
// streamed the same rows (HBM traffic 1.34 x the algorithmic bytes) with two k-steps of twelve registers in flight.
__device__ __forceinline__ double lanes_xor_sum_16_32(double v) // lanes c, c+16, c+32, c+48 -> (g0+g1)+(g2+g3)
{
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}


// The dense correction with every row read ONCE: one wave dots 16 rows against all 64 chains.
// A lane (c = l&15, g = l>>4) loads one float and one double per k-step -- row c (float) / row 4(c&3) + (c>>2)
// (double, the permutation that lines the two MFMAs' results up) of its 16-row quarter of a 64-row
// segment, column 4k + g -- and runs four v_mfma_f32_16x16x4_f32 and four v_mfma_f64_16x16x4_f64 on it, one per
// 16-chain group n: accumulator element (n, r) of lane (c, g) is row 4g + r, chain 16n + c.  A k-step costs 3
// registers per lane, so D of them are in flight per wave where the 64-row x 16-chain shape had two of twelve
// registers each, and the four waves that used to stream the same rows now stream four different quarters:
// nothing is requested twice.  An address is a wave-uniform base (segment, k-step) plus one 32-bit lane offset.
// Theta sits in LDS as doubles only, [column][chain pair n>>1][c][n&1] (a lane's four B operands are two
// conflict-free 16-byte reads); the float operand is (float) of the double, which is how mpb_theta_prep_kernel
// makes the float Theta of the other kernels.
// A workgroup (NT threads, NT/256 "stripes" of four waves) owns the same run of blocks as before and compacts the
// blocks worth a visit the same way.  The visited blocks are taken round_blocks (<= 16) at a time; the round's
// 64-row segments, numbered block by block, are dealt to the stripes in turn, and wave (stripe q = w>>2,
// sub = w&3) streams rows 16 sub .. 16 sub + 15 of every segment of stripe q (one load pipeline across segments
// and blocks; a 1024-row block is four stripes' work, four 64-row blocks are one each), adds per chain the sums
// over those rows into res[block of round][stripe slot][sub][chain] (slot = segment mod the slots per block,
// min(segments per block, stripes): an entry has one writer), and after a barrier thread (block of round, chain)
// adds the block's entries in order, resamples the indicator and keeps the chain's running bright sum and count:
// the outputs have the layout mpb_fix_kernel writes ([16 rows per workgroup][64]).
// Order of every sum: a lane's four rows, the four lane groups (xor 16, xor 32), a slot's segments (seg, seg + 4,
// ...), the slots, the four quarters of each -- fixed by block_rows alone.
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int kMpbDenseRound = 16; // visited blocks per round at most (one row of the outputs each)
constexpr int kMpbDenseDepth = 4;  // k-steps in flight per wave (8 spill at 16 waves per workgroup: profiles/r03_fix_dense_lab.txt)

template <int NT, int D>
__global__ __launch_bounds__(NT) void mpb_fix_dense_kernel(const float *__restrict__ Xf, TileGeom gf,
                                                             const double *__restrict__ Xd, TileGeom gd,
                                                             const double *__restrict__ Theta, int theta_ld, int d_pad,
                                                             MpbParams p, const MpbChainArgs *__restrict__ ca,
                                                             const MpbSeg *__restrict__ segtab, int8_t *__restrict__ z,
                                                             double *__restrict__ acc_out, int32_t *__restrict__ cnt_out,
                                                             int round_blocks)
{
  constexpr int NW = NT / kWave, NQ = NW / 4;
  constexpr int kOutRows = kMpbFixBlock / kWave; // rows of acc_out / cnt_out per workgroup
  static_assert(NW % 4 == 0 && kMpbDenseRound <= kOutRows && kOutRows % NW == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *thd = reinterpret_cast<double *>(smem);                                               // [d_pad][2][16][2]
  double *res = reinterpret_cast<double *>(smem + sizeof(double) * (size_t)d_pad * kBatchCols); // [round][spb][4][64][2]
  // Theta: [K][theta_ld] in global memory (coalesced reads), zero beyond K chains / theta_ld columns
  for (int i = threadIdx.x; i < d_pad * kBatchCols; i += NT) {
    const int chain = i / d_pad, j = i - chain * d_pad;
    const int n = chain >> 4, cc = chain & 15;
    thd[((j * 2 + (n >> 1)) * 16 + cc) * 2 + (n & 1)] = (chain < p.K && j < theta_ld) ? Theta[(size_t)chain * theta_ld + j] : 0.0;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int sub = wave & 3, stripe = wave >> 2;
  const int d = gf.d;
  const int nk = d_pad >> 2; // a multiple of D; k-steps past the last column meet zeros in Theta
  const int row_f = 16 * sub + c, row_d = 16 * sub + 4 * (c & 3) + (c >> 2);
  const int csf = (int)gf.col_stride, csd = (int)gd.col_stride; // offsets inside one tile fit 32 bits
  const int64_t all = (int64_t)p.blocks << p.block_shift;
  const int64_t covered = all < p.n_rows ? all : p.n_rows;
  const int segs = p.block_rows >> 6, seg_shift = p.block_shift - 6; // a power of two
  const int spb = segs < NQ ? segs : NQ; // table slots per block: one per stripe that works on it
  const int my_choose = ca->choose[lane];
  const uint64_t my_seed = ca->seed[lane];
  double acc_b = 0.0;
  int32_t cnt = 0;

  const MpbSeg sg = segtab[blockIdx.x / p.seg_groups]; // as in mpb_fix_kernel
  const int64_t chunk = ((int64_t)(sg.blk_hi - sg.blk_lo) + p.seg_groups - 1) / p.seg_groups;
  const int64_t wg_begin = sg.blk_lo + (int64_t)(blockIdx.x % p.seg_groups) * chunk;
  const int64_t wg_end = wg_begin + chunk < sg.blk_hi ? wg_begin + chunk : sg.blk_hi;

  __shared__ int visit_list[NT];
  __shared__ int visit_count[NW];
  const int choose0 = ca->choose[0];
  for (int64_t base = wg_begin; base < wg_end; base += NT) { // workgroup-uniform trip count
    {
      const int64_t b = base + threadIdx.x;
      bool visit = false;
      if (b < wg_end) {
        const uint32_t *zw = reinterpret_cast<const uint32_t *>(z + b * kBatchCols);
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < kBatchCols / 4; w++) {
          const int left = p.K - 4 * w; // chains of this word that exist (entries past K are never cleared)
          const uint32_t mask = left >= 4 ? 0xffffffffu : (left > 0 ? (1u << (8 * left)) - 1u : 0u);
          any |= zw[w] & mask;
        }
        visit = any != 0 || ((int)(b % p.resamp_factor) == choose0 && (b << p.block_shift) < p.n_rows);
      }
      const uint64_t vm = __ballot(visit);
      if (lane == 0) visit_count[wave] = __builtin_popcountll(vm);
      __syncthreads(); // also orders the LDS copy of Theta before its first use
      int before = 0;
      for (int w = 0; w < wave; w++) before += visit_count[w];
      if (visit) visit_list[before + __builtin_popcountll(vm & ((1ull << lane) - 1ull))] = (int)threadIdx.x;
    }
    int n_visit = 0;
    for (int w = 0; w < NW; w++) n_visit += visit_count[w];
    __syncthreads();

    for (int r0 = 0; r0 < n_visit; r0 += round_blocks) { // workgroup-uniform trip count
      const int rc = n_visit - r0 < round_blocks ? n_visit - r0 : round_blocks;
      // ---- stream: the round's 64-row segments are dealt to the stripes in order; this wave's quarter of its
      // stripe's segments ----------------------------------------------------------------------------------
      const int units = rc << seg_shift;
      auto slot_segs = [&](int64_t b) -> int { // 64-row segments of block b that hold rows
        const int64_t left = covered - (b << p.block_shift);
        if (left <= 0) return 0;
        const int64_t s = (left + 63) >> 6;
        return s < segs ? (int)s : segs;
      };
      struct Pos { // a quarter-segment ("unit") of this wave's stream and a k-step in it; all wave-uniform
        int u, k, r, seg; // unit = (block of the round r, segment seg), u = r * segs + seg
        int64_t b;
        const float *pf;  // the segment's row 0, column 0
        const double *pd;
      };
      auto seek = [&](Pos &q) -> bool { // first unit of this stripe at or after q.u whose segment holds rows
        for (; q.u < units; q.u += NQ) {
          q.r = q.u >> seg_shift;
          q.seg = q.u & (segs - 1);
          q.b = base + visit_list[r0 + q.r];
          if (q.seg < slot_segs(q.b)) {
            const int64_t lfirst = (q.b << p.block_shift) + (int64_t)q.seg * 64 - p.row_base;
            q.pf = Xf + (lfirst >> 10) * gf.tile_stride + (lfirst & 1023);
            q.pd = Xd + (lfirst >> 9) * gd.tile_stride + (lfirst & 511);
            return true;
          }
        }
        return false;
      };
      auto next_unit = [&](Pos &q) -> bool { // false (and q unchanged) after the wave's last unit
        Pos n = q;
        n.u += NQ;
        if (!seek(n)) return false;
        q = n;
        return true;
      };
      struct Buf {
        float f;
        double x;
      };
      Buf ring[D];
      auto load = [&](Buf &bf, const Pos &q) { // column 4k + g, or the last real column (Theta is zero there)
        int j = 4 * q.k + g;
        j = j < d ? j : d - 1;
        bf.f = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(q.pf) +
                                                (unsigned)((j * csf + row_f) * (int)sizeof(float)));
        bf.x = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(q.pd) +
                                                 (unsigned)((j * csd + row_d) * (int)sizeof(double)));
      };
      auto load_next = [&](Buf &bf, Pos &q) { // request the load position's k-step, then step it; past the end of
        load(bf, q);                          // the stream it wraps inside the last unit (requests nobody reads)
        if (++q.k == nk) {
          q.k = 0;
          (void)next_unit(q);
        }
      };
      f32x4 af[4];
      f64x4 ad[4];
      Pos w{stripe, 0, 0, 0, 0, nullptr, nullptr};
      if (seek(w)) {
        Pos l = w;
#pragma unroll
        for (int b = 0; b < D; b++) load_next(ring[b], l);
        bool more = true;
        while (more) {
          // which of this lane's four chains need block w.b, which are bright (read again per segment: cheap)
          unsigned needm = 0, brightm = 0;
#pragma unroll
          for (int n = 0; n < 4; n++) {
            const int chain = 16 * n + c;
            const bool live = chain < p.K;
            const int8_t zc = live ? z[w.b * kBatchCols + chain] : (int8_t)0;
            const bool bright = zc != 0;
            const bool re = !bright && live && ((int)(w.b % p.resamp_factor) == ca->choose[chain]);
            if (live && (bright || re)) needm |= 1u << n;
            if (bright) brightm |= 1u << n;
            af[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            ad[n] = f64x4{0.0, 0.0, 0.0, 0.0};
          }
          // chain groups none of whose 16 chains needs the block (a block visited for one bright chain) cost no MFMAs
          unsigned groups = 0;
#pragma unroll
          for (int n = 0; n < 4; n++) groups |= __ballot((needm >> n) & 1u) != 0 ? 1u << n : 0u;
          auto ksteps = [&](auto all_tag) {
            constexpr bool kAll = decltype(all_tag)::value;
            for (int kk = 0; kk < nk; kk += D) {
#pragma unroll
              for (int b = 0; b < D; b++) {
                const double *tb = thd + (size_t)(4 * (kk + b) + g) * kBatchCols + c * 2;
                const f64x2 t01 = *reinterpret_cast<const f64x2 *>(tb);
                const f64x2 t23 = *reinterpret_cast<const f64x2 *>(tb + 32);
                const double tn[4] = {t01[0], t01[1], t23[0], t23[1]};
                const float xf = ring[b].f;
                const double xd = ring[b].x;
#pragma unroll
                for (int n = 0; n < 4; n++)
                  if (kAll || ((groups >> n) & 1u)) af[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, (float)tn[n], af[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < 4; n++)
                  if (kAll || ((groups >> n) & 1u)) ad[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd, tn[n], ad[n], 0, 0, 0);
                load_next(ring[b], l);
              }
            }
          };
          if (groups == 0xfu)
            ksteps(std::true_type{});
          else
            ksteps(std::false_type{});
          // the quarter-segment is complete: its terms, per chain
          const int64_t first = (w.b << p.block_shift) + (int64_t)w.seg * 64 + 16 * sub + 4 * g;
          // the table entry of (block, stripe slot, sub): the block's segments seg, seg + NQ, ... add up in it
          double *out = res + ((size_t)((w.r * spb + (w.seg & (spb - 1))) * 4 + sub) * kBatchCols) * 2;
#pragma unroll
          for (int n = 0; n < 4; n++) {
            double s_rs = 0.0, s_lb = 0.0;
            if ((needm >> n) & 1u) {
              const bool bright = (brightm >> n) & 1u;
#pragma unroll
              for (int r = 0; r < 4; r++) {
                if (first + r < covered) {
                  double rs_fast;
                  if (!bright && mpb_reexamined_term(af[n][r], ad[n][r], p.eps, rs_fast)) {
                    s_rs += rs_fast;
                  } else {
                    const MpbTerms t = mpb_pair_terms(af[n][r], ad[n][r], p.eps, bright);
                    s_rs += t.rs;
                    if (bright) s_lb += t.lb;
                  }
                }
              }
            }
            s_rs = lanes_xor_sum_16_32(s_rs);
            s_lb = lanes_xor_sum_16_32(s_lb);
            if (g == 0) { // the block's segments add up in res, in order (this wave alone touches the entry)
              double *o = out + (16 * n + c) * 2;
              o[0] = w.seg < spb ? s_rs : o[0] + s_rs;
              o[1] = w.seg < spb ? s_lb : o[1] + s_lb;
            }
          }
          more = next_unit(w);
        }
      }
      __syncthreads();
      // ---- resample: thread (block of the round, chain) ----------------------------------------------------
      for (int slot = wave; slot < rc; slot += NW) {
        const int64_t b = base + visit_list[r0 + slot];
        int8_t zc = lane < p.K ? z[b * kBatchCols + lane] : (int8_t)0;
        const bool has_rows = (b << p.block_shift) < p.n_rows; // a block without rows keeps its indicator
        const bool bright = zc != 0;
        const bool re = !bright && ((int)(b % p.resamp_factor) == my_choose);
        const bool need = lane < p.K && has_rows && (bright || re);
        if (need) { // the block's stripe slots that hold rows, four quarters each, in order
          const int nsg = slot_segs(b);
          const int parts = 4 * (nsg < spb ? nsg : spb);
          const double *in = res + ((size_t)(slot * spb * 4) * kBatchCols + lane) * 2;
          double rs_tot = in[0], lb_tot = in[1];
          for (int s = 1; s < parts; s++) {
            rs_tot += in[(size_t)s * kBatchCols * 2 + 0];
            lb_tot += in[(size_t)s * kBatchCols * 2 + 1];
          }
          const double u = mp_uniform(my_seed, p.u_call, (uint64_t)b);
          const double p_bright = 1.0 - exp(rs_tot);
          const int8_t zn = (p_bright <= u) ? 0 : 1; // mcmc_mp.cu:185-193
          if (bright) acc_b += lb_tot;
          if (zn != zc) z[b * kBatchCols + lane] = zn;
          zc = zn;
        }
        cnt += zc != 0 ? 1 : 0;
      }
      __syncthreads(); // the next round rewrites res (and the next window the list)
    }
  }
  // rows wave, wave + NW, ... of this workgroup's 16: a wave's running sums in the first, zeros in the others
  for (int row = wave; row < kOutRows; row += NW) {
    const int64_t o = ((int64_t)blockIdx.x * kOutRows + row) * kBatchCols + lane;
    acc_out[o] = row == wave ? acc_b : 0.0;
    cnt_out[o] = row == wave ? cnt : 0;
  }
}

// grid = (K, local segments).  Per chain k and local segment: the dark pass's tile partials of the segment's tiles and
// the correction kernels' per-wave sums / counts of the segment's seg_groups * 16 waves, each in a fixed order:
//   v0 = sum over blocks bright at entry (double), v1 = sum over rows of blocks dark at entry, rounded to float,
//   v2 = bright blocks after resampling.
// One segment, one GPU (slots == nullptr): res[k*4 + 0..3] = v0, v1, v2, blocks - v2 -- the whole result.
// Otherwise slots[(k*4 + i)*nseg + global segment] = v_i, combined over segments (and GPUs) by mpb_combine_kernel.
__global__ __launch_bounds__(kBlock) void mpb_final_kernel(const double *__restrict__ partials, int64_t pstride,
                                                           const SegDesc *__restrict__ segs,
                                                           const MpbSeg *__restrict__ segtab, int seg_groups, int nseg,
                                                           const double *__restrict__ acc,
                                                           const int32_t *__restrict__ cnt, int32_t blocks,
                                                           double *__restrict__ res, double *__restrict__ slots)
{
  __shared__ double lds[kBlock / kWave];
  const int k = blockIdx.x, ls = blockIdx.y;
  const SegDesc sd = segs[ls];
  double dark = strided_sum<kBlock>(partials + (int64_t)k * pstride + sd.tile_begin, threadIdx.x, sd.tile_count);
  double br = 0.0, nb = 0.0;
  const int w0 = ls * seg_groups * (kMpbFixBlock / kWave), w1 = w0 + seg_groups * (kMpbFixBlock / kWave);
  for (int w = w0 + threadIdx.x; w < w1; w += kBlock) {
    br += acc[(int64_t)w * kBatchCols + k];
    nb += (double)cnt[(int64_t)w * kBatchCols + k];
  }
  dark = block_sum(dark, lds);
  __syncthreads();
  br = block_sum(br, lds);
  __syncthreads();
  nb = block_sum(nb, lds);
  if (threadIdx.x == 0) {
    if (slots == nullptr) {
      res[k * 4 + 0] = br;
      res[k * 4 + 1] = (double)(float)dark;
      res[k * 4 + 2] = nb;
      res[k * 4 + 3] = (double)blocks - nb;
    } else {
      const int g = segtab[ls].global;
      slots[(int64_t)(k * 4 + 0) * nseg + g] = br;
      slots[(int64_t)(k * 4 + 1) * nseg + g] = (double)(float)dark;
      slots[(int64_t)(k * 4 + 2) * nseg + g] = nb;
      slots[(int64_t)(k * 4 + 3) * nseg + g] = 0.0;
    }
  }
}

// thread k: the segments' values of chain k added in segment order; the float sums are added in double and rounded to
// float once (the single-chain path does the same on the host, mhlik_mp_loglik).
__global__ __launch_bounds__(kWave) void mpb_combine_kernel(const double *__restrict__ slots, int nseg, int K,
                                                            int32_t blocks, double *__restrict__ res)
{
  const int k = threadIdx.x;
  if (k >= K) return;
  double br = 0.0, dk = 0.0, nb = 0.0;
  for (int g = 0; g < nseg; g++) {
    br += slots[(int64_t)(k * 4 + 0) * nseg + g];
    dk += slots[(int64_t)(k * 4 + 1) * nseg + g];
    nb += slots[(int64_t)(k * 4 + 2) * nseg + g];
  }
  res[k * 4 + 0] = br;
  res[k * 4 + 1] = (double)(float)dk;
  res[k * 4 + 2] = nb;
  res[k * 4 + 3] = (double)blocks - nb;
}

} // namespace mhlik
