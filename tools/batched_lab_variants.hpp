// batched_lab_variants.hpp -- LAB copies of the f32-MFMA and f64-MFMA K-chain kernels of mcmc_amd/csrc/mhlik_batched.hpp
// with the experiment switches of rounds 1-2 (EXP bits: 1 plain sum instead of the epilogue, 2 no global loads inside the
// k-loop, 4 ocml log1p/exp per element (f64), 8 product-form epilogue (f64)).  The product header carries none of
// them; nothing under mcmc_amd/ includes this file.  tools/batched_lab.hip and tools/x3_lab.hip time these.
#pragma once

#include "mhlik_batched.hpp"

namespace mhlik {

// One 128-row pass of a wave: acc[m][n] = rows {row0 + 4i + m} x chains of
// N-block n, accumulated over all d columns.  ubase = tile + row0 (wave-uniform): a load's address is a
// scalar base (column of the k-step) plus one 32-bit per-lane byte offset, not a 64-bit VGPR pair per load.
template <int NB, int KU, int EXP>
__device__ __forceinline__ void lab_batched_f32_pass(const float *__restrict__ ubase, int64_t col_stride, int d,
                                                 const float *thetaL, int arow, int kk, f32x16 (&acc)[4][NB])
{
  typedef Vec16<float>::type V4;
  const unsigned loff = (unsigned)((4 * arow + kk * col_stride) * sizeof(float)); // lane's rows, lane's column of the k-step
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int n = 0; n < NB; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

  // Software pipeline over batches of KU k-steps with two register buffers and
  // no copies between them: the loads of batch b+1 are issued, then the MFMAs
  // of batch b run, and only then is batch b+1 waited for -- a full batch of
  // matrix work (KU*4*NB MFMAs of 64 cycles) covers the HBM latency.  (With a
  // single "next" buffer copied into the "current" one the compiler hoists the
  // copies, and their waits, up between the MFMAs of the current batch.)
  int j0 = 0;
  const int nfull = d / (2 * KU);
  V4 bufA[KU], bufB[KU];
  auto load_batch = [&](V4 (&buf)[KU], int jb) {
#pragma unroll
    for (int u = 0; u < KU; u++)
      buf[u] = load16<float, true>(reinterpret_cast<const float *>(
          reinterpret_cast<const char *>(ubase + (int64_t)(jb + 2 * u) * col_stride) + loff));
  };
  auto mfma_batch = [&](const V4 (&buf)[KU], int jb) {
#pragma unroll
    for (int u = 0; u < KU; u++) {
      const f32x2 b = *reinterpret_cast<const f32x2 *>(thetaL + ((jb + 2 * u + kk) * 32 + arow) * 2);
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < NB; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[u][m], b[n], acc[m][n], 0, 0, 0);
    }
  };
  if (nfull > 0) load_batch(bufA, 0);
  int bt = 0;
  for (; bt + 1 < nfull; bt += 2, j0 += 4 * KU) {
    if (!(EXP & 2)) load_batch(bufB, j0 + 2 * KU);
    __builtin_amdgcn_sched_barrier(0); // keep the loads ahead of the MFMAs: the scheduler sinks them otherwise
    mfma_batch(bufA, j0);
    __builtin_amdgcn_sched_barrier(0);
    if (!(EXP & 2) && bt + 2 < nfull) load_batch(bufA, j0 + 4 * KU);
    __builtin_amdgcn_sched_barrier(0);
    mfma_batch(bufB, j0 + 2 * KU);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (bt < nfull) { // odd number of batches: the last one sits in bufA
    mfma_batch(bufA, j0);
    j0 += 2 * KU;
  }
  for (; j0 < d; j0 += 2) { // leftover k-steps; a column >= d contributes zeros
    const int col = j0 + kk;
    V4 a1 = {0.f, 0.f, 0.f, 0.f};
    if (col < d)
      a1 = load16<float, true>(reinterpret_cast<const float *>(
          reinterpret_cast<const char *>(ubase + (int64_t)j0 * col_stride) + loff));
    const f32x2 b = *reinterpret_cast<const f32x2 *>(thetaL + (col * 32 + arow) * 2);
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int n = 0; n < NB; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b[n], acc[m][n], 0, 0, 0);
  }
}

// ------------------------------------------------------------------- f32
// Workgroup = 256 threads = 4 waves; f32 tile = 1024 rows; wave w owns rows
// [256w, 256w+256) in two passes of 128 rows (32 A-rows x 4 row sets).
// NB = number of 32-chain blocks (1: K <= 32, 2: K <= 64).  KU = k-steps
// (pairs of columns) whose loads are in flight together.  EXP != 0 builds the
// ablations tools/batched_lab.hip times (bit 0: plain sum instead of the
// transcendental epilogue, bit 1: no global loads inside the k-loop).
// LDS: thetaL[(j*32 + c)*2 + n] = Theta[j][n*32 + c]  (float2 per (j, c)).
template <int NB, int KU, int EXP = 0>
__global__ __launch_bounds__(kBlock, 2) void lab_lik_batched_f32_kernel(const float *__restrict__ X, TileGeom g,
                                                                    const int32_t *__restrict__ tile_valid,
                                                                    const float *__restrict__ theta_dev, int d_pad,
                                                                    double *__restrict__ partials, int64_t pstride,
                                                                    int K)
{
  typedef Vec16<float>::type V4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *thetaL = reinterpret_cast<float *>(smem);                    // d_pad * 64 floats
  double *red = reinterpret_cast<double *>(smem + (size_t)d_pad * kBatchCols * sizeof(float)); // [4 waves][NB][64 lanes]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile = blockIdx.x;
  const int valid = tile_valid[tile];
  const int d = g.d;

  // stage Theta (already in B-operand order on the device) into LDS
  for (int i = threadIdx.x * 4; i < d_pad * kBatchCols; i += kBlock * 4)
    *reinterpret_cast<V4 *>(thetaL + i) = *reinterpret_cast<const V4 *>(theta_dev + i);
  __syncthreads();

  const int arow = lane & 31; // MFMA A-row supplied by this lane
  const int kk = lane >> 5;   // which of the two columns of a k-step
  const float *tbase = X + tile * g.tile_stride;
  float sum[NB];
#pragma unroll
  for (int n = 0; n < NB; n++) sum[n] = 0.f;

#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
    const int row0 = wave * 256 + pass * 128; // first tile row of this pass
    f32x16 acc[4][NB];
    lab_batched_f32_pass<NB, KU, EXP>(tbase + row0, g.col_stride, d, thetaL, arow, kk, acc);

    // epilogue: accumulator (m, n, r) of lane l is tile row  row0 + 4*i + m  with
    // i = (r&3) + 8*(r>>2) + 4*(l>>5), chain n*32 + (l&31)
    // Product form (SoftplusAccF): a lane's 64 elements per chain block and pass share one log.
    SoftplusAccF sp[NB];
    auto each = [&](auto full_tag, auto fn) { // only a shard's last tile is ragged: the full case has no predicates
      constexpr bool kFull = decltype(full_tag)::value;
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * kk;
          const bool ok = kFull || (row0 + 4 * i + m < valid);
#pragma unroll
          for (int n = 0; n < NB; n++)
            if (ok) fn(n, acc[m][n][r]);
        }
    };
    auto add = [&](int n, float x) {
      if (EXP & 1)
        sp[n].lin -= x;
      else
        sp[n].add(x);
    };
    if (row0 + 128 <= valid)
      each(std::true_type{}, add);
    else
      each(std::false_type{}, add);
    if (!(EXP & 1)) {
      bool look = false;
#pragma unroll
      for (int n = 0; n < NB; n++) look |= sp[n].lin > SoftplusAccF::kOverflow;
      if (look) each(std::false_type{}, [&](int n, float x) { sp[n].flag_overflow(x); });
    }
#pragma unroll
    for (int n = 0; n < NB; n++) sum[n] += (EXP & 1) ? -sp[n].lin : sp[n].neg_sum();
  }

  // per-chain sum over the workgroup: [wave][n][lane] -> thread per chain, fixed order
#pragma unroll
  for (int n = 0; n < NB; n++) red[(wave * NB + n) * 64 + lane] = (double)sum[n];
  __syncthreads();
  if ((int)threadIdx.x < NB * 32) {
    const int n = threadIdx.x >> 5, c = threadIdx.x & 31;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < 4; w++) s += red[(w * NB + n) * 64 + c] + red[(w * NB + n) * 64 + c + 32];
    const int chain = n * 32 + c;
    if (chain < K) partials[(int64_t)chain * pstride + tile] = s;
  }
}

// ------------------------------------------------------------------- f64
// f64 tile = 512 rows; wave w owns rows [128w, 128w+128) in two passes of 64
// rows (16 A-rows x 4 row sets: two double2 loads per lane per k-step).
// NB = number of 16-chain blocks (1..4).  One k-step = 4 columns.
// LDS: thetaL[(j*16 + c)*4 + n] = Theta[j][n*16 + c]  (4 doubles per (j, c)); the per-chain reduction buffer
// [4 waves][NB][64 lanes] reuses that region once the last MFMA has read Theta (theta_doubles >= 4*NB*64).
// EXP & 8: the product-form epilogue of mhlik_softplus.h (one log per lane and chain block instead of one per element).
template <int NB, int KU, int EXP = 0>
__global__ __launch_bounds__(kBlock, 2) void lab_lik_batched_f64_kernel(const double *__restrict__ X, TileGeom g,
                                                                    const int32_t *__restrict__ tile_valid,
                                                                    const double *__restrict__ theta_dev, int d_pad,
                                                                    double *__restrict__ partials, int64_t pstride,
                                                                    int K, int theta_doubles)
{
  typedef Vec16<double>::type V2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *thetaL = reinterpret_cast<double *>(smem);                  // d_pad * 64 doubles (region: theta_doubles)
  double *red = thetaL;                                               // [4 waves][NB][64 lanes] (product form: twice), after the last pass
  (void)theta_doubles;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile = blockIdx.x;
  const int valid = tile_valid[tile];
  const int d = g.d;

  for (int i = threadIdx.x * 2; i < d_pad * kBatchCols; i += kBlock * 2)
    *reinterpret_cast<V2 *>(thetaL + i) = *reinterpret_cast<const V2 *>(theta_dev + i);
  __syncthreads();

  const int arow = lane & 15; // MFMA A-row supplied by this lane
  const int kk = lane >> 4;   // which of the four columns of a k-step
  const double *tbase = X + tile * g.tile_stride;
  double sum[NB];
  mhlik_softplus_acc pacc[NB]; // EXP & 8: 32 elements per lane and chain block over the two passes
#pragma unroll
  for (int n = 0; n < NB; n++) {
    sum[n] = 0.0;
    mhlik_softplus_acc_init(&pacc[n]);
  }

  const int nfull = d / (4 * KU);
  V2 n0[KU], n1[KU]; // the k-batch in flight
  // the first k-batch of a pass: requested before the previous pass's epilogue, so that it lands under it
  auto first_batch = [&](int pass) {
    if (nfull > 0 && !(EXP & 2)) {
      const double *rb = tbase + wave * 128 + pass * 64 + 2 * arow;
#pragma unroll
      for (int u = 0; u < KU; u++) {
        const double *col = rb + (int64_t)(4 * u + kk) * g.col_stride;
        n0[u] = load16<double, true>(col);
        n1[u] = load16<double, true>(col + 32);
      }
    }
  };
  if (EXP & 2) {
#pragma unroll
    for (int u = 0; u < KU; u++) n0[u] = n1[u] = V2{1.0 + lane, 2.0 + u};
  }
  first_batch(0);

#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
    const int row0 = wave * 128 + pass * 64;
    const double *rbase = tbase + row0 + 2 * arow; // rows 2i, 2i+1 and 32+2i, 32+2i+1
    f64x4 acc[4][NB];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int n = 0; n < NB; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[m][n][r] = 0.0;

    int j0 = 0;
    V2 a0[KU], a1[KU];
    for (int bt = 0; bt < nfull; bt++, j0 += 4 * KU) {
#pragma unroll
      for (int u = 0; u < KU; u++) {
        a0[u] = n0[u];
        a1[u] = n1[u];
      }
      if (!(EXP & 2) && bt + 1 < nfull) {
#pragma unroll
        for (int u = 0; u < KU; u++) {
          const double *col = rbase + (int64_t)(j0 + 4 * KU + 4 * u + kk) * g.col_stride;
          n0[u] = load16<double, true>(col);
          n1[u] = load16<double, true>(col + 32);
        }
      }
#pragma unroll
      for (int u = 0; u < KU; u++) {
        const double *bp = thetaL + ((j0 + 4 * u + kk) * 16 + arow) * 4;
        double b[4];
#pragma unroll
        for (int n = 0; n < 4; n++) b[n] = (n < NB) ? bp[n] : 0.0;
#pragma unroll
        for (int n = 0; n < NB; n++) {
          acc[0][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u][0], b[n], acc[0][n], 0, 0, 0);
          acc[1][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u][1], b[n], acc[1][n], 0, 0, 0);
          acc[2][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u][0], b[n], acc[2][n], 0, 0, 0);
          acc[3][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u][1], b[n], acc[3][n], 0, 0, 0);
        }
      }
    }
    for (; j0 < d; j0 += 4) {
      const int colj = j0 + kk;
      V2 a0 = {0.0, 0.0}, a1 = {0.0, 0.0};
      if (colj < d) {
        const double *col = rbase + (int64_t)colj * g.col_stride;
        a0 = load16<double, true>(col);
        a1 = load16<double, true>(col + 32);
      }
      const double *bp = thetaL + (colj * 16 + arow) * 4;
#pragma unroll
      for (int n = 0; n < NB; n++) {
        const double b = bp[n];
        acc[0][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], b, acc[0][n], 0, 0, 0);
        acc[1][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], b, acc[1][n], 0, 0, 0);
        acc[2][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], b, acc[2][n], 0, 0, 0);
        acc[3][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], b, acc[3][n], 0, 0, 0);
      }
    }

    if (pass == 0) first_batch(1);
    __builtin_amdgcn_sched_barrier(0); // keep the request ahead of the epilogue

    // accumulator (m, n, r) of lane l: A-row i = (l>>4) + 4*r, chain n*16 + (l&15);
    // tile row = row0 + (m>>1)*32 + 2*i + (m&1).  Only the last tile of a shard is ragged: the full case is
    // straight-line code with no per-element predicate.
    const bool full = row0 + 64 <= valid;
    mhlik_softplus_acc before[NB];
#pragma unroll
    for (int n = 0; n < NB; n++) before[n] = pacc[n];
    auto epilogue = [&](auto full_tag) {
      constexpr bool kFull = decltype(full_tag)::value;
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = kk + 4 * r;
          const bool ok = kFull || (row0 + (m >> 1) * 32 + 2 * i + (m & 1) < valid);
#pragma unroll
          for (int n = 0; n < NB; n++)
            if (ok) {
              if constexpr ((EXP & 8) != 0)
                mhlik_softplus_acc_add(&pacc[n], acc[m][n][r]);
              else
                sum[n] += (EXP & 1) ? acc[m][n][r] : (EXP & 4) ? neg_log1pexp(acc[m][n][r]) : mhlik_neg_softplus(acc[m][n][r]);
            }
        }
    };
    if (full)
      epilogue(std::true_type{});
    else
      epilogue(std::false_type{});
    if constexpr ((EXP & 8) != 0) {
      // the fast adds leave out the tests for the reference's overflow to -inf above 709.78 and for arguments
      // beyond the range reduction's reach: a group that looks suspect is redone from its state before the pass
      // while its elements are still in the accumulators
#pragma unroll
      for (int n = 0; n < NB; n++)
        if (mhlik_softplus_acc_suspect(&pacc[n])) {
          pacc[n] = before[n];
#pragma unroll
          for (int m = 0; m < 4; m++)
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (full || (row0 + (m >> 1) * 32 + 2 * (kk + 4 * r) + (m & 1) < valid))
                mhlik_softplus_acc_add_safe(&pacc[n], acc[m][n][r]);
        }
    }
  }
  __syncthreads(); // every wave is done with Theta: the reduction buffer takes its place
  if constexpr ((EXP & 8) != 0) {
    // product form: a chain's 16 lane groups are merged first, one log per (tile, chain)
    double *red_prod = red + 4 * NB * 64;
#pragma unroll
    for (int n = 0; n < NB; n++) {
      red[(wave * NB + n) * 64 + lane] = pacc[n].lin;
      red_prod[(wave * NB + n) * 64 + lane] = pacc[n].prod;
    }
    __syncthreads();
    if ((int)threadIdx.x < NB * 16) {
      const int n = threadIdx.x >> 4, c = threadIdx.x & 15;
      mhlik_softplus_acc t;
      mhlik_softplus_acc_init(&t);
#pragma unroll
      for (int w = 0; w < 4; w++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const mhlik_softplus_acc e{red[(w * NB + n) * 64 + c + 16 * q], red_prod[(w * NB + n) * 64 + c + 16 * q]};
          mhlik_softplus_acc_merge(&t, &e);
        }
      const int chain = n * 16 + c;
      if (chain < K) partials[(int64_t)chain * pstride + tile] = mhlik_softplus_acc_neg_sum(&t);
    }
    return;
  }
#pragma unroll
  for (int n = 0; n < NB; n++) red[(wave * NB + n) * 64 + lane] = sum[n];
  __syncthreads();
  if ((int)threadIdx.x < NB * 16) {
    const int n = threadIdx.x >> 4, c = threadIdx.x & 15;
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int q = 0; q < 4; q++) s += red[(w * NB + n) * 64 + c + 16 * q];
    const int chain = n * 16 + c;
    if (chain < K) partials[(int64_t)chain * pstride + tile] = s;
  }
}

} // namespace mhlik
