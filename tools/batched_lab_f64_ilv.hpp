// batched_lab_f64_ilv.hpp (lab, a negative result: see DESIGN.md section 3.4) -- the K-chain double pass with the epilogue of one half-pass issued between the MFMAs
// of the next.
//
// On gfx950 the MFMA work of one wave and the VALU work of another wave of the same SIMD do not overlap; inside one
// wave independent VALU instructions do issue under a running MFMA (tools/check_f64_overlap.hip).  lik_batched_f64_kernel
// finishes a 64-row pass and then evaluates -log(1+e^x) on its 128 accumulator registers with the matrix pipe idle
// (1.1 of 3.9 ms at N = 10M, d = 128, K = 64).  Here a wave's share of the tile is cut into four half-passes of 32
// rows with two accumulator sets of 64 registers (the same 128 in total): while the MFMAs of half-pass h fill one
// set, the softplus of half-pass h-1 drains the other, one or two accumulator elements per k-step, in the same
// straight-line block as that k-step's eight MFMAs.  Element order and therefore every sum is that of
// lik_batched_f64_kernel: the partials are the same bits.
//
// Written for NB = 4 (K > 48) and a k-step count NK = d_pad/4 that divides the 32 elements of a half-pass evenly
// into the k-steps (NK = 32: d in 125..128, NK = 16: d in 61..64); other shapes take lik_batched_f64_kernel.
#pragma once

#include "mhlik_batched.hpp"

namespace mhlik {

template <int NK, int EXP = 0>
__global__ __launch_bounds__(kBlock, 2) void lik_batched_f64_ilv_kernel(const double *__restrict__ X, TileGeom g,
                                                                        const int32_t *__restrict__ tile_valid,
                                                                        const double *__restrict__ theta_dev, int d_pad,
                                                                        double *__restrict__ partials, int64_t pstride,
                                                                        int K)
{
  constexpr int NB = 4;
  constexpr int E = 8 * NB;      // accumulator elements of a half-pass per lane
  constexpr int EPK = E / NK;    // elements drained per k-step
  static_assert(EPK * NK == E, "k-steps must divide the elements of a half-pass");
  constexpr int DEPTH = 4;       // k-steps of rows in flight
  typedef Vec16<double>::type V2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *thetaL = reinterpret_cast<double *>(smem);                  // d_pad * 64 doubles
  double *red = thetaL + (size_t)d_pad * kBatchCols;                  // [4 waves][NB][64 lanes]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile = blockIdx.x;
  const int valid = tile_valid[tile];
  const int d = g.d;
  for (int i = threadIdx.x * 2; i < d_pad * kBatchCols; i += kBlock * 2)
    *reinterpret_cast<V2 *>(thetaL + i) = *reinterpret_cast<const V2 *>(theta_dev + i);
  __syncthreads();

  const int arow = lane & 15, kk = lane >> 4;
  const double *tbase = X + tile * g.tile_stride;
  const int cs = (int)g.col_stride;
  double sum[NB];
#pragma unroll
  for (int n = 0; n < NB; n++) sum[n] = 0.0;

  // element e of a half-pass whose first row is prow0: (mm, r, n) in the order lik_batched_f64_kernel adds them;
  // accumulator (mm, n, r) of lane l is tile row prow0 + 2*((l>>4) + 4r) + mm, chain 16n + (l&15)
  auto drain = [&](const f64x4 (&A)[2][NB], int e, int prow0) {
    const int n = e % NB, r = (e / NB) % 4, mm = e / (4 * NB);
    const bool ok = prow0 >= 0 && prow0 + 2 * (kk + 4 * r) + mm < valid;
    const double x = A[mm][n][r];
    const double v = (EXP & 1) ? x : mhlik_neg_softplus(x);
    sum[n] += ok ? v : 0.0; // + 0.0 leaves the bits of a sum that started at +0.0
  };
  // MFMAs of half-pass hp into cur, the elements of the previous half-pass (first row prow0, < 0: none) out of prev
  auto halfpass = [&](int hp, f64x4 (&cur)[2][NB], const f64x4 (&prev)[2][NB], int prow0) {
    const double *rbase = tbase + wave * 128 + 32 * hp + 2 * arow; // rows 2i, 2i+1 of the half-pass
#pragma unroll
    for (int mm = 0; mm < 2; mm++)
#pragma unroll
      for (int n = 0; n < NB; n++) cur[mm][n] = f64x4{0.0, 0.0, 0.0, 0.0};
    auto load = [&](int ks) { // column 4ks + kk, clamped to the last real one (its Theta row is zero)
      const int j = 4 * ks + kk;
      return load16<double, true>(rbase + (j < d ? j : d - 1) * cs);
    };
    V2 buf[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; u++) buf[u] = load(u < NK ? u : NK - 1);
#pragma unroll
    for (int ks = 0; ks < NK; ks++) {
      const V2 a = buf[ks % DEPTH];
      if (!(EXP & 2) && ks + DEPTH < NK) buf[ks % DEPTH] = load(ks + DEPTH);
      const double *bp = thetaL + ((4 * ks + kk) * 16 + arow) * 4;
      double b[NB];
#pragma unroll
      for (int n = 0; n < NB; n++) b[n] = bp[n];
#pragma unroll
      for (int n = 0; n < NB; n++) {
        cur[0][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[n], cur[0][n], 0, 0, 0);
        cur[1][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[n], cur[1][n], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < EPK; q++) drain(prev, ks * EPK + q, prow0);
      if (EXP & 8) { // scheduling hint: one MFMA, then a share of the k-step's VALU work
#pragma unroll
        for (int i = 0; i < 2 * NB; i++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6 * EPK, 0);
        }
      }
    }
  };

  f64x4 P[2][NB], Q[2][NB];
#pragma unroll
  for (int mm = 0; mm < 2; mm++)
#pragma unroll
    for (int n = 0; n < NB; n++) Q[mm][n] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int row0 = wave * 128;
#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
    halfpass(2 * pass, P, Q, pass == 0 ? -1 : row0 + 32); // first time round there is nothing to drain
    halfpass(2 * pass + 1, Q, P, row0 + 64 * pass);
  }
#pragma unroll
  for (int e = 0; e < E; e++) drain(Q, e, row0 + 96);

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
