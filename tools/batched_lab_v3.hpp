// batched_lab_v3.hpp -- experiment: the f32 K-chain pass with the epilogue of
// one 64-row half-pass issued between the MFMAs of the next one (two ping-pong
// accumulator sets of 64 VGPRs, 8-byte loads = 2 rows per lane).  Needs
// d == 16 * KU (8 stages of one KU-step batch each).  Lab only.
#pragma once
#include "mhlik_batched.hpp"

namespace mhlik {

typedef float f32x2v __attribute__((ext_vector_type(2)));

template <int KU, int HINT>
__global__ __launch_bounds__(kBlock, 2) void lik_batched3_f32_kernel(const float *__restrict__ X, TileGeom g,
                                                                     const int32_t *__restrict__ tile_valid,
                                                                     const float *__restrict__ theta_dev, int d_pad,
                                                                     double *__restrict__ partials, int64_t pstride,
                                                                     int K)
{
  typedef Vec16<float>::type V4;
  constexpr int NB = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *thetaL = reinterpret_cast<float *>(smem);
  double *red = reinterpret_cast<double *>(smem + (size_t)d_pad * kBatchCols * sizeof(float));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile = blockIdx.x;
  const int valid = tile_valid[tile];
  for (int i = threadIdx.x * 4; i < d_pad * kBatchCols; i += kBlock * 4)
    *reinterpret_cast<V4 *>(thetaL + i) = *reinterpret_cast<const V4 *>(theta_dev + i);
  __syncthreads();

  const int arow = lane & 31, kk = lane >> 5;
  // uniform base + 32-bit per-lane byte offset: one scalar address per load instead of a 64-bit VGPR pair
  const float *ubase = X + tile * g.tile_stride + wave * 256; // + 64*hp + col*col_stride   (wave-uniform)
  const int64_t cs = g.col_stride;
  const unsigned loff = (unsigned)((2 * arow + kk * cs) * sizeof(float));
  float sum[NB] = {0.f, 0.f};
  f32x16 accA[2][NB], accB[2][NB];
  f32x2v buf0[KU], buf1[KU];

  auto load_batch = [&](f32x2v (&buf)[KU], int hp, int jb) {
#pragma unroll
    for (int u = 0; u < KU; u++)
      buf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x2v *>(
          reinterpret_cast<const char *>(ubase + 64 * hp + (int64_t)(jb + 2 * u) * cs) + loff));
  };
  auto zero = [&](f32x16 (&acc)[2][NB]) {
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int n = 0; n < NB; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
  };
  auto mfma_batch = [&](const f32x2v (&buf)[KU], int jb, f32x16 (&acc)[2][NB]) {
#pragma unroll
    for (int u = 0; u < KU; u++) {
      const f32x2 b = *reinterpret_cast<const f32x2 *>(thetaL + ((jb + 2 * u + kk) * 32 + arow) * 2);
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < NB; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[u][m], b[n], acc[m][n], 0, 0, 0);
    }
  };
  // elements [8s, 8s+8) of a finished half-pass: element e -> (m, n, r) = (e>>5, (e>>4)&1, e&15)
  auto epi_slice = [&](const f32x16 (&acc)[2][NB], int s, int hp) {
    const int row0 = wave * 256 + 64 * hp;
    const bool full = row0 + 64 <= valid;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int e = 8 * s + q;
      const int m = e >> 5, n = (e >> 4) & 1, r = e & 15;
      const float t = neg_log1pexp_fast(acc[m][n][r]);
      const int i = (r & 3) + 8 * (r >> 2) + 4 * kk;
      sum[n] += (full || row0 + 2 * i + m < valid) ? t : 0.f;
    }
  };
  // one half-pass into `acc` while the epilogue of `prev` (half-pass hp-1) is issued stage by stage
  auto half_pass = [&](f32x16 (&acc)[2][NB], const f32x16 (&prev)[2][NB], int hp, bool with_epi, bool more) {
    zero(acc);
#pragma unroll
    for (int s = 0; s < 8; s++) {
      // stage s: batch s of this half-pass sits in buf[s&1]; prefetch the next batch (or batch 0 of the next half-pass)
      if (s & 1) {
        if (s < 7) load_batch(buf0, hp, 2 * KU * (s + 1));
        else if (more) load_batch(buf0, hp + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_batch(buf1, 2 * KU * s, acc);
      } else {
        load_batch(buf1, hp, 2 * KU * (s + 1));
        __builtin_amdgcn_sched_barrier(0);
        mfma_batch(buf0, 2 * KU * s, acc);
      }
      if (with_epi) epi_slice(prev, s, hp - 1);
      if (HINT) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++) {
          __builtin_amdgcn_sched_group_barrier(0x008, (4 * KU) / 8, 0); // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);            // VALU
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  load_batch(buf0, 0, 0);
  half_pass(accA, accB, 0, false, true);
  half_pass(accB, accA, 1, true, true);
  half_pass(accA, accB, 2, true, true);
  half_pass(accB, accA, 3, true, false);
#pragma unroll
  for (int s = 0; s < 8; s++) epi_slice(accB, s, 3);

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

} // namespace mhlik
