// batched_lab_v2.hpp -- EXPERIMENTAL schedules of the f32 K-chain kernel, timed
// by tools/batched_lab.hip; not part of the library.
//
// Persistent workgroups (Theta staged once, static round-robin over half-tile
// units), two register buffers one batch of KU k-steps ahead, next unit's
// first batch issued before the epilogue, LDS-only barrier, scalar-base
// addressing; optional AccVGPR accumulators (EXP & 16), LDS-DMA staging of the
// A operand (EXP & 32), one wave per SIMD (OCC = 1).  Measured on MI355X at
// N=1e7, d=128, K=64 (profiles/r01_batched_lab.txt): 102-110 TF for every
// variant against 102 TF for the shipped kernel -- none of the schedules buys
// more than 8 %, so the library keeps the simpler one-tile-per-workgroup kernel.
// The ablations there show why: with the k-loop's loads issued but never
// waited for, the same loop runs at 140 TF on constant operands; the chip
// holds a lower clock on changing operands (DVFS, MI355X_MICROARCH.md).
#pragma once

#include "mhlik_batched.hpp"

namespace mhlik {

// Tell the compiler a byte offset is wave-uniform (it is: derived from
// blockIdx and kernel arguments), so that loads take the scalar-base + 32-bit
// lane-offset form and spend no vector registers or vector ALU work on
// addresses.  The offset, not the pointer, goes through readfirstlane: the
// pointer keeps its global address space.
__device__ __forceinline__ int64_t uniform_i64(int64_t v)
{
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Workgroup barrier that orders LDS traffic only (lgkmcnt), leaving global
// loads in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------- f32
// LDS: thetaL[(j*32 + c)*2 + n] = Theta[j][n*32 + c], j < d_pad (zero for j >= d)
//      red[2][4 waves][NB][64 lanes] doubles
template <int NB, int KU, int EXP = 0, int OCC = 2>
__global__ __launch_bounds__(kBlock, OCC) void lik_batched2_f32_kernel(const float *__restrict__ X, TileGeom g,
                                                                     const int32_t *__restrict__ tile_valid,
                                                                     const float *__restrict__ theta_dev, int d_pad,
                                                                     double *__restrict__ partials, int64_t pstride,
                                                                     int K, int64_t nunits)
{
  typedef Vec16<float>::type V4;
  constexpr int BC = 2 * KU; // columns per batch
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *thetaL = reinterpret_cast<float *>(smem);
  double *red = reinterpret_cast<double *>(smem + (size_t)d_pad * kBatchCols * sizeof(float));

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = g.d;
  for (int i = threadIdx.x * 4; i < d_pad * kBatchCols; i += kBlock * 4)
    *reinterpret_cast<V4 *>(thetaL + i) = *reinterpret_cast<const V4 *>(theta_dev + i);
  __syncthreads();

  const int arow = lane & 31, kk = lane >> 5;
  // full batches run in pairs through the two register buffers; what is left
  // of d (fewer than 2*BC columns) goes through the one-k-step-at-a-time tail
  const int npair = d / (2 * BC);
  const int tail0 = npair * 2 * BC;
  const int dlast = d - 1;
  // lane part of every address, in bytes: the 4 rows of this lane inside the
  // wave's 128 rows, plus one column for the upper half-wave (kk = 1)
  const uint32_t loff0 = (uint32_t)(wave * 128 + 4 * arow) * 4u;
  const uint32_t loff1 = loff0 + (uint32_t)kk * (uint32_t)g.col_stride * 4u;
  const int64_t pair_bytes = 2 * g.col_stride * (int64_t)sizeof(float);
  const float *bl = thetaL + (kk * 32 + arow) * 2; // + j*64 per column pair start

  const char *Xc = reinterpret_cast<const char *>(X);
  // byte offset of a unit's first element
  auto unit_base = [&](int64_t u) -> int64_t {
    if (EXP & 4) u &= 1023; // lab: keep the footprint inside the caches
    return ((u >> 1) * g.tile_stride + (u & 1) * 512) * (int64_t)sizeof(float);
  };
  // loads of full batch b of the unit at ub: wave-uniform offset + lane offset
  auto load_batch = [&](V4(&buf)[KU], int64_t ub, int b) {
#pragma unroll
    for (int q = 0; q < KU; q++) {
      const char *p = Xc + uniform_i64(ub + ((int64_t)b * KU + q) * pair_bytes);
      buf[q] = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(p + loff1));
    }
  };

  // lab (EXP & 32): LDS-DMA staging -- loads land in a per-wave LDS ring
  // (two batches of KU KiB), A fragments come back by ds_read_b128
  float *stage = reinterpret_cast<float *>(red + 2 * 4 * NB * 64) + wave * (2 * KU * 256);
  auto glds_batch = [&](int slot, int64_t ub, int b) {
#pragma unroll
    for (int q = 0; q < KU; q++) {
      const char *p = Xc + uniform_i64(ub + ((int64_t)b * KU + q) * pair_bytes) + loff1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                       (__attribute__((address_space(3))) void *)(stage + (slot * KU + q) * 256), 16, 0, 0);
    }
  };

  int64_t u = blockIdx.x;
  V4 bufA[KU], bufB[KU];
  if (u < nunits && npair > 0) {
    if (EXP & 32)
      glds_batch(0, unit_base(u), 0);
    else
      load_batch(bufA, unit_base(u), 0);
  }
  int parity = 0;
  for (; u < nunits; u += gridDim.x) {
    const int64_t ub = unit_base(u);
    const int64_t un = u + gridDim.x;
    f32x16 acc[4][NB];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int n = 0; n < NB; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    auto mfma_step = [&](const V4 &a, int colu) {
      const f32x2 bv = *reinterpret_cast<const f32x2 *>(bl + colu * 64);
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < NB; n++) {
          if (EXP & 16) // lab: accumulators pinned to AccVGPRs
            asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[m][n]) : "v"(a[m]), "v"(bv[n]));
          else
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[n], acc[m][n], 0, 0, 0);
        }
    };
    auto mfma_batch = [&](const V4(&buf)[KU], int b) {
#pragma unroll
      for (int q = 0; q < KU; q++) mfma_step(buf[q], b * BC + 2 * q);
    };
    // A is consumed while B is in flight and vice versa; the scheduling
    // barriers keep the loads where they are written, ahead of the MFMAs
    if (EXP & 32) {
      auto mfma_lds = [&](int slot, int b) {
#pragma unroll
        for (int q = 0; q < KU; q++) {
          const V4 a = *reinterpret_cast<const V4 *>(stage + (slot * KU + q) * 256 + lane * 4);
          mfma_step(a, b * BC + 2 * q);
        }
      };
      for (int pr = 0; pr < npair; pr++) {
        const int b = 2 * pr;
        glds_batch(1, ub, b + 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KU) : "memory");
        mfma_lds(0, b);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (pr + 1 < npair)
          glds_batch(0, ub, b + 2);
        else if (un < nunits)
          glds_batch(0, unit_base(un), 0);
        if (pr + 1 < npair || un < nunits)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KU) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mfma_lds(1, b + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else
    for (int pr = 0; pr < npair; pr++) {
      const int b = 2 * pr;
      load_batch(bufB, ub, b + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_batch(bufA, b);
      __builtin_amdgcn_sched_barrier(0);
      if (EXP & 8) { // lab: loads issued, never waited for inside the loop
        load_batch(bufB, ub, b);
        __builtin_amdgcn_sched_barrier(0);
        mfma_batch(bufA, b + 1);
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if (pr + 1 < npair)
        load_batch(bufA, ub, b + 2);
      else if (un < nunits)
        load_batch(bufA, unit_base(un), 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_batch(bufB, b + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int colu = tail0; colu < d; colu += 2) { // a column >= d: clamped load, zero Theta
      const char *p = Xc + uniform_i64(ub + (int64_t)(colu < dlast ? colu : dlast) * (pair_bytes / 2));
      const V4 a = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(p + (colu < dlast ? loff1 : loff0)));
      mfma_step(a, colu);
    }

    // epilogue: accumulator (m, n, r) of lane l is unit row  wave*128 + 4*i + m,
    // i = (r&3) + 8*(r>>2) + 4*(l>>5), chain n*32 + (l&31)
    if (EXP & 8) {
#pragma unroll
      for (int q = 0; q < KU; q++) acc[0][0][q] += bufB[q][0];
    }
    const int valid = tile_valid[u >> 1] - (int)(u & 1) * 512;
    const int wrow0 = wave * 128;
    float sum[NB];
#pragma unroll
    for (int n = 0; n < NB; n++) sum[n] = 0.f;
    if (wrow0 + 128 <= valid) {
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < NB; n++)
#pragma unroll
          for (int r = 0; r < 16; r++) sum[n] += (EXP & 1) ? acc[m][n][r] : neg_log1pexp_fast(acc[m][n][r]);
    } else {
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * kk;
          const bool ok = wrow0 + 4 * i + m < valid;
#pragma unroll
          for (int n = 0; n < NB; n++) sum[n] += ok ? neg_log1pexp_fast(acc[m][n][r]) : 0.f;
        }
    }
    double *rp = red + parity * (4 * NB * 64);
#pragma unroll
    for (int n = 0; n < NB; n++) rp[(wave * NB + n) * 64 + lane] = (double)sum[n];
    // LDS-only rendezvous: __syncthreads() would also drain vmcnt, i.e. wait
    // for the next unit's prefetched loads
    lds_barrier();
    if ((int)threadIdx.x < NB * 32) {
      const int n = threadIdx.x >> 5, c = threadIdx.x & 31;
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < 4; w++) s += rp[(w * NB + n) * 64 + c] + rp[(w * NB + n) * 64 + c + 32];
      const int chain = n * 32 + c;
      if (chain < K) partials[(int64_t)chain * pstride + u] = s;
    }
    parity ^= 1;
  }
}

} // namespace mhlik
