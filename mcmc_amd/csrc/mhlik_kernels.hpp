// mhlik_kernels.hpp -- CDNA4 (gfx950) device code of the MH likelihood path.
//
// One streaming pass over the HBM-resident design matrix replaces the
// reference's cublas{D,S}gemv + reduceRegr2 + reduce2 passes + D2H tail sum
// (src/mcmc_gpu/mcmc_gpu.cu:177-257, src/reduction/reduction_kernel.cu:140-280):
//
//   tile kernel   : each 256-thread workgroup owns one row tile.  Every lane
//                   holds VEC adjacent rows per 16-byte load (2 doubles /
//                   4 floats), so a wave reads 1 KiB contiguous per column.
//                   Row dots accumulate column by column in the gslcblas
//                   order; then -log(1+exp(.)) per row, a wave64 DPP
//                   reduction, four wave sums through LDS, one partial per
//                   tile.
//   segment kernel: one wave per row segment adds that segment's tile
//                   partials in a fixed order and publishes the result to
//                   pinned host memory with a completion word the host polls.
//
// Every sum has a fixed shape that depends only on (tile_rows, segments), not
// on the grid, the GPU count or timing: results are bitwise reproducible.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mhlik {

constexpr int kBlock = 256;      // threads per workgroup (4 waves)
constexpr int kWave = 64;
constexpr int kMaxKargCols = 256; // theta travels in the kernarg segment up to this many columns

template <typename T> struct Vec16;
template <> struct Vec16<double> {
  typedef double type __attribute__((ext_vector_type(2)));
  static constexpr int N = 2;
};
template <> struct Vec16<float> {
  typedef float type __attribute__((ext_vector_type(4)));
  static constexpr int N = 4;
};

// theta by value: it rides in the kernarg segment of the launch, so there is
// no per-iteration H2D copy (the reference does one cudaMemcpy per call,
// mcmc_gpu.cu:231).  Scalar loads fetch it; it never occupies VGPRs.
template <typename T> struct ThetaArg {
  T v[kMaxKargCols];
};

// Where a tile lives:  element (row r of tile t, column j) is at
//   X[t * tile_stride + j * col_stride + r]
// TILED:    tile_stride = tile_rows * d, col_stride = tile_rows
// COLMAJOR: tile_stride = tile_rows,     col_stride = padded leading dimension
struct TileGeom {
  int64_t tile_stride;
  int64_t col_stride;
  int32_t d;
};

// ---------------------------------------------------------------- reductions

template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_f64(double x)
{
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes of a wave with DPP moves only (no LDS, no permute
// network): row_shr 1,2,4,8 scans each 16-lane row, row_bcast:15 / :31 carry
// the row totals across.  Lane 63 ends up with the wave total.  Lanes with no
// source (or rows masked off) read 0.0 and add nothing.
__device__ __forceinline__ double wave_sum_lane63(double v)
{
  v += dpp_f64<0x111, 0xf>(v); // row_shr:1
  v += dpp_f64<0x112, 0xf>(v); // row_shr:2
  v += dpp_f64<0x114, 0xf>(v); // row_shr:4
  v += dpp_f64<0x118, 0xf>(v); // row_shr:8
  v += dpp_f64<0x142, 0xa>(v); // row_bcast:15 -> rows 1,3
  v += dpp_f64<0x143, 0xc>(v); // row_bcast:31 -> rows 2,3
  return v;
}

// Workgroup sum: wave totals staged through LDS, added in wave order by
// thread 0.  Returns the total in thread 0 (other threads: unspecified).
__device__ __forceinline__ double block_sum(double v, double *lds /* kBlock/kWave doubles */)
{
  v = wave_sum_lane63(v);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == kWave - 1) lds[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kBlock / kWave; w++) s += lds[w];
  }
  return s;
}

// ------------------------------------------------------------ per-row math

// -log(1 + exp(x)) exactly as the reference writes it (mcmc_cpu.c:440,
// reduction_kernel.cu:150): it overflows to -inf for x > 709.78 (double) /
// 88.7 (float) like the reference does; no "stable softplus" rewrite.
__device__ __forceinline__ double neg_log1pexp(double x) { return -log(1.0 + exp(x)); }
__device__ __forceinline__ float neg_log1pexp(float x) { return -logf(1.0f + expf(x)); }

// acc + t*x with the product rounded before the add (double: gslcblas on
// baseline x86-64 has no FMA, and the row dots must match it bit for bit);
// float: a fused chain, which is also what the f32 matrix cores compute.
__device__ __forceinline__ double madd(double acc, double t, double x)
{
#pragma clang fp contract(off)
  const double p = t * x;
  return acc + p;
}
__device__ __forceinline__ float madd(float acc, float t, float x) { return fmaf(t, x, acc); }

template <typename T, bool NT> __device__ __forceinline__ typename Vec16<T>::type load16(const T *p)
{
  typedef typename Vec16<T>::type V;
  if constexpr (NT)
    return __builtin_nontemporal_load(reinterpret_cast<const V *>(p));
  else
    return *reinterpret_cast<const V *>(p);
}

// ------------------------------------------------------------- tile kernel
//
// R row groups per lane (tile_rows = 256 * VEC * R), U columns in flight per
// step: R*U independent 16-byte loads per lane are issued before the first
// is consumed.  KARG: theta from the kernarg struct, else from `theta_g`.
// ROWDOTS: also store the row dots (debug / mvout).

// Columns [j0, j0+U) of one tile: issue all loads, then accumulate in column
// order.  Columns are always consumed in increasing j, whatever U is, so the
// row dots do not depend on the unroll factor.
template <typename T, int R, int U, bool NT, bool KARG>
__device__ __forceinline__ void tile_columns(const T *__restrict__ tbase, int64_t col_stride, uint32_t lane_row,
                                             int j0, const ThetaArg<T> &theta_k, const T *__restrict__ theta_g,
                                             T (&acc)[R][Vec16<T>::N])
{
  typedef typename Vec16<T>::type V;
  constexpr int VEC = Vec16<T>::N;
  constexpr int STEP = kBlock * VEC;
  V x[U][R];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const T *col = tbase + (int64_t)(j0 + u) * col_stride;
#pragma unroll
    for (int r = 0; r < R; r++) x[u][r] = load16<T, NT>(col + (lane_row + (uint32_t)(r * STEP)));
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    T t;
    if constexpr (KARG)
      t = theta_k.v[j0 + u];
    else
      t = theta_g[j0 + u];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
      for (int k = 0; k < VEC; k++) acc[r][k] = madd(acc[r][k], t, x[u][r][k]);
  }
}

// One tile's contribution: the row dots of the tile's rows against theta (columns in increasing order), then
// -log(1+exp(.)) of the valid rows summed per lane in (r, k) order.  The caller adds the lanes with block_sum.
// `theta_g` may point to global memory or LDS (KARG == false); KARG: theta from the kernarg struct.
template <typename T, int R, int U, bool NT, bool KARG, bool ROWDOTS>
__device__ __forceinline__ double tile_lane_sum(const T *__restrict__ tbase, int64_t col_stride, int d, int valid,
                                                const ThetaArg<T> &theta_k, const T *__restrict__ theta_g,
                                                T *__restrict__ rowdots_tile)
{
  constexpr int VEC = Vec16<T>::N;
  constexpr int STEP = kBlock * VEC; // rows covered by one load of the whole workgroup
  const uint32_t lane_row = threadIdx.x * VEC; // first of this lane's VEC rows in group 0

  T acc[R][VEC];
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int k = 0; k < VEC; k++) acc[r][k] = (T)0;

  int j0 = 0;
  for (; j0 + U <= d; j0 += U) tile_columns<T, R, U, NT, KARG>(tbase, col_stride, lane_row, j0, theta_k, theta_g, acc);
  // d % U leftover columns: halve the batch until nothing is left
  if constexpr (U >= 32)
    if (j0 + 16 <= d) { tile_columns<T, R, 16, NT, KARG>(tbase, col_stride, lane_row, j0, theta_k, theta_g, acc); j0 += 16; }
  if constexpr (U >= 16)
    if (j0 + 8 <= d) { tile_columns<T, R, 8, NT, KARG>(tbase, col_stride, lane_row, j0, theta_k, theta_g, acc); j0 += 8; }
  if constexpr (U >= 8)
    if (j0 + 4 <= d) { tile_columns<T, R, 4, NT, KARG>(tbase, col_stride, lane_row, j0, theta_k, theta_g, acc); j0 += 4; }
  if constexpr (U >= 4)
    if (j0 + 2 <= d) { tile_columns<T, R, 2, NT, KARG>(tbase, col_stride, lane_row, j0, theta_k, theta_g, acc); j0 += 2; }
  if constexpr (U >= 2)
    if (j0 + 1 <= d) { tile_columns<T, R, 1, NT, KARG>(tbase, col_stride, lane_row, j0, theta_k, theta_g, acc); j0 += 1; }

  double s = 0.0;
#pragma unroll
  for (int r = 0; r < R; r++) {
#pragma unroll
    for (int k = 0; k < VEC; k++) {
      const int row = (int)lane_row + r * STEP + k;
      if (row < valid) {
        s += (double)neg_log1pexp(acc[r][k]);
        if constexpr (ROWDOTS) rowdots_tile[row] = acc[r][k];
      }
    }
  }
  return s;
}

template <typename T, int R, int U, bool NT, bool KARG, bool ROWDOTS>
__global__ __launch_bounds__(kBlock) void lik_tile_kernel(const T *__restrict__ X, TileGeom g,
                                                          const int32_t *__restrict__ tile_valid,
                                                          ThetaArg<T> theta_k, const T *__restrict__ theta_g,
                                                          double *__restrict__ partials,
                                                          T *__restrict__ rowdots)
{
  constexpr int VEC = Vec16<T>::N;
  constexpr int STEP = kBlock * VEC;
  __shared__ double lds[kBlock / kWave];

  const int64_t tile = blockIdx.x;
  const int valid = tile_valid[tile];
  const T *tbase = X + tile * g.tile_stride;   // wave-uniform
  double s = tile_lane_sum<T, R, U, NT, KARG, ROWDOTS>(tbase, g.col_stride, g.d, valid, theta_k, theta_g,
                                                       ROWDOTS ? rowdots + tile * (int64_t)(STEP * R) : nullptr);
  s = block_sum(s, lds);
  if (threadIdx.x == 0) partials[tile] = s;
}

// ---------------------------------------------------------- segment kernel
//
// grid = local segments * nvec.  Workgroup (s, k) adds partials[k*pstride +
// t] for the tiles t of segment s: lane i takes t = begin+i, begin+i+256, ...
// in order, then the fixed block tree.  The per-segment totals go to
// out[k*out_stride + global_segment].
struct SegDesc {
  int64_t tile_begin;
  int64_t tile_count;
  int32_t global_index;
  int32_t pad;
};

// p[first] + p[first+STRIDE] + ... (< count), added in that order.  UN loads are put in flight together; slots past
// the end contribute +0.0, which leaves the running sum's bits alone, so the bits do not depend on UN.
template <int STRIDE, int UN>
__device__ __forceinline__ double strided_sum_batched(const double *__restrict__ p, int first, int64_t count)
{
  double v = 0.0;
  for (int64_t t = first; t < count; t += (int64_t)STRIDE * UN) {
    double x[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int64_t i = t + (int64_t)u * STRIDE;
      x[u] = i < count ? p[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UN; u++) v += x[u];
  }
  return v;
}
// Long runs take 40 loads per batch: a segment of the N=10M job (2441 tiles, 38 per lane of a wave) is one round
// trip to memory instead of five.  Short runs keep batches of 8 (fewer predicated-off loads).
template <int STRIDE> __device__ __forceinline__ double strided_sum(const double *__restrict__ p, int first, int64_t count)
{
  return count > (int64_t)STRIDE * 8 ? strided_sum_batched<STRIDE, 40>(p, first, count)
                                     : strided_sum_batched<STRIDE, 8>(p, first, count);
}

constexpr int kSegBlock = 1024;                   // threads of the segment kernel
constexpr int kSegWaves = kSegBlock / kWave;      // 16 jobs in flight per workgroup

// Job (s, k) = (local segment s, result vector k): one WAVE adds
// partials[k*pstride + t] over the tiles t of segment s -- lane i takes
// t = begin+i, begin+i+64, ... in order, then the fixed DPP wave tree -- and
// lane 63 stores out[k*out_stride + global_segment].  Jobs are dealt to the
// waves of the grid round-robin; which wave runs a job does not change its
// bits.  With a single workgroup (flag != nullptr) the kernel also publishes
// completion: every storing lane fences its store to system scope, the
// workgroup meets, thread 0 writes `seq` to *flag (pinned host memory the
// host thread is polling).
__global__ __launch_bounds__(kSegBlock) void segment_sum_kernel(const double *__restrict__ partials, int64_t pstride,
                                                                const SegDesc *__restrict__ segs, int nseg_local,
                                                                int nvec, double *__restrict__ out, int out_stride,
                                                                unsigned long long *flag, unsigned long long seq)
{
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = blockIdx.x * kSegWaves + (threadIdx.x / kWave);
  const int nwaves = gridDim.x * kSegWaves;
  const int jobs = nseg_local * nvec;
  for (int job = wave; job < jobs; job += nwaves) {
    const int s = job % nseg_local, k = job / nseg_local;
    const SegDesc sd = segs[s];
    const double *p = partials + (int64_t)k * pstride + sd.tile_begin;
    const double v0 = strided_sum<kWave>(p, lane, sd.tile_count);
    const double v = wave_sum_lane63(v0);
    if (lane == kWave - 1) {
      out[(int64_t)k * out_stride + sd.global_index] = v;
      if (flag) __threadfence_system();
    }
  }
  if (flag) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The same sum for segments with many tiles (N ~ 1e8): one 1024-thread
// workgroup per job -- thread i takes t = begin+i, begin+i+1024, ... in order,
// DPP wave tree, the 16 wave totals through LDS in wave order.  No completion
// word (several workgroups): the host waits on the stream.
__global__ __launch_bounds__(kSegBlock) void segment_sum_wide_kernel(const double *__restrict__ partials,
                                                                     int64_t pstride,
                                                                     const SegDesc *__restrict__ segs, int nseg_local,
                                                                     double *__restrict__ out, int out_stride)
{
  __shared__ double lds[kSegWaves];
  const int job = blockIdx.x;
  const int s = job % nseg_local, k = job / nseg_local;
  const SegDesc sd = segs[s];
  const double *p = partials + (int64_t)k * pstride + sd.tile_begin;
  double v = strided_sum<kSegBlock>(p, threadIdx.x, sd.tile_count);
  v = wave_sum_lane63(v);
  if ((threadIdx.x & (kWave - 1)) == kWave - 1) lds[threadIdx.x / kWave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < kSegWaves; w++) tot += lds[w];
    out[(int64_t)k * out_stride + sd.global_index] = tot;
  }
}

// After an all-reduce: copy the combined slots to pinned host memory and
// publish completion (one wave).
__global__ __launch_bounds__(kWave) void publish_slots_kernel(const double *__restrict__ src, double *__restrict__ dst,
                                                              int n, unsigned long long *flag, unsigned long long seq)
{
  for (int i = threadIdx.x; i < n; i += kWave) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

} // namespace mhlik
