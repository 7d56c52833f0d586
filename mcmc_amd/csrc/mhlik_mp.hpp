// mhlik_mp.hpp -- mixed-precision ("bright / dark" data blocks) likelihood.
//
// What the reference is after in src/mcmc_mp/mcmc_mp.cu:28-200,475-552
// (adapted from Liu et al., FPT'15; marked "in progress" in its README): a
// float pass over ALL rows gives sigma_S = 1/(1+exp(x.theta)) and from it the
// lower bound  B_i = sigma_S - eps  of the exact double value  L_i = sigma_D.
// Each data block carries an indicator z (bright = 1 / dark = 0):
//   dark  block: contributes  sum_i log(B_i)               (float data only)
//   bright block: contributes sum_i log(L_i - B_i)          (needs double data)
// and a bright (or a periodically re-examined dark) block resamples z with
//   P(bright) = 1 - exp( sum_i (log B_i - log L_i) ).
// The reference launches one CUDA block per data block, calls device-side
// cuBLAS from thread 0 for the double dots, and hands results out through
// atomic counters.  Here one workgroup streams a 1024-row tile of the float
// matrix with 16-byte loads, re-reads only the lanes of bright / re-examined
// blocks from the double matrix, reduces each data block with DPP row scans
// (a 64-row block is exactly one 16-lane DPP row at 4 rows per lane) and
// writes one record per data block, indexed by block -- no atomics, so the
// result does not depend on scheduling.
#pragma once

#include "mhlik_kernels.hpp"
#include "mhlik_synth.h"
#include "mhlik_detmath.h"

namespace mhlik {

constexpr int kMpTileRows = 1024; // 256 lanes x 4 rows; f32 tile (R=1) and two f64 tiles (R=1)

struct MpParams {
  int32_t block_rows;  // rows per data block: 64, 128, 256, 512 or 1024
  int32_t blocks;      // number of data blocks considered
  int32_t resamp_factor;
  int32_t resamp_choose;
  uint64_t u_seed, u_call;
  double eps;
  int64_t n_rows;   // rows that exist (global)
  int64_t row_base; // global row of this shard's first tile (a multiple of 1024)
};

// per-block uniforms: mhlik_mp_uniform in mhlik_synth.h (stands in for the
// reference's curandGenerateUniformDouble stream, mcmc_mp.cu:507)
__host__ __device__ inline double mp_uniform(uint64_t seed, uint64_t call, uint64_t block)
{
  return mhlik_mp_uniform(seed, call, block);
}

// inclusive scan over a 16-lane DPP row; lane 15 of each row holds the row sum
__device__ __forceinline__ double dpp_row_sum(double v)
{
  v += dpp_f64<0x111, 0xf>(v);
  v += dpp_f64<0x112, 0xf>(v);
  v += dpp_f64<0x114, 0xf>(v);
  v += dpp_f64<0x118, 0xf>(v);
  return v;
}

template <int CTRL> __device__ __forceinline__ float dpp_f32(float x)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_sum(float v)
{
  v += dpp_f32<0x111>(v);
  v += dpp_f32<0x112>(v);
  v += dpp_f32<0x114>(v);
  v += dpp_f32<0x118>(v);
  return v;
}

// Per data block record written by the tile kernel.
struct MpBlockOut {
  double value;    // bright at entry: sum log(L - B) (double); dark: sum log(B), accumulated in float
  int8_t was_bright;
  int8_t pad[7];
};

// grid = f32 tiles (1024 rows each); block = 256 threads.
// Xf: float matrix, tile t at Xf + t*gf.tile_stride, column stride gf.col_stride.
// Xd: double matrix in 512-row tiles (f32 tile t <-> f64 tiles 2t, 2t+1).
template <int U>
__global__ __launch_bounds__(kBlock) void mp_tile_kernel(const float *__restrict__ Xf, TileGeom gf,
                                                         const double *__restrict__ Xd, TileGeom gd,
                                                         ThetaArg<double> theta, MpParams p,
                                                         int8_t *__restrict__ z, MpBlockOut *__restrict__ out)
{
  typedef Vec16<float>::type V4;
  typedef Vec16<double>::type V2;
  // [quantity][16 DPP rows of the workgroup]
  __shared__ double lds_lb[16], lds_rs[16];
  __shared__ float lds_ld[16];

  const int d = gf.d;
  const int64_t tile = blockIdx.x;
  const int lane_row = threadIdx.x * 4;                // first of this lane's 4 rows in the tile
  const int64_t row0 = p.row_base + tile * kMpTileRows + lane_row;  // global row
  const int64_t covered = (int64_t)p.blocks * p.block_rows < p.n_rows ? (int64_t)p.blocks * p.block_rows : p.n_rows;
  const int64_t b = row0 / p.block_rows;               // this lane's data block
  const bool lane_live = row0 < covered;
  const bool bright = lane_live && z[b] != 0;
  const bool resample = lane_live && !bright && (b % p.resamp_factor == p.resamp_choose);
  const bool need_dp = bright || resample;

  // ---- float pass: every row ------------------------------------------------
  float accf[4] = {0.f, 0.f, 0.f, 0.f};
  {
    const float *tb = Xf + tile * gf.tile_stride + lane_row;
    int j0 = 0;
    for (; j0 + U <= d; j0 += U) {
      V4 x[U];
#pragma unroll
      for (int u = 0; u < U; u++) x[u] = load16<float, true>(tb + (int64_t)(j0 + u) * gf.col_stride);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const float t = (float)theta.v[j0 + u];
#pragma unroll
        for (int k = 0; k < 4; k++) accf[k] = fmaf(t, x[u][k], accf[k]);
      }
    }
    for (; j0 < d; j0++) {
      V4 x = load16<float, true>(tb + (int64_t)j0 * gf.col_stride);
      const float t = (float)theta.v[j0];
#pragma unroll
      for (int k = 0; k < 4; k++) accf[k] = fmaf(t, x[k], accf[k]);
    }
  }

  // ---- double pass: only lanes of bright / re-examined blocks ---------------
  double accd[4] = {0.0, 0.0, 0.0, 0.0};
  if (need_dp) {
    const int64_t dtile = tile * 2 + (lane_row >= 512 ? 1 : 0);
    const double *tb = Xd + dtile * gd.tile_stride + (lane_row & 511);
    constexpr int UD = 8; // columns in flight: 16 independent 16-byte loads per lane
    int j0 = 0;
    for (; j0 + UD <= d; j0 += UD) {
      V2 a[UD], c[UD];
#pragma unroll
      for (int u = 0; u < UD; u++) {
        const double *col = tb + (int64_t)(j0 + u) * gd.col_stride;
        a[u] = load16<double, true>(col);
        c[u] = load16<double, true>(col + 2);
      }
#pragma unroll
      for (int u = 0; u < UD; u++) {
        const double t = theta.v[j0 + u];
        accd[0] = madd(accd[0], t, a[u][0]);
        accd[1] = madd(accd[1], t, a[u][1]);
        accd[2] = madd(accd[2], t, c[u][0]);
        accd[3] = madd(accd[3], t, c[u][1]);
      }
    }
    for (; j0 < d; j0++) {
      const double *col = tb + (int64_t)j0 * gd.col_stride;
      V2 a = load16<double, true>(col), c = load16<double, true>(col + 2);
      const double t = theta.v[j0];
      accd[0] = madd(accd[0], t, a[0]);
      accd[1] = madd(accd[1], t, a[1]);
      accd[2] = madd(accd[2], t, c[0]);
      accd[3] = madd(accd[3], t, c[1]);
    }
  }

  // ---- per-row terms, then this lane's 4-row partial -------------------------
  float s_ld = 0.f;   // sum log(B_i), float            (dark value)
  double s_lb = 0.0;  // sum log(L_i - B_i)              (bright value)
  double s_rs = 0.0;  // sum (log B_i - log L_i)         (resampling exponent)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (row0 + k < covered) {
      const float sig_s = mhlik_sigmoidf_det(accf[k]);
      const double Bd = (double)sig_s - p.eps; // mcmc_mp.cu:63,146: float sigmoid minus the double bound
      if (need_dp) {
        const double L = 1.0 / (1.0 + exp(accd[k]));
        if (bright) {
          const float Bf = (float)Bd; // the reference keeps B in a float array (sLDblock)
          s_rs += (double)logf(Bf) - log(L);
          s_lb += log(L - (double)Bf);
        } else {
          s_ld += (float)log(Bd);
          s_rs += log(Bd) - log(L);
        }
      } else {
        s_ld += (float)log(Bd);
      }
    }
  }

  // ---- 16-lane rows (64 table rows) -> LDS -> data blocks --------------------
  s_ld = dpp_row_sum(s_ld);
  s_lb = dpp_row_sum(s_lb);
  s_rs = dpp_row_sum(s_rs);
  const int dpp_row = threadIdx.x >> 4;
  if ((threadIdx.x & 15) == 15) {
    lds_ld[dpp_row] = s_ld;
    lds_lb[dpp_row] = s_lb;
    lds_rs[dpp_row] = s_rs;
  }
  __syncthreads();

  const int rows_per_block = p.block_rows / 64;    // DPP rows per data block: 1..16
  const int blocks_per_tile = 16 / rows_per_block; // data blocks in this tile
  if ((int)threadIdx.x < blocks_per_tile) {
    const int64_t blk = p.row_base / p.block_rows + tile * blocks_per_tile + threadIdx.x; // global block
    if (blk < p.blocks && blk * p.block_rows < p.n_rows) {
      float ld = 0.f;
      double lb = 0.0, rs = 0.0;
      for (int g = 0; g < rows_per_block; g++) {
        const int i = threadIdx.x * rows_per_block + g;
        ld += lds_ld[i];
        lb += lds_lb[i];
        rs += lds_rs[i];
      }
      const bool was_bright = z[blk] != 0;
      const bool re = !was_bright && (blk % p.resamp_factor == p.resamp_choose);
      MpBlockOut o;
      o.value = was_bright ? lb : (double)ld;
      o.was_bright = was_bright ? 1 : 0;
      for (int q = 0; q < 7; q++) o.pad[q] = 0;
      out[blk] = o;
      if (was_bright || re) {
        const double u = mp_uniform(p.u_seed, p.u_call, (uint64_t)blk);
        const double p_bright = 1.0 - exp(rs);
        z[blk] = (p_bright <= u) ? 0 : 1; // mcmc_mp.cu:185-193
      }
    }
  }
}

// Add the block records in a fixed order, in two stages (a single workgroup
// walking 1e5+ records is latency-bound): workgroup g of kMpReduceGroups sums
// its contiguous run of records -- thread i takes records i, i+256, ... of the
// run, doubles for blocks bright at entry, a float accumulator for blocks dark
// at entry, eight loads in flight -- then the fixed block tree; the final
// single workgroup adds the kMpReduceGroups partials.
//   res[0] = red_d (double sum over blocks bright at entry)
//   res[1] = red_f (float sum over blocks dark at entry)
//   res[2] = bright count after resampling, res[3] = dark count
constexpr int kMpReduceGroups = 256; // groups over ALL segments: a segment gets kMpReduceGroups / segments of them

// grid = G groups of one row segment, whose blocks are [lo_blk, hi_blk): group j sums the records of its contiguous
// run.  part[j], part[G + j], part[2G + j] = double sum / float sum / bright count.  The shape depends on the
// segment's block range and on G alone, never on which GPU holds the segment.
__global__ __launch_bounds__(kBlock) void mp_reduce_kernel(const MpBlockOut *__restrict__ out,
                                                           const int8_t *__restrict__ z, int32_t lo_blk, int32_t hi_blk,
                                                           double *__restrict__ part /* [3][G] */)
{
  __shared__ double lds[kBlock / kWave];
  const int G = gridDim.x;
  const int blocks = hi_blk - lo_blk;
  const int per = (blocks + G - 1) / G;
  const int lo = lo_blk + blockIdx.x * per;
  const int hi = lo + per < hi_blk ? lo + per : hi_blk;
  double sd = 0.0, nb = 0.0;
  float sf = 0.f;
  for (int b0 = lo + (int)threadIdx.x; b0 < hi; b0 += kBlock * 8) {
    MpBlockOut o[8];
    int8_t zz[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int b = b0 + u * kBlock;
      if (b < hi) {
        o[u] = out[b];
        zz[u] = z[b];
      } else {
        o[u].value = 0.0;
        o[u].was_bright = 0;
        zz[u] = 0;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (o[u].was_bright)
        sd += o[u].value;
      else
        sf += (float)o[u].value;
      nb += (zz[u] != 0) ? 1.0 : 0.0;
    }
  }
  double t = block_sum(sd, lds);
  __syncthreads();
  double tf = block_sum((double)sf, lds); // per-thread float sums added in double
  __syncthreads();
  double tb = block_sum(nb, lds);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = t;
    part[G + blockIdx.x] = tf;
    part[2 * G + blockIdx.x] = tb;
  }
}

// grid = local segments.  Workgroup s adds the G group partials of local segment s (part + s*3*G) and stores the
// segment's three sums into the slot vector: slots[global], slots[nseg + global], slots[2*nseg + global].
__global__ __launch_bounds__(kBlock) void mp_reduce_final_kernel(const double *__restrict__ part, int G,
                                                                 const int32_t *__restrict__ seg_global, int nseg,
                                                                 double *__restrict__ slots)
{
  __shared__ double lds[kBlock / kWave];
  const double *p = part + (int64_t)blockIdx.x * 3 * G;
  const bool has = (int)threadIdx.x < G;
  double t = block_sum(has ? p[threadIdx.x] : 0.0, lds);
  __syncthreads();
  double tf = block_sum(has ? p[G + threadIdx.x] : 0.0, lds);
  __syncthreads();
  double tb = block_sum(has ? p[2 * G + threadIdx.x] : 0.0, lds);
  if (threadIdx.x == 0) {
    const int g = seg_global[blockIdx.x];
    slots[g] = t;
    slots[nseg + g] = tf;
    slots[2 * nseg + g] = tb;
  }
}

// likelihood_bound (mcmc_mp.cu:746-821): per tile, sum of double sigmoids and
// the minimum of eps_i = sigma_D - sigma_S.
__global__ __launch_bounds__(kBlock) void mp_bound_kernel(const float *__restrict__ Xf, TileGeom gf,
                                                          const double *__restrict__ Xd, TileGeom gd,
                                                          ThetaArg<double> theta, int64_t n_rows, int64_t row_base,
                                                          double *__restrict__ tile_sum,
                                                          double *__restrict__ tile_min)
{
  typedef Vec16<float>::type V4;
  typedef Vec16<double>::type V2;
  __shared__ double lds[kBlock / kWave];
  __shared__ double lds_min[kBlock / kWave];
  const int d = gf.d;
  const int64_t tile = blockIdx.x;
  const int lane_row = threadIdx.x * 4;
  const int64_t row0 = row_base + tile * kMpTileRows + lane_row; // global row
  float accf[4] = {0.f, 0.f, 0.f, 0.f};
  double accd[4] = {0.0, 0.0, 0.0, 0.0};
  const float *tf = Xf + tile * gf.tile_stride + lane_row;
  const int64_t dtile = tile * 2 + (lane_row >= 512 ? 1 : 0);
  const double *td = Xd + dtile * gd.tile_stride + (lane_row & 511);
  for (int j = 0; j < d; j++) {
    V4 x = load16<float, true>(tf + (int64_t)j * gf.col_stride);
    const double *col = td + (int64_t)j * gd.col_stride;
    V2 a = load16<double, true>(col), c = load16<double, true>(col + 2);
    const double t = theta.v[j];
    const float tfl = (float)t;
#pragma unroll
    for (int k = 0; k < 4; k++) accf[k] = fmaf(tfl, x[k], accf[k]);
    accd[0] = madd(accd[0], t, a[0]);
    accd[1] = madd(accd[1], t, a[1]);
    accd[2] = madd(accd[2], t, c[0]);
    accd[3] = madd(accd[3], t, c[1]);
  }
  double s = 0.0, mn = __builtin_inf();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (row0 + k < n_rows) {
      const double L = 1.0 / (1.0 + exp(accd[k]));
      const double S = (double)mhlik_sigmoidf_det(accf[k]);
      s += L;
      mn = fmin(mn, L - S);
    }
  }
  s = block_sum(s, lds);
  // min over the workgroup: shuffles are fine here, this is not a hot kernel
  for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_xor(mn, off));
  if ((threadIdx.x & 63) == 0) lds_min[threadIdx.x >> 6] = mn;
  __syncthreads();
  if (threadIdx.x == 0) {
    tile_sum[tile] = s;
    tile_min[tile] = fmin(fmin(lds_min[0], lds_min[1]), fmin(lds_min[2], lds_min[3]));
  }
}

// grid = local segments: total of the segment's tile sums and minimum of its tile minima, stored at
// slots[global] and slots[nseg + global] (a segment without rows: 0 and +inf).
__global__ __launch_bounds__(kBlock) void mp_bound_final_kernel(const double *__restrict__ tile_sum,
                                                                const double *__restrict__ tile_min,
                                                                const SegDesc *__restrict__ segs, int nseg,
                                                                double *__restrict__ slots)
{
  __shared__ double lds[kBlock / kWave];
  __shared__ double lds_min[kBlock / kWave];
  const SegDesc sd = segs[blockIdx.x];
  double s = 0.0, mn = __builtin_inf();
  for (int64_t t = threadIdx.x; t < sd.tile_count; t += kBlock) {
    s += tile_sum[sd.tile_begin + t];
    mn = fmin(mn, tile_min[sd.tile_begin + t]);
  }
  s = block_sum(s, lds);
  for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_xor(mn, off));
  if ((threadIdx.x & 63) == 0) lds_min[threadIdx.x >> 6] = mn;
  __syncthreads();
  if (threadIdx.x == 0) {
    slots[sd.global_index] = s;
    slots[nseg + sd.global_index] = fmin(fmin(lds_min[0], lds_min[1]), fmin(lds_min[2], lds_min[3]));
  }
}

} // namespace mhlik
