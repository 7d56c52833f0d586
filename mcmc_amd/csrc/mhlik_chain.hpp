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

// What follows the segment sums in one iteration, shared by the step kernel (state in global memory) and the
// persistent kernel (every workgroup keeps its own copy of the state in LDS and runs this redundantly: the same
// instructions on the same numbers, hence the same bits everywhere).
//   slots     LDS, the nseg per-segment sums (written before the call; the first barrier in here orders them)
//   terms     LDS scratch, kPriorChunk doubles;  s_accept  LDS word
//   h, cur, prop   the chain state this workgroup reads and updates
//   write_out  whether this workgroup files the iteration's outputs (lik, ratio, acc, sample row)
// mode 0: MH iteration `it` (0 <= it < c.n) of the batch
// mode 1: chain start -- the pass was over prop = theta0: cur = prop,
//         cposterior = prior + L; nothing is written to the batch outputs
// The prior of the proposal, sum_j log N(prop_j; 0, sigma): the terms in parallel, then added in column order like the
// host loop does (float chain: a float accumulator += a double term).  Every thread returns the same value.
template <typename T>
__device__ __forceinline__ T chain_prior(const ChainDev &c, const T *prop, double *terms)
{
  const int d = c.d;
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
  return prior;
}

// The rest of the iteration once the slots are in LDS (the caller has put a barrier between their stores and this).
// pref: the caller has loaded this thread's share of the iteration's inputs ahead of time (the persistent kernel does,
// at the top of the iteration, so that no memory round trip sits between the gather and the next proposal):
// logu_pref = logu[it] (thread 0), inc_pref = inc[it + 1][threadIdx.x] (threads < d; needs d <= blockDim.x).
template <typename T>
__device__ __forceinline__ void chain_accept(const ChainDev &c, int it, int mode, T prior, const double *slots, int nseg,
                                             int *s_accept, ChainHdr *h, T *cur, T *prop, bool write_out,
                                             bool pref = false, T logu_pref = (T)0, T inc_pref = (T)0)
{
  const int d = c.d;
  if (threadIdx.x == 0) {
    double total = 0.0;
    for (int g = 0; g < nseg; g++) total += slots[g];
    const T L = (T)total;
    const T ppost = prior + L;
    if (mode == 1) {
      h->cpost = (double)ppost;
      h->accepted = 0;
      *s_accept = 1;
    } else {
      const T ratio = ppost - (T)h->cpost;
      const T logu = pref ? logu_pref : ((const T *)c.logu)[it];
      const int a = logu <= ratio ? 1 : 0;
      if (a) {
        h->cpost = (double)ppost;
        h->accepted += 1;
      }
      if (write_out) {
        c.lik[it] = (double)L;
        c.ratio[it] = (double)ratio;
        c.acc[it] = (unsigned char)a;
      }
      *s_accept = a;
    }
  }
  __syncthreads();
  const bool accept = *s_accept != 0;
  const T *inc = (const T *)c.inc;
  T *samples = (T *)c.samples;
  for (int k = threadIdx.x; k < d; k += blockDim.x) {
    const T x = accept ? prop[k] : cur[k];
    if (accept) cur[k] = x;
    if (mode == 0) {
      if (samples && write_out) samples[(int64_t)it * d + k] = x;
      if (it + 1 < c.n) prop[k] = x + (pref ? inc_pref : inc[(int64_t)(it + 1) * d + k]);
    }
  }
}

template <typename T>
__device__ __forceinline__ void chain_decide(const ChainDev &c, int it, int mode, const double *slots, int nseg,
                                             double *terms, int *s_accept, ChainHdr *h, T *cur, T *prop, bool write_out)
{
  const T prior = chain_prior<T>(c, prop, terms); // its barriers also order the slots' stores before chain_accept
  chain_accept<T>(c, it, mode, prior, slots, nseg, s_accept, h, cur, prop, write_out);
}

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
  chain_decide<T>(c, it, mode, slots, nseg, terms, &s_accept, c.hdr, cur, prop, true);
  // the batch's last iteration leaves a copy of the state at the head of the download image
  if (mode == 0 && it + 1 == c.n && c.cur_out != nullptr) {
    __syncthreads();
    for (int k = threadIdx.x; k < c.d; k += blockDim.x) ((T *)c.cur_out)[k] = cur[k];
    if (threadIdx.x == 0) *c.hdr_out = *c.hdr;
  }
}

// ------------------------------------------------------------------------------------------------------------
// The whole batch in ONE launch, for matrices small enough that two dependent dispatches per iteration are what an
// iteration costs (BASELINE configs[0] and [1]: the reference runs them 3000 x 40000 iterations,
// scr/sh_scr/multiple_runs_gpu.sh:3-6).  Work unit = job (segment s, lane i): the tiles i, i+64, ... of segment s,
// whose partials lane i of segment_sum_kernel's wave adds in that order.  A workgroup computes its jobs' sums with
// the tile kernel's code, publishes each as two 8-byte {epoch, half} granules (write-through stores), and EVERY
// workgroup gathers all jobs' granules, finishes the segment sums with the same DPP tree and runs chain_decide on its
// own copy of the state: no second barrier, no broadcast, and the bits of the two-kernel path.  Granules alternate
// between two buffers by iteration parity: a workgroup can only publish iteration e+2 after every workgroup has
// published e+1, i.e. has finished reading e.
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

struct PersistJob {
  int64_t first_tile; // local tile index of the job's first tile (then +64 per further tile)
  int32_t ntiles;
  int32_t pub_idx;    // segment * 64 + lane: where the job's sum sits in the segment-sum wave
};

struct PersistArgs {
  const void *X;
  TileGeom g;
  const int32_t *tile_valid;
  const PersistJob *jobs;
  int njobs, nseg;
  int lds_tiles;           // tiles a workgroup keeps in LDS behind the ones it keeps in registers
  unsigned long long *pub; // [2][2 * njobs] granules, zero before the launch
  unsigned int *abort_word; // zero before the launch; non-zero: some workgroup gave up waiting
  unsigned long long timeout_ticks; // of the 100 MHz wall clock
};

constexpr int kPersistMaxSeg = 16;
constexpr int kPersistMaxJobs = kPersistMaxSeg * kWave;
constexpr int kPersistMaxMyJobs = 8;   // jobs one workgroup may hold
constexpr int kPersistMaxMyTiles = 64; // tiles one workgroup may hold

// A workgroup's tiles live where they fit, in this order: the first RT in REGISTERS (DP columns each, the columns
// beyond d zero: adding theta_j * 0 = +0 to a sum that is never -0 leaves its bits alone), the next `lds_tiles` in
// LDS, the rest are re-read from global memory (L2 / Infinity Cache) every iteration.  N = 1M, d = 16 (128 MB) on 256
// workgroups: 7-8 tiles of 64 KB each = 5 + 2 + at most 1.  RT = 0: nothing in registers, any d.
template <typename T, int DP, int RT>
__global__ __launch_bounds__(kBlock, 1) void chain_persist_kernel(ChainDev c, PersistArgs a)
{
  typedef typename Vec16<T>::type V;
  constexpr int VEC = Vec16<T>::N;
  constexpr int TILE_ROWS = kBlock * VEC;
  constexpr int RTA = RT > 0 ? RT : 1, DPA = DP > 0 ? DP : 1;
  __shared__ double vals[kPersistMaxJobs]; // [segment][lane] lane sums, 0.0 where a lane has no tiles
  __shared__ int pubidx[kPersistMaxJobs];
  __shared__ double slots[kMaxSegmentsDev];
  __shared__ double terms[kPriorChunk];
  __shared__ double wsum[kPersistMaxMyTiles * (kBlock / kWave)]; // wave sums of this workgroup's tiles
  __shared__ int ftile[kPersistMaxMyTiles], fvalid[kPersistMaxMyTiles]; // its tiles, job after job, in summation order
  __shared__ int jfirst[kPersistMaxMyJobs + 1], jq[kPersistMaxMyJobs];  // its jobs: range in ftile, global job index
  __shared__ int s_accept, s_failed, s_nm, s_nj;
  __shared__ ChainHdr hdr;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[]; // cur, prop, then the LDS-resident tiles
  const int d = c.d;
  const int dpad = ((d > DP ? d : DP) + 3) & ~3;
  T *cur = (T *)smem, *prop = cur + dpad;
  T *ltiles = prop + dpad; // 16-byte aligned: dpad is a multiple of 4 elements

  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int njobs = a.njobs;
  const bool out = blockIdx.x == 0;
  const T *X = (const T *)a.X;
  for (int k = tid; k < kPersistMaxJobs; k += kBlock) vals[k] = 0.0;
  for (int q = tid; q < njobs; q += kBlock) pubidx[q] = a.jobs[q].pub_idx;
  for (int k = tid; k < dpad; k += kBlock) {
    cur[k] = k < d ? ((const T *)c.cur)[k] : (T)0;
    prop[k] = k < d ? ((const T *)c.prop)[k] : (T)0;
  }
  if (tid == 0) {
    hdr = *c.hdr;
    s_failed = 0;
    int nm = 0, nj = 0;
    for (int q = blockIdx.x; q < njobs && nj < kPersistMaxMyJobs; q += gridDim.x) {
      const PersistJob job = a.jobs[q];
      jq[nj] = q;
      jfirst[nj++] = nm;
      for (int k = 0; k < job.ntiles && nm < kPersistMaxMyTiles; k++) {
        const int tile = (int)job.first_tile + k * kWave;
        ftile[nm] = tile;
        fvalid[nm++] = a.tile_valid[tile];
      }
    }
    jfirst[nj] = nm;
    s_nm = nm;
    s_nj = nj;
  }
  __syncthreads();
  const int nm = s_nm, nj = s_nj;
  const uint32_t lane_row = (uint32_t)tid * VEC;
  const int nlds_end = nm < RT + a.lds_tiles ? nm : RT + a.lds_tiles;

  // ---- once: this workgroup's tiles into registers and LDS
  V xr[RTA][DPA];
#pragma unroll
  for (int r = 0; r < RT; r++) {
    const T *tb = X + (int64_t)(r < nm ? ftile[r] : 0) * a.g.tile_stride + lane_row;
#pragma unroll
    for (int j = 0; j < DP; j++) {
      V z;
#pragma unroll
      for (int k = 0; k < VEC; k++) z[k] = (T)0;
      xr[r][j] = (r < nm && j < d) ? *reinterpret_cast<const V *>(tb + (int64_t)j * a.g.col_stride) : z;
    }
  }
  for (int k = RT; k < nlds_end; k++) {
    const T *tb = X + (int64_t)ftile[k] * a.g.tile_stride + lane_row;
    T *lt = ltiles + (size_t)(k - RT) * d * TILE_ROWS + lane_row;
    for (int j = 0; j < d; j++)
      *reinterpret_cast<V *>(lt + (size_t)j * TILE_ROWS) = *reinterpret_cast<const V *>(tb + (int64_t)j * a.g.col_stride);
  }
  __syncthreads();
  ThetaArg<T> no_karg; // never read: theta is the proposal in LDS
  uint32_t *vals32 = (uint32_t *)vals;

  for (int it = 0; it < c.n; it++) {
    const unsigned epoch = (unsigned)it + 1u;
    gu64 *pb = (gu64 *)a.pub + (size_t)(it & 1) * 2 * (size_t)njobs;
    // this iteration's log u and the next proposal's increments: requested now, used after the gather
    const bool pref = d <= kBlock;
    T logu_pref = (T)0, inc_pref = (T)0;
    if (pref) {
      if (tid == 0) logu_pref = ((const T *)c.logu)[it];
      if (tid < d && it + 1 < c.n) inc_pref = ((const T *)c.inc)[(int64_t)(it + 1) * d + tid];
    }
    // ---- every tile's four wave sums
    if constexpr (RT > 0) {
      T th[DPA];
#pragma unroll
      for (int j = 0; j < DP; j++) th[j] = prop[j];
      // two tiles at a time: their 2 * VEC row chains are independent, and -log(1+exp(.)) of every row is computed
      // unconditionally (rows past `valid` then add +0.0, which leaves the sum's bits alone: it is never -0.0) so that
      // the scheduler can interleave the chains -- one wave per SIMD has nothing else to hide their latency with
#pragma unroll
      for (int r0 = 0; r0 < RT; r0 += 2) {
        if (r0 < nm) {
          constexpr int G = 2;
          T acc[G][VEC];
#pragma unroll
          for (int g = 0; g < G; g++)
#pragma unroll
            for (int k = 0; k < VEC; k++) acc[g][k] = (T)0;
#pragma unroll
          for (int j = 0; j < DP; j++)
#pragma unroll
            for (int g = 0; g < G; g++)
              if (r0 + g < RT)
#pragma unroll
                for (int k = 0; k < VEC; k++) acc[g][k] = madd(acc[g][k], th[j], xr[r0 + g < RT ? r0 + g : 0][j][k]);
          const bool two = r0 + 1 < RT && r0 + 1 < nm;
          double sl[G] = {0.0, 0.0};
          if (two) {
            double v[G][VEC];
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
              for (int k = 0; k < VEC; k++) v[g][k] = (double)neg_log1pexp(acc[g][k]);
#pragma unroll
            for (int g = 0; g < G; g++) {
              const int valid = fvalid[r0 + g < RT ? r0 + g : 0];
#pragma unroll
              for (int k = 0; k < VEC; k++) sl[g] += ((int)lane_row + k < valid) ? v[g][k] : 0.0;
            }
          } else {
            double v[VEC];
#pragma unroll
            for (int k = 0; k < VEC; k++) v[k] = (double)neg_log1pexp(acc[0][k]);
            const int valid = fvalid[r0];
#pragma unroll
            for (int k = 0; k < VEC; k++) sl[0] += ((int)lane_row + k < valid) ? v[k] : 0.0;
          }
          sl[0] = wave_sum_lane63(sl[0]);
          if (lane == kWave - 1) wsum[r0 * (kBlock / kWave) + wave] = sl[0];
          if (two) {
            sl[1] = wave_sum_lane63(sl[1]);
            if (lane == kWave - 1) wsum[(r0 + 1) * (kBlock / kWave) + wave] = sl[1];
          }
        }
      }
    }
    for (int k = RT; k < nlds_end; k++) { // LDS-resident
      double sl = tile_lane_sum<T, 1, 16, false, false, false>(ltiles + (size_t)(k - RT) * d * TILE_ROWS, TILE_ROWS, d,
                                                               fvalid[k], no_karg, prop, nullptr);
      sl = wave_sum_lane63(sl);
      if (lane == kWave - 1) wsum[k * (kBlock / kWave) + wave] = sl;
    }
    for (int k = nlds_end > RT ? nlds_end : RT; k < nm; k++) { // re-read from memory
      double sl = tile_lane_sum<T, 1, 16, false, false, false>(X + (int64_t)ftile[k] * a.g.tile_stride, a.g.col_stride, d,
                                                               fvalid[k], no_karg, prop, nullptr);
      sl = wave_sum_lane63(sl);
      if (lane == kWave - 1) wsum[k * (kBlock / kWave) + wave] = sl;
    }
    __syncthreads();
    // ---- a job's sum: its tiles' partials in order, each partial the four wave sums in wave order (block_sum's shape)
    if (tid < nj) {
      double v = 0.0;
      for (int k = jfirst[tid]; k < jfirst[tid + 1]; k++) {
        double sp = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; w++) sp += wsum[k * (kBlock / kWave) + w];
        v += sp;
      }
      const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
      const int q = jq[tid];
      __hip_atomic_store(pb + 2 * q, ((unsigned long long)epoch << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(pb + 2 * q + 1, ((unsigned long long)epoch << 32) | (bits >> 32), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the prior needs only the proposal: it runs while the other workgroups' granules are on their way
    const T prior = chain_prior<T>(c, prop, terms);
    // ---- gather every job's granules (each thread polls its own; the data is the flag)
    // A thread owns granules tid, tid + 256, ...: all of them are requested together, and only those whose tag is not
    // this iteration's yet are requested again (one round trip per pass, not one per granule).
    bool failed = false;
    {
      constexpr int M = 2 * kPersistMaxJobs / kBlock; // 8
      unsigned long long x[M];
      unsigned pending = 0;
#pragma unroll
      for (int m = 0; m < M; m++)
        if (tid + m * kBlock < 2 * njobs) pending |= 1u << m;
      unsigned long long t0 = 0;
      for (unsigned spins = 0; pending; spins++) {
#pragma unroll
        for (int m = 0; m < M; m++)
          if (pending & (1u << m)) x[m] = __hip_atomic_load(pb + tid + m * kBlock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int m = 0; m < M; m++)
          if ((pending & (1u << m)) && (unsigned)(x[m] >> 32) == epoch) {
            const int g = tid + m * kBlock;
            vals32[2 * pubidx[g >> 1] + (g & 1)] = (uint32_t)x[m];
            pending &= ~(1u << m);
          }
        if (!pending) break;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 255u) { // every wave reaches an exit: a bounded wait, and a word that tells the others
          const unsigned long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          if (now - t0 > a.timeout_ticks ||
              __hip_atomic_load((gu32 *)a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            failed = true;
            break;
          }
        }
      }
    }
    if (failed) {
      __hip_atomic_store((gu32 *)a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_failed = 1;
    }
    __syncthreads();
    if (s_failed) break;
    // ---- the segment sums' DPP tree over the 64 lane sums of each segment, then the decision
    for (int sg = wave; sg < a.nseg; sg += kBlock / kWave) {
      const double v = wave_sum_lane63(vals[sg * kWave + lane]);
      if (lane == kWave - 1) slots[sg] = v;
    }
    __syncthreads();
    chain_accept<T>(c, it, 0, prior, slots, a.nseg, &s_accept, &hdr, cur, prop, out, pref, logu_pref, inc_pref);
    __syncthreads();
  }
  if (out) { // the state after the batch: to the chain's buffers and to the head of the download image
    for (int k = tid; k < d; k += kBlock) {
      ((T *)c.cur)[k] = cur[k];
      if (c.cur_out) ((T *)c.cur_out)[k] = cur[k];
    }
    if (tid == 0) {
      hdr.pad[0] = s_failed | (int)__hip_atomic_load((gu32 *)a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *c.hdr = hdr;
      if (c.hdr_out) *c.hdr_out = hdr;
    }
  }
}

} // namespace mhlik
