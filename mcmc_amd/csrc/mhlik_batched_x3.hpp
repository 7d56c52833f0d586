// mhlik_batched_x3.hpp -- the K-chain float pass with every float split into
// three bfloat16 pieces, X.Theta on v_mfma_f32_16x16x32_bf16.
//
// Why: on gfx950 the bf16 matrix rate is 16x the f32 rate (32x32x16 bf16 in 32
// cycles against 32x32x2 f32 in 64), so the f32-MFMA pass of mhlik_batched.hpp
// is bound by the matrix pipe at K = 64 (1.5 ms at N = 10M, d = 128) although
// the bytes it reads would only take 0.72 ms.  A float has 24 significant bits
// and a bfloat16 has 8, so
//     x = a + b + c,   a = top 16 bits of x,  b = top 16 bits of (x - a),  c = top 16 bits of (x - a - b)
// holds EXACTLY (every subtraction is exact; truncation, no rounding), likewise
// theta = p + q + r, and
//     x*theta = a*p + (a*q + b*p) + (a*r + b*q + c*p) + [b*r + c*q + c*r]
// TERMS = 6 keeps the first six products (each exact in the f32 the MFMA
// accumulates in) and drops the bracket, which is below 2^-23 of |x*theta|;
// TERMS = 9 keeps all of them.  Six bf16 MFMAs per 16 columns cost 12 cycles per
// column against 32 for the f32 MFMA, which puts the pass back under the HBM
// roof.  The result is NOT bit-identical to the fmaf chain of the single-theta
// float kernel (the f32-MFMA pass is, and stays available); against a double
// dot it is as accurate (tests/test_gpu_batched.py).
//
// Layout: the tile is column-major, a lane loads 16 bytes = 4 adjacent rows of
// one column (exactly as the f32 pass does); the four 16-lane groups of a wave
// take columns 0-7, 8-15, 16-23, 24-31 of a 32-column k-block, so after 8 loads
// a lane holds a 4-row x 8-column patch: component m of the 8 loads IS the
// 8-element A fragment (row l&15, k = 8*(l>>4) + j) of row set m, with no lane
// movement.  The 16x16 shape keeps the accumulators at 64 registers per wave
// (64 rows x 64 chains), which leaves room for two load buffers at two waves
// per SIMD; the 32x32 shape needs 128 and spills.  Theta pieces sit in LDS in
// B-fragment order, one ds_read_b128 per fragment (mhlik_x3_stage_theta).
#pragma once

#include "mhlik_batched.hpp"

namespace mhlik {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kX3KBlock = 32; // columns per v_mfma_f32_16x16x32_bf16
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (hi16(v1) << 16) | hi16(v0): two truncated bfloat16 in one dword, element order (v0, v1)
__device__ __forceinline__ unsigned x3_pack_hi(float v0, float v1)
{
  return __builtin_amdgcn_perm(__float_as_uint(v1), __float_as_uint(v0), 0x07060302u);
}
__device__ __forceinline__ float x3_residual(float v) // v - (top 16 bits of v), exact
{
  return v - __uint_as_float(__float_as_uint(v) & 0xffff0000u);
}

// Split the 8 values of one row set (component m of the 8 loaded vectors) into three fragments.
template <typename V4>
__device__ __forceinline__ void x3_split(const V4 (&x)[8], int m, bf16x8 &fa, bf16x8 &fb, bf16x8 &fc)
{
  u32x4 ua, ub, uc;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float v0 = x[2 * i][m], v1 = x[2 * i + 1][m];
    ua[i] = x3_pack_hi(v0, v1);
    const float s0 = x3_residual(v0), s1 = x3_residual(v1);
    ub[i] = x3_pack_hi(s0, s1);
    uc[i] = x3_pack_hi(x3_residual(s0), x3_residual(s1));
  }
  fa = __builtin_bit_cast(bf16x8, ua);
  fb = __builtin_bit_cast(bf16x8, ub);
  fc = __builtin_bit_cast(bf16x8, uc);
}

// Lab variant (EXP & 512): the same for all four row sets of a block at once, written on whole loaded vectors
// (4 adjacent rows of one column): the two exact subtractions of a vector are two v_pk_add_f32 each instead of
// four v_sub_f32, which takes a sixth off the split's instructions.  Same pieces, same bits as x3_split -- and
// 4 % slower in the full kernel (tools/x3_ablate, profiles/r02_x3_ablate.txt), so the kernel splits row set by
// row set.
template <typename V4>
__device__ __forceinline__ void x3_split_all(const V4 (&x)[8], bf16x8 (&fa)[4], bf16x8 (&fb)[4], bf16x8 (&fc)[4])
{
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto residual2 = [](f32x2 v) -> f32x2 {
    const u32x2 top = __builtin_bit_cast(u32x2, v) & 0xffff0000u;
    return v - __builtin_bit_cast(f32x2, top);
  };
  u32x4 ua[4], ub[4], uc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const V4 v0 = x[2 * i], v1 = x[2 * i + 1];
#pragma unroll
    for (int h = 0; h < 2; h++) { // rows 2h, 2h+1 of both columns
      const f32x2 p0 = {v0[2 * h], v0[2 * h + 1]}, p1 = {v1[2 * h], v1[2 * h + 1]};
      const f32x2 s0 = residual2(p0), s1 = residual2(p1);
      const f32x2 t0 = residual2(s0), t1 = residual2(s1);
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int m = 2 * h + e;
        ua[m][i] = x3_pack_hi(p0[e], p1[e]);
        ub[m][i] = x3_pack_hi(s0[e], s1[e]);
        uc[m][i] = x3_pack_hi(t0[e], t1[e]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 4; m++) {
    fa[m] = __builtin_bit_cast(bf16x8, ua[m]);
    fb[m] = __builtin_bit_cast(bf16x8, ub[m]);
    fc[m] = __builtin_bit_cast(bf16x8, uc[m]);
  }
}

// A wave's share of one 1024-row tile: four passes of 64 rows (pass p of wave w: tile rows 256p + 64w ..,
// so the four waves of a workgroup read 1 KB of every column side by side), each over d_pad/32 k-blocks.
// v_mfma_f32_16x16x32_bf16: A[row l&15][k = 8*(l>>4) + j], B[k = 8*(l>>4) + j][col l&15], C: col = l&15,
// row = 4*(l>>4) + reg.  A lane's 8 loads of a k-block are columns 8*(l>>4) + 0..7, rows 4*(l&15) .. +3;
// acc[m][n] = rows {row0 + 4*i + m : i < 16} x chains 16n .. 16n+15.  epi(pass, row0, acc) consumes a pass.
//
// The (pass, k-block) pairs form ONE software pipeline over two register buffers: a buffer's next block is
// requested as soon as it has been split into fragments, so two blocks are in flight under the MFMAs of a
// third, across pass boundaries too.  Every load in the steady loop is unconditional: a conditional load makes
// the compiler's wait-count bookkeeping assume the shorter queue and wait for everything in flight.
// The pipeline also runs on across the tiles a workgroup owns (tile0, tile0 + tile_step, ... < ntiles), so
// Theta is staged into LDS once per workgroup and the load queue never drains.  tile_done(tile) runs after the
// last pass of a tile (all four waves reach it the same number of times: it may use barriers).
template <int NB, int TERMS, int EXP, int NBUF = 2, class Epi, class Done>
__device__ __forceinline__ void batched_x3_stream(const float *__restrict__ X, int64_t tile_stride, int64_t tile0,
                                                  int64_t tile_step, int64_t ntiles, int wave, int64_t col_stride,
                                                  int d, int d_pad, const u32x4 *thetaL, int lane, Epi epi,
                                                  Done tile_done)
{
  typedef Vec16<float>::type V4;
  const int arow = lane & 15, g = lane >> 4;
  const unsigned loff = (unsigned)((4 * arow + 8 * g * col_stride) * sizeof(float));
  const int nkb = d_pad / kX3KBlock;   // k-blocks per pass
  const int nfull = d / kX3KBlock;     // those whose 32 columns all exist (nkb or nkb - 1)
  const int64_t my_tiles = tile0 < ntiles ? (ntiles - tile0 + tile_step - 1) / tile_step : 0;
  if (my_tiles == 0) return;
  const int64_t total = my_tiles * 4 * nkb; // even and >= 4
  f32x4 acc[4][NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int n = 0; n < NB; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  V4 ring[NBUF][8]; // NBUF blocks in flight (statically indexed: they live in registers)
  auto load_block = [&](V4 (&buf)[8], int64_t tile, int pass, int kb) {
    const float *ubase = X + tile * tile_stride + pass * 256 + wave * 64;
    if (EXP & 8) { // lab: the same bytes as 1 KB-contiguous wave loads (column (pass*nkb + kb)*8 + u, rows 256*wave + 4*lane)
#pragma unroll
      for (int u = 0; u < 8; u++)
        buf[u] = load16<float, true>(X + tile * tile_stride + (int64_t)((pass * nkb + kb) * 8 + u) * col_stride + 256 * wave + 4 * lane);
    } else if (kb < nfull) { // scalar column base + the lane's 32-bit byte offset
#pragma unroll
      for (int u = 0; u < 8; u++)
        buf[u] = load16<float, true>(reinterpret_cast<const float *>(
            reinterpret_cast<const char *>(ubase + (int64_t)(kb * kX3KBlock + u) * col_stride) + loff));
    } else { // ragged last block: a column >= d reads column d-1 instead (finite data) and meets zeros in Theta
      const int j0 = kb * kX3KBlock + 8 * g;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 + u < d ? j0 + u : d - 1;
        buf[u] = load16<float, true>(ubase + (int64_t)j * col_stride + 4 * arow);
      }
    }
  };
  bf16x8 fa[4], fb[4], fc[4];
  auto split_block = [&](const V4 (&buf)[8]) {
    if (EXP & 512) { // lab: whole vectors at once, packed subtractions -- fewer instructions, 4 % slower (tools/x3_ablate)
      x3_split_all(buf, fa, fb, fc);
    } else {
#pragma unroll
      for (int m = 0; m < 4; m++) x3_split(buf, m, fa[m], fb[m], fc[m]);
    }
  };
  auto mfma_block = [&](int kb) { // chain block outermost: three Theta fragments live at a time
    const u32x4 *tb = thetaL + (size_t)kb * NB * 3 * 64 + lane;
#pragma unroll
    for (int n = 0; n < NB; n++) {
      const bf16x8 p = __builtin_bit_cast(bf16x8, tb[(n * 3 + 0) * 64]);
      const bf16x8 q = __builtin_bit_cast(bf16x8, tb[(n * 3 + 1) * 64]);
      const bf16x8 r = __builtin_bit_cast(bf16x8, tb[(n * 3 + 2) * 64]);
#pragma unroll
      for (int m = 0; m < 4; m++) {
        f32x4 c = acc[m][n];
        if (TERMS >= 9) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[m], r, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[m], q, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[m], r, c, 0, 0, 0);
        }
        if (TERMS >= 6) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[m], p, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[m], q, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], r, c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[m], p, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], q, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], p, c, 0, 0, 0);
        acc[m][n] = c;
      }
    }
  };
  // (tile, pass, kb) of the linear block index, kept incrementally
  struct Pos {
    int64_t tile;
    int pass, kb;
  };
  Pos w{tile0, 0, 0}; // block being worked on
  Pos l{tile0, 0, 0}; // next block to request
  auto advance = [&](Pos &q) {
    if (++q.kb == nkb) {
      q.kb = 0;
      if (++q.pass == 4) {
        q.pass = 0;
        q.tile += tile_step;
      }
    }
  };
  // Interleaved form (EXP & 16): the split of row set m+1 is issued between the MFMAs of row set m, so the VALU work
  // of the split hides under the matrix pipe inside one wave (an MFMA occupies the issue port 4 of its 16 cycles).
  // Row set outermost, so all of the block's Theta fragments stay in registers for its duration; only two row
  // sets' fragments are live at a time.  Per accumulator the MFMAs come in the same order as in mfma_block.
  auto mfma_rowset = [&](int m, const bf16x8 (&T)[NB][3], const bf16x8 &a, const bf16x8 &b, const bf16x8 &c3) {
    f32x4 c[NB];
#pragma unroll
    for (int n = 0; n < NB; n++) c[n] = acc[m][n];
#pragma unroll
    for (int n = 0; n < NB; n++) c[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c3, T[n][0], c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; n++) c[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, T[n][1], c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; n++) c[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, T[n][2], c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; n++) c[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, T[n][0], c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; n++) c[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, T[n][1], c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; n++) c[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, T[n][0], c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NB; n++) acc[m][n] = c[n];
  };
  auto finish_ilv = [&](V4 (&buf)[8], bool more) {
    static_assert(!(EXP & 16) || TERMS == 6, "the interleaved form is written for six terms");
    bf16x8 T[NB][3];
    const u32x4 *tb = thetaL + (size_t)w.kb * NB * 3 * 64 + lane;
#pragma unroll
    for (int n = 0; n < NB; n++)
#pragma unroll
      for (int t = 0; t < 3; t++) T[n][t] = __builtin_bit_cast(bf16x8, tb[(n * 3 + t) * 64]);
    bf16x8 a0, b0, c0, a1, b1, c1;
    x3_split(buf, 0, a0, b0, c0);
    __builtin_amdgcn_sched_barrier(0);
    x3_split(buf, 1, a1, b1, c1); // under the MFMAs of row set 0
    mfma_rowset(0, T, a0, b0, c0);
    __builtin_amdgcn_sched_barrier(0);
    x3_split(buf, 2, a0, b0, c0);
    mfma_rowset(1, T, a1, b1, c1);
    __builtin_amdgcn_sched_barrier(0);
    x3_split(buf, 3, a1, b1, c1);
    mfma_rowset(2, T, a0, b0, c0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) { // the buffer has been read out: request its successor-but-one
      load_block(buf, l.tile, l.pass, l.kb);
      advance(l);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_rowset(3, T, a1, b1, c1);
    __builtin_amdgcn_sched_barrier(0);
    if (w.kb == nkb - 1) {
      epi(w.tile, w.pass * 256 + wave * 64, acc);
      zero_acc();
      if (w.pass == 3) tile_done(w.tile);
    }
    advance(w);
  };
  auto finish = [&](V4 (&buf)[8], bool more) { // work on the block in buf; request its successor-but-one
    if constexpr ((EXP & 16) != 0) {
      finish_ilv(buf, more);
      return;
    }
    if (EXP & 256) __builtin_amdgcn_s_barrier(); // lab: the four waves of a workgroup stay on the same (pass, k-block)
    if (EXP & 4) { // lab: loads only
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int m = 0; m < 4; m++) acc[m][0][u & 3] += buf[u][m];
    } else if (EXP & 64) { // lab: no split -- the fragments are the loaded registers as they are (the loads stay live)
#pragma unroll
      for (int m = 0; m < 4; m++) {
        fa[m] = __builtin_bit_cast(bf16x8, u32x4{__float_as_uint(buf[0][m]), __float_as_uint(buf[1][m]), __float_as_uint(buf[2][m]), __float_as_uint(buf[3][m])});
        fb[m] = __builtin_bit_cast(bf16x8, u32x4{__float_as_uint(buf[4][m]), __float_as_uint(buf[5][m]), __float_as_uint(buf[6][m]), __float_as_uint(buf[7][m])});
        fc[m] = fa[m];
      }
    } else {
      split_block(buf);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      if (!(EXP & 2)) load_block(buf, l.tile, l.pass, l.kb);
      advance(l);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (EXP & 32) { // lab: no MFMAs -- the split's results are kept live, the accumulators take one word of each
#pragma unroll
      for (int m = 0; m < 4; m++) {
        __asm__ volatile("" ::"v"(fa[m]), "v"(fb[m]), "v"(fc[m]));
        acc[m][0][0] += __uint_as_float(__builtin_bit_cast(u32x4, fc[m])[0]);
      }
    } else if (!(EXP & 4))
      mfma_block(w.kb);
    __builtin_amdgcn_sched_barrier(0);
    if (w.kb == nkb - 1) {
      epi(w.tile, w.pass * 256 + wave * 64, acc);
      zero_acc();
      if (w.pass == 3) tile_done(w.tile);
    }
    advance(w);
  };
  zero_acc();
  static_assert(NBUF >= 2 && NBUF <= 8, "two to eight load buffers");
  if constexpr (NBUF == 2) { // the production shape, written out (the general form below costs it registers)
    V4(&x)[8] = ring[0];
    V4(&y)[8] = ring[1];
    if (EXP & 2) {
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = y[u] = V4{1.f + lane, 2.f, 3.f, 4.f + u};
      advance(l);
      advance(l);
    } else {
      load_block(x, l.tile, l.pass, l.kb);
      advance(l);
      load_block(y, l.tile, l.pass, l.kb);
      advance(l);
    }
    for (int64_t b = 0; b + 3 < total; b += 2) {
      finish(x, true);
      finish(y, true);
    }
    finish(x, false);
    finish(y, false);
  } else {
#pragma unroll
    for (int b = 0; b < NBUF; b++) { // total >= 4: with more buffers than blocks the surplus requests are skipped
      if (EXP & 2) {
#pragma unroll
        for (int u = 0; u < 8; u++) ring[b][u] = V4{1.f + lane, 2.f, 3.f, 4.f + u};
      } else if (b < total) {
        load_block(ring[b], l.tile, l.pass, l.kb);
      }
      advance(l);
    }
    // full rounds while every finished block still has a successor NBUF blocks ahead, then the last NBUF .. 2 NBUF - 1
    int64_t done = 0;
    for (; done + 2 * NBUF <= total; done += NBUF) {
#pragma unroll
      for (int b = 0; b < NBUF; b++) finish(ring[b], true);
    }
    const int rem = (int)(total - done);
#pragma unroll
    for (int i = 0; i < 2 * NBUF - 1; i++)
      if (i < rem) finish(ring[i % NBUF], i + NBUF < rem);
  }
}

// Workgroup = 4 waves; grid = a fixed number of workgroups (2 per CU), workgroup g takes the 1024-row tiles
// g, g + gridDim.x, ...  NB = number of 16-chain blocks (1..4).  theta_dev: the LDS image
// (d_pad/32 * NB * 3 * 64 * 16 bytes).  Partials as lik_batched_f32_kernel: partials[chain*pstride + tile].
template <int NB, int TERMS, int EXP = 0, int NBUF = 2>
__global__ __launch_bounds__(kBlock, NBUF > 3 ? 1 : 2) void lik_batched_x3_kernel(const float *__restrict__ X, TileGeom g,
                                                                   const int32_t *__restrict__ tile_valid,
                                                                   const u32x4 *__restrict__ theta_dev, int d_pad,
                                                                   double *__restrict__ partials, int64_t pstride,
                                                                   int K, int64_t ntiles)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4 *thetaL = reinterpret_cast<u32x4 *>(smem);
  const int theta_vecs = d_pad / kX3KBlock * NB * 3 * 64;
  double *red = reinterpret_cast<double *>(smem + (size_t)theta_vecs * sizeof(u32x4)); // [4 waves][NB][64 lanes]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // uniform: the column bases of the loads stay scalar
  for (int i = threadIdx.x; i < theta_vecs; i += kBlock) thetaL[i] = theta_dev[i];
  __syncthreads();

  const int grp = lane >> 4;
  SoftplusAccF sp[NB]; // product-form epilogue: a lane's 64 elements per chain block and tile (16 per pass)
  // accumulator (m, n, r) of lane l: tile row row0 + 4*(4*(l>>4) + r) + m, chain 16n + (l&15)
  auto epi = [&](int64_t tile, int row0, const f32x4 (&acc)[4][NB]) {
    const int valid = tile_valid[tile];
    auto each = [&](auto full_tag, auto fn) { // only a shard's last tile is ragged: the full case has no predicates
      constexpr bool kFull = decltype(full_tag)::value;
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const bool ok = kFull || (row0 + 4 * (4 * grp + r) + m < valid);
#pragma unroll
          for (int n = 0; n < NB; n++)
            if (ok) fn(n, acc[m][n][r]);
        }
    };
    auto add = [&](int n, float x) {
      if (EXP & 1)
        sp[n].lin += x;
      else if (EXP & 128) // lab: one v_exp_f32 and one v_log_f32 per element
        sp[n].lin -= neg_log1pexp_fast(x);
      else
        sp[n].add(x);
    };
    auto flag = [&](int n, float x) { sp[n].flag_overflow(x); };
    const bool full = row0 + 64 <= valid;
    if (full)
      each(std::true_type{}, add);
    else
      each(std::false_type{}, add);
    if (!(EXP & 1)) {
      bool look = false;
#pragma unroll
      for (int n = 0; n < NB; n++) look |= sp[n].lin > SoftplusAccF::kOverflow; // the tile's sum so far: a superset
      if (look) each(std::false_type{}, flag);
    }
  };
  // per-chain sum over the tile: [wave][n][lane] -> thread per chain, fixed order
  auto tile_done = [&](int64_t tile) {
#pragma unroll
    for (int n = 0; n < NB; n++) {
      red[(wave * NB + n) * 64 + lane] = (double)((EXP & 1) ? sp[n].lin : sp[n].neg_sum());
      sp[n].reset();
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
      if (chain < K) partials[(int64_t)chain * pstride + tile] = s;
    }
    __syncthreads();
  };
  batched_x3_stream<NB, TERMS, EXP, NBUF>(X, g.tile_stride, (int64_t)blockIdx.x, (int64_t)gridDim.x, ntiles, wave,
                                          g.col_stride, g.d, d_pad, thetaL, lane, epi, tile_done);
}

// piece t (0, 1, 2) of v as a truncated bfloat16: the top 16 bits of v, of v - piece 0, of v - piece 0 - piece 1
__device__ __forceinline__ unsigned x3_piece(float v, int t)
{
  for (int s = 0; s < t; s++) v = x3_residual(v);
  return __float_as_uint(v) >> 16;
}

// The LDS image of Theta for the split kernels, built on the device from the caller's row-major Theta
// (raw[chain*d + j], float or double: a double is rounded to float first, as the single-chain kernels do):
//   out[(((kb*NB + n)*3 + t)*64 + lane)*4 + i] = piece_t(Theta[kb*32 + 8g + 2i][16n + c]) | piece_t(Theta[.. + 1][..]) << 16
// with c = lane & 15, g = lane >> 4; zero outside d columns / K chains.  Grid-stride over the dwords.
template <typename TIn>
__device__ __forceinline__ void x3_theta_image(const TIn *__restrict__ raw, int d, int K, int d_pad, int NB,
                                               unsigned *__restrict__ out, int64_t first, int64_t step)
{
  const int64_t total = (int64_t)d_pad / kX3KBlock * NB * 3 * 64 * 4;
  for (int64_t o = first; o < total; o += step) {
    const int i = (int)(o & 3), lane = (int)((o >> 2) & 63);
    int64_t rest = o >> 8;
    const int t = (int)(rest % 3);
    rest /= 3;
    const int n = (int)(rest % NB), kb = (int)(rest / NB);
    const int c = lane & 15, g = lane >> 4, chain = n * 16 + c;
    unsigned w = 0;
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const int j = kb * kX3KBlock + 8 * g + 2 * i + e;
      const float v = (j < d && chain < K) ? (float)raw[(int64_t)chain * d + j] : 0.f;
      w |= x3_piece(v, t) << (16 * e);
    }
    out[o] = w;
  }
}

template <typename TIn>
__global__ __launch_bounds__(kBlock) void x3_theta_image_kernel(const TIn *__restrict__ raw, int d, int K, int d_pad,
                                                                int NB, unsigned *__restrict__ out)
{
  x3_theta_image(raw, d, K, d_pad, NB, out, (int64_t)blockIdx.x * kBlock + threadIdx.x, (int64_t)gridDim.x * kBlock);
}

} // namespace mhlik

// Host: the LDS image of Theta for lik_batched_x3_kernel.  Theta[j][chain] = theta[j*sj + chain*sc], K <= 16*NB:
//   out[(((kb*NB + n)*3 + t)*64 + lane)*4 + i] = pack(piece_t(Theta[kb*32 + 8g + 2i][16n + c]), piece_t(Theta[.. + 1][..]))
// with c = lane & 15, g = lane >> 4; d_pad/32 * NB * 3 * 64 * 4 dwords.
static inline void mhlik_x3_stage_theta(const float *theta, size_t sj, size_t sc, int d, int K, int d_pad, int NB,
                                        uint32_t *out)
{
  auto piece = [](float v, int t) -> uint32_t {
    for (int s = 0; s < t; s++) {
      uint32_t u;
      memcpy(&u, &v, 4);
      u &= 0xffff0000u;
      float a;
      memcpy(&a, &u, 4);
      v -= a;
    }
    uint32_t u;
    memcpy(&u, &v, 4);
    return u >> 16;
  };
  for (int kb = 0; kb < d_pad / 32; kb++)
    for (int n = 0; n < NB; n++)
      for (int t = 0; t < 3; t++)
        for (int lane = 0; lane < 64; lane++)
          for (int i = 0; i < 4; i++) {
            const int c = lane & 15, g = lane >> 4, chain = n * 16 + c;
            uint32_t w = 0;
            for (int e = 0; e < 2; e++) {
              const int j = kb * 32 + 8 * g + 2 * i + e;
              const float v = (j < d && chain < K) ? theta[(size_t)j * sj + (size_t)chain * sc] : 0.f;
              w |= piece(v, t) << (16 * e);
            }
            out[((((size_t)kb * NB + n) * 3 + t) * 64 + lane) * 4 + i] = w;
          }
}
