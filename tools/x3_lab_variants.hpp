// x3_lab_variants.hpp -- LAB ONLY: the split-bf16 K-chain float pass of mcmc_amd/csrc/mhlik_batched_x3.hpp with the
// experiment switches of rounds 1-3 (EXP bits: 1 plain sum epilogue, 2 no loads, 4 loads only, 8 1 KB-contiguous wave
// loads, 16 split interleaved with the MFMAs, 32 no MFMAs, 64 no split, 128 exp+log epilogue, 256 barrier per k-block,
// 512 packed-subtraction split).  tools/x3_ablate.hip and tools/x3_lab.hip instantiate these; the product header holds
// the production shape alone and nothing in mcmc_amd/ includes this file.  With EXP = 0 the stream IS the product's
// (a call into mhlik_batched_x3.hpp), and tests/test_x3_lab_matches_product.py checks that the lab kernel built that
// way is instruction for instruction the product kernel: the "production" row of the ablation tables is the product.
#pragma once

#include "mhlik_batched_x3.hpp"

namespace mhlik {

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
__device__ __forceinline__ void lab_batched_x3_stream(const float *__restrict__ X, int64_t tile_stride, int64_t tile0,
                                                  int64_t tile_step, int64_t ntiles, int wave, int64_t col_stride,
                                                  int d, int d_pad, const u32x4 *thetaL, int lane, Epi epi,
                                                  Done tile_done)
{
  if constexpr (EXP == 0) { // every switch off: the product's own stream, not a copy of it
    batched_x3_stream<NB, TERMS, NBUF>(X, tile_stride, tile0, tile_step, ntiles, wave, col_stride, d, d_pad, thetaL, lane, epi,
                                       tile_done);
    return;
  }
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
__global__ __launch_bounds__(kBlock, NBUF > 3 ? 1 : 2) void lab_lik_batched_x3_kernel(const float *__restrict__ X, TileGeom g,
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
  lab_batched_x3_stream<NB, TERMS, EXP, NBUF>(X, g.tile_stride, (int64_t)blockIdx.x, (int64_t)gridDim.x, ntiles, wave,
                                          g.col_stride, g.d, d_pad, thetaL, lane, epi, tile_done);
}

} // namespace mhlik
