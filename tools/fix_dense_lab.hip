// fix_dense_lab.hip -- the two mixed-precision correction kernels of mcmc_amd/csrc/mhlik_mp_batched.hpp on their own,
// on the same stores, indicators and Theta: mpb_fix_kernel (one row per lane, Theta broadcast from LDS) against
// mpb_fix_dense_kernel (matrix cores, all chains on one phase).  Their double dots differ in the last place (unfused
// chain against the MFMA's fused one), so a handful of indicators may differ; the sums agree to ~1e-15.
// Development tool.  Usage: fix_dense_lab [N] [d] [block_rows] [F] [reps] [K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "mhlik_mp_batched.hpp"
using namespace mhlik;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// value of (row, col): the same in both stores (the float one rounded)
__device__ inline double cell(int64_t row, int col)
{
  uint64_t x = (uint64_t)row * 0x9E3779B97F4A7C15ull + (uint64_t)col * 0xC2B2AE3D27D4EB4Full + 12345u;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
  return (double)(x >> 11) / 9007199254740992.0 - 0.5;
}
__global__ void fill_f(float *X, int64_t tiles, int d)
{
  const int64_t n = tiles * 1024 * d;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tile = i / (1024 * (int64_t)d), in = i % (1024 * (int64_t)d);
    const int col = (int)(in / 1024), r = (int)(in % 1024);
    X[i] = (float)cell(tile * 1024 + r, col);
  }
}
__global__ void fill_d(double *X, int64_t tiles, int d)
{
  const int64_t n = tiles * 512 * d;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t tile = i / (512 * (int64_t)d), in = i % (512 * (int64_t)d);
    const int col = (int)(in / 512), r = (int)(in % 512);
    X[i] = cell(tile * 512 + r, col);
  }
}

int main(int argc, char **argv)
{
  const int64_t N = argc > 1 ? atol(argv[1]) : 10000000;
  const int d = argc > 2 ? atoi(argv[2]) : 128;
  const int block_rows = argc > 3 ? atoi(argv[3]) : 64;
  const int F = argc > 4 ? atoi(argv[4]) : 64;
  const int reps = argc > 5 ? atoi(argv[5]) : 20;
  const int K = argc > 6 ? atoi(argv[6]) : 64;
  const int64_t tf = (N + 1023) / 1024, td = (N + 511) / 512;
  float *Xf; double *Xd;
  CK(hipMalloc(&Xf, sizeof(float) * tf * 1024 * d));
  CK(hipMalloc(&Xd, sizeof(double) * td * 512 * d));
  fill_f<<<4096, 256>>>(Xf, tf, d);
  fill_d<<<4096, 256>>>(Xd, td, d);
  CK(hipDeviceSynchronize());
  TileGeom gf{(int64_t)1024 * d, 1024, d}, gd{(int64_t)512 * d, 512, d};
  const int d_fix = (d + 3) / 4 * 4;
  std::vector<double> th((size_t)K * d_fix, 0.0);
  std::vector<float> thf((size_t)K * d_fix, 0.f);
  unsigned s = 777u;
  for (int k = 0; k < K; k++)
    for (int j = 0; j < d; j++) {
      s = s * 1664525u + 1013904223u;
      th[(size_t)k * d_fix + j] = ((s >> 8) / 16777216.0 - 0.5) * 0.3;
      thf[(size_t)k * d_fix + j] = (float)th[(size_t)k * d_fix + j];
    }
  double *Th; float *Thf;
  CK(hipMalloc(&Th, th.size() * 8)); CK(hipMemcpy(Th, th.data(), th.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&Thf, thf.size() * 4)); CK(hipMemcpy(Thf, thf.data(), thf.size() * 4, hipMemcpyHostToDevice));
  MpbParams p{};
  p.block_rows = block_rows; p.block_shift = __builtin_ctz(block_rows);
  p.blocks = (int32_t)((N + block_rows - 1) / block_rows);
  p.resamp_factor = F; p.K = K; p.u_call = 5; p.eps = 1.7e-7; p.n_rows = N; p.row_base = 0; p.seg_groups = kMpbFixGroups; p.nseg = 1;
  MpbSeg seg{0, p.blocks, 0, 0};
  MpbSeg *segtab; CK(hipMalloc(&segtab, sizeof(seg))); CK(hipMemcpy(segtab, &seg, sizeof(seg), hipMemcpyHostToDevice));
  MpbChainArgs ca{};
  for (int k = 0; k < 64; k++) { ca.choose[k] = 3 % F; ca.seed[k] = 1000 + 17 * k; }
  MpbChainArgs *cad; CK(hipMalloc(&cad, sizeof(ca))); CK(hipMemcpy(cad, &ca, sizeof(ca), hipMemcpyHostToDevice));
  const size_t zbytes = (size_t)p.blocks * 64;
  int8_t *z0, *zA, *zB;
  CK(hipMalloc(&z0, zbytes)); CK(hipMalloc(&zA, zbytes)); CK(hipMalloc(&zB, zbytes));
  double *accA, *accB; int32_t *cntA, *cntB;
  const size_t rows = (size_t)kMpbFixWaves * 64;
  CK(hipMalloc(&accA, rows * 8)); CK(hipMalloc(&accB, rows * 8)); CK(hipMalloc(&cntA, rows * 4)); CK(hipMalloc(&cntB, rows * 4));

  const size_t ldsA = (size_t)12 * K * d_fix;
  auto kA = mpb_fix_kernel<true>;
  CK(hipFuncSetAttribute((const void *)kA, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA));
  constexpr int LAB_D = kMpbDenseDepth, LAB_NT = kMpbFixBlock;
  const int d_padB = (d + 4 * LAB_D - 1) / (4 * LAB_D) * (4 * LAB_D);
  const int spb = block_rows / 64 < LAB_NT / 256 ? block_rows / 64 : LAB_NT / 256;
  int round_blocks = (int)((150 * 1024 - (size_t)8 * 64 * d_padB) / ((size_t)4096 * spb));
  if (round_blocks > kMpbDenseRound) round_blocks = kMpbDenseRound;
  const size_t ldsB = (size_t)8 * 64 * d_padB + (size_t)round_blocks * 4096 * spb;
  auto kB = mpb_fix_dense_kernel<LAB_NT, LAB_D>;
  CK(hipFuncSetAttribute((const void *)kB, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
  printf("N=%ld d=%d block_rows=%d F=%d K=%d blocks=%d round_blocks=%d ldsA=%zu ldsB=%zu D=%d\n", (long)N, d, block_rows, F, K, p.blocks,
         round_blocks, ldsA, ldsB, LAB_D); 
  auto runA = [&](int8_t *zz) { hipLaunchKernelGGL(kA, dim3(kMpbFixGroups), dim3(kMpbFixBlock), ldsA, 0, Xf, gf, Xd, gd, Th, Thf, d_fix, p, cad, segtab, zz, accA, cntA); };
  auto runB = [&](int8_t *zz) { hipLaunchKernelGGL(kB, dim3(kMpbFixGroups), dim3(LAB_NT), ldsB, 0, Xf, gf, Xd, gd, Th, d_fix, d_padB, p, cad, segtab, zz, accB, cntB, round_blocks); };

  std::vector<int8_t> hA(zbytes), hB(zbytes);
  std::vector<double> aA(rows), aB(rows);
  std::vector<int32_t> cA(rows), cB(rows);
  auto compare = [&](const char *what) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hA.data(), zA, zbytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(hB.data(), zB, zbytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(aA.data(), accA, rows * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(aB.data(), accB, rows * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cA.data(), cntA, rows * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cB.data(), cntB, rows * 4, hipMemcpyDeviceToHost));
    size_t zdiff = 0, bright = 0;
    for (size_t i = 0; i < zbytes; i++) { if ((i & 63) < (size_t)K) { zdiff += hA[i] != hB[i]; bright += hA[i] != 0; } }
    double worst = 0; long cdiff = 0;
    for (int k = 0; k < K; k++) {
      double sa = 0, sb = 0; long na = 0, nb = 0;
      for (size_t w = 0; w < (size_t)kMpbFixWaves; w++) { sa += aA[w * 64 + k]; sb += aB[w * 64 + k]; na += cA[w * 64 + k]; nb += cB[w * 64 + k]; }
      const double rel = fabs(sa - sb) / (fabs(sa) > 1e-300 ? fabs(sa) : 1.0);
      if (rel > worst) worst = rel;
      cdiff += labs(na - nb);
      if (k == 0) printf("  chain 0: bright sum %.17g / %.17g, bright count %ld / %ld\n", sa, sb, na, nb);
    }
    printf("%s: indicators that differ %zu (bright after: %zu), worst relative difference of a chain's bright sum %.3g, count differences %ld\n",
           what, zdiff, bright, worst, cdiff);
  };
  // 1. everything bright
  CK(hipMemset(z0, 1, zbytes));
  CK(hipMemcpy(zA, z0, zbytes, hipMemcpyDeviceToDevice)); CK(hipMemcpy(zB, z0, zbytes, hipMemcpyDeviceToDevice));
  runA(zA); runB(zB);
  compare("all bright");
  // 2. steady state: the indicators the first call left, phase 3
  CK(hipMemcpy(z0, zA, zbytes, hipMemcpyDeviceToDevice));
  p.u_call = 6;
  CK(hipMemcpy(zB, z0, zbytes, hipMemcpyDeviceToDevice));
  runA(zA); runB(zB);
  compare("steady state");
  // timing, steady state (indicators restored before every launch, outside the timed events)
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int which = 0; which < 4; which++) {
    float tot = 0, best = 1e9;
    for (int r = 0; r < reps; r++) {
      CK(hipMemcpyAsync(zA, z0, zbytes, hipMemcpyDeviceToDevice, 0));
      CK(hipEventRecord(e0, 0));
      if (which & 1) runB(zA); else runA(zA);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      tot += ms; if (ms < best) best = ms;
    }
    printf("%-44s mean %.1f us  best %.1f us\n", (which & 1) ? "mpb_fix_dense_kernel (16 rows x 64 chains)" : "mpb_fix_kernel (row per lane, 4 chains)", tot / reps * 1e3, best * 1e3);
  }
  return 0;
}
