// x3_lab.hip -- timing and accuracy of the split-bf16 K-chain float pass
// (mcmc_amd/csrc/mhlik_batched_x3.hpp) against the f32-MFMA pass and a double
// reference on the host (development tool).  Usage: x3_lab [N] [d] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "x3_lab_variants.hpp"
#include "batched_lab_variants.hpp"
using namespace mhlik;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill(float *p, size_t n, unsigned seed)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = (float)((x & 0xffffff) / 16777216.0 - 0.5);
  }
}

template <typename F> float time_it(F f, int reps)
{
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char **argv)
{
  const long N = argc > 1 ? atol(argv[1]) : 10000000;
  const int d = argc > 2 ? atoi(argv[2]) : 128;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  const int K = 64, NB = 4;
  const double flops = 2.0 * N * d * K;
  const int tile_rows = 1024; const long ntiles = (N + tile_rows - 1) / tile_rows;
  float *X; CK(hipMalloc(&X, sizeof(float) * ntiles * tile_rows * d));
  fill<<<4096, 256>>>(X, (size_t)ntiles * tile_rows * d, 1u);
  std::vector<int> valid(ntiles, tile_rows); valid[ntiles - 1] = (int)(N - (ntiles - 1) * tile_rows);
  int *tv; CK(hipMalloc(&tv, sizeof(int) * ntiles)); CK(hipMemcpy(tv, valid.data(), sizeof(int) * ntiles, hipMemcpyHostToDevice));
  double *P; CK(hipMalloc(&P, sizeof(double) * ntiles * 64));
  TileGeom g{(int64_t)tile_rows * d, tile_rows, d};

  // logical Theta[d][64]
  std::vector<float> theta((size_t)d * K);
  unsigned s = 12345u;
  for (auto &v : theta) { s = s * 1664525u + 1013904223u; v = (float)((s >> 8) / 16777216.0 - 0.5) * 0.5f; }
  // f32 image: thetaL[(j*32 + c)*2 + n] = Theta[j][n*32 + c]
  std::vector<float> img32((size_t)d * 64);
  for (int j = 0; j < d; j++) for (int c = 0; c < 32; c++) for (int n = 0; n < 2; n++) img32[((size_t)j * 32 + c) * 2 + n] = theta[(size_t)j * K + n * 32 + c];
  float *th32; CK(hipMalloc(&th32, sizeof(float) * d * 64)); CK(hipMemcpy(th32, img32.data(), sizeof(float) * d * 64, hipMemcpyHostToDevice));
  const int d_pad = (d + 31) / 32 * 32;
  std::vector<uint32_t> img3((size_t)d_pad / 32 * NB * 3 * 64 * 4);
  mhlik_x3_stage_theta(theta.data(), K, 1, d, K, d_pad, NB, img3.data());
  u32x4 *th3; CK(hipMalloc(&th3, img3.size() * 4)); CK(hipMemcpy(th3, img3.data(), img3.size() * 4, hipMemcpyHostToDevice));

  // double reference over the first T tiles: per chain, sum of dots and sum of -log(1+exp(dot))
  const long T = ntiles < 16 ? ntiles : 16;
  std::vector<float> hx((size_t)T * tile_rows * d);
  CK(hipMemcpy(hx.data(), X, hx.size() * 4, hipMemcpyDeviceToHost));
  std::vector<double> ref_dot((size_t)K * T, 0.0), ref_lik((size_t)K * T, 0.0), ref_abs((size_t)K * T, 0.0);
  for (long t = 0; t < T; t++)
    for (int i = 0; i < valid[t]; i++)
      for (int c = 0; c < K; c++) {
        double acc = 0, aa = 0;
        for (int j = 0; j < d; j++) { double pr = (double)hx[((size_t)t * d + j) * tile_rows + i] * (double)theta[(size_t)j * K + c]; acc += pr; aa += fabs(pr); }
        ref_dot[(size_t)c * T + t] += acc; ref_abs[(size_t)c * T + t] += aa; ref_lik[(size_t)c * T + t] += -log1p(exp(acc));
      }

  auto report = [&](const char *label, float ms, bool plain) {
    std::vector<double> h((size_t)ntiles * 64);
    CK(hipMemcpy(h.data(), P, sizeof(double) * ntiles * 64, hipMemcpyDeviceToHost));
    double worst = 0, tot = 0, tref = 0;
    for (int c = 0; c < K; c++) for (long t = 0; t < T; t++) {
      const double got = h[(size_t)c * ntiles + t];
      const double want = plain ? ref_dot[(size_t)c * T + t] : ref_lik[(size_t)c * T + t];
      const double scale = plain ? ref_abs[(size_t)c * T + t] : fabs(want);
      const double e = fabs(got - want) / scale; if (e > worst) worst = e;
      tot += got; tref += want;
    }
    printf("%-34s %.3f ms  %6.1f TF-equivalent  %6.0f GB/s   max err per (chain,tile) %.2e  total rel %.2e\n", label, ms, flops / ms / 1e9,
           (double)N * d * 4 / ms / 1e6, worst, fabs(tot - tref) / fabs(tref));
  };
  size_t lds32 = sizeof(float) * d * 64 + sizeof(double) * 4 * 2 * 64;
  size_t lds3 = img3.size() * 4 + sizeof(double) * 4 * 4 * 64;
#define RUN32(EXPV, label) { auto k = lab_lik_batched_f32_kernel<2, 8, EXPV>; \
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32)); \
    float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(ntiles), dim3(256), lds32, 0, X, g, tv, th32, d, P, (int64_t)ntiles, K); }, reps); \
    CK(hipDeviceSynchronize()); report(label, ms, (EXPV) & 1); }
#define RUNX3(TERMSV, EXPV, label) RUNX3D(TERMSV, EXPV, 0, label)
#define RUNX3D(TERMSV, EXPV, DBV, label) { auto k = lab_lik_batched_x3_kernel<4, TERMSV, EXPV>; \
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3)); \
    float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(ntiles < 512 ? ntiles : 512), dim3(256), lds3, 0, X, g, tv, th3, d_pad, P, (int64_t)ntiles, K, (int64_t)ntiles); }, reps); \
    CK(hipDeviceSynchronize()); report(label, ms, (EXPV) & 1); }
  RUN32(1, "f32 MFMA, plain sum of dots")
  RUNX3(3, 1, "x3 3 terms, plain sum of dots")
  RUNX3(6, 1, "x3 6 terms, plain sum of dots")
  RUNX3(9, 1, "x3 9 terms, plain sum of dots")
  RUN32(0, "f32 MFMA")
  RUNX3(3, 0, "x3 3 terms")
  RUNX3(6, 0, "x3 6 terms")
  RUNX3(9, 0, "x3 9 terms")
  RUNX3(6, 2, "x3 6 terms, no loads in k-loop")
  RUNX3(9, 2, "x3 9 terms, no loads in k-loop")
  RUNX3(6, 3, "x3 6 terms, no loads, plain sum")
  RUNX3(6, 5, "x3 loads only, plain sum")
  RUNX3(6, 0, "x3 6 terms (again)")
#define RUNX3B(label) { auto k = lab_lik_batched_x3_kernel<4, 6, 0, 3>; \
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3)); \
    float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(ntiles < 512 ? ntiles : 512), dim3(256), lds3, 0, X, g, tv, th3, d_pad, P, (int64_t)ntiles, K, (int64_t)ntiles); }, reps); \
    CK(hipDeviceSynchronize()); report(label, ms, false); }
  RUNX3B("x3 6 terms, three load buffers")
  RUNX3(6, 16, "x3 6 terms, split under the MFMAs")
  RUNX3(6, 0, "x3 6 terms (plain again)")
  RUNX3(6, 16, "x3 6 terms, split under the MFMAs")
  // occupancy experiments: loads only, one chain block (few registers), 2 / 4 / 8 workgroups per CU
#define RUNOCC(PERCU, label) { auto k = lab_lik_batched_x3_kernel<1, 6, 5>; \
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3)); \
    const long gr = ntiles < 256 * PERCU ? ntiles : 256 * PERCU; \
    float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(gr), dim3(256), (size_t)(lds3 / 4 + 4096), 0, X, g, tv, th3, d_pad, P, (int64_t)ntiles, 16, (int64_t)ntiles); }, reps); \
    CK(hipDeviceSynchronize()); report(label, ms, true); }
#define RUNPAT(EXPV, GRID, label) { auto k = lab_lik_batched_x3_kernel<1, 6, EXPV>; \
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3)); \
    const long gr = (GRID); \
    float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(gr), dim3(256), (size_t)(lds3 / 4 + 4096), 0, X, g, tv, th3, d_pad, P, (int64_t)ntiles, 16, (int64_t)ntiles); }, reps); \
    CK(hipDeviceSynchronize()); report(label, ms, true); }
  RUNPAT(5, ntiles, "loads only, x3 pattern, 1 tile/WG")
  RUNPAT(13, ntiles, "loads only, 1KB pattern, 1 tile/WG")
  RUNPAT(13, 512, "loads only, 1KB pattern, 2 WG/CU")
  RUNPAT(13, 1024, "loads only, 1KB pattern, 4 WG/CU")
  RUNPAT(5, 1000, "loads only, x3 pattern, grid 1000")
  RUNOCC(2, "loads only NB=1, 2 WG/CU")
  RUNOCC(3, "loads only NB=1, 3 WG/CU")
  RUNOCC(4, "loads only NB=1, 4 WG/CU")
  RUNOCC(6, "loads only NB=1, 6 WG/CU")
  RUNOCC(8, "loads only NB=1, 8 WG/CU")
  return 0;
}
