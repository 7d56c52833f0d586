// x3_ablate.hip -- which part of the split-bf16 K-chain float pass (mcmc_amd/csrc/mhlik_batched_x3.hpp) costs
// what: the production instantiation with its loads, its split, its MFMAs or its epilogue taken out, one at a
// time, on one box in one run (development tool).  Usage: x3_ablate [N] [d] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "x3_lab_variants.hpp"
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

#define RUN(EXPV, label) { auto k = lab_lik_batched_x3_kernel<4, 6, EXPV>; \
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3)); \
    float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(ntiles < 512 ? ntiles : 512), dim3(256), lds3, 0, X, g, tv, th3, d_pad, P, (int64_t)ntiles, K, (int64_t)ntiles); }, reps); \
    CK(hipDeviceSynchronize()); printf("%-46s %.3f ms  %6.0f GB/s\n", label, ms, (double)N * d * 4 / ms / 1e6); }
  size_t lds3 = img3.size() * 4 + sizeof(double) * 4 * 4 * 64;
  (void)flops; (void)th32;
#define RUNN(NBUFV, label) { auto k = lab_lik_batched_x3_kernel<4, 6, 0, NBUFV>; \
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3)); \
    float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(ntiles < 256 ? ntiles : 256), dim3(256), lds3, 0, X, g, tv, th3, d_pad, P, (int64_t)ntiles, K, (int64_t)ntiles); }, reps); \
    CK(hipDeviceSynchronize()); printf("%-46s %.3f ms  %6.0f GB/s\n", label, ms, (double)N * d * 4 / ms / 1e6); }
#define RUNNL(NBUFV, label) { auto k = lab_lik_batched_x3_kernel<4, 6, 5, NBUFV>; \
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3)); \
    float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(ntiles < 256 ? ntiles : 256), dim3(256), lds3, 0, X, g, tv, th3, d_pad, P, (int64_t)ntiles, K, (int64_t)ntiles); }, reps); \
    CK(hipDeviceSynchronize()); printf("%-46s %.3f ms  %6.0f GB/s\n", label, ms, (double)N * d * 4 / ms / 1e6); }
  for (int round = 0; round < 2; round++) {
    RUNN(4, "one wave per SIMD, four load buffers")
    RUNNL(4, "one wave per SIMD, four buffers, loads only")
    RUN(0, "production")
    RUN(8, "production, 1 KB-contiguous wave loads")
    RUN(13, "loads only, 1 KB-contiguous wave loads")
    RUN(256, "production + workgroup barrier per k-block")
    RUN(128, "exp + log per element (the round-1 epilogue)")
    RUN(512, "packed-subtraction split (v_pk_add_f32)")
    RUN(640, "packed split, round-1 epilogue")
    RUN(1, "plain sum instead of the softplus epilogue")
    RUN(2, "no loads in the k-loop")
    RUN(5, "loads only")
    RUN(32, "no MFMAs (loads, split, epilogue)")
    RUN(64, "no split (loads, MFMAs, epilogue)")
    RUN(33, "no MFMAs, plain sum (loads, split)")
    RUN(65, "no split, plain sum (loads, MFMAs)")
  }
  return 0;
}
