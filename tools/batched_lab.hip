// batched_lab.hip -- stand-alone timing of batched-likelihood kernel variants
// (development tool; not part of the library).  Usage: batched_lab [N] [d] [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "batched_lab_v2.hpp"
#include "batched_lab_v3.hpp"
#include "batched_lab_f64_ilv.hpp"
#include "batched_lab_variants.hpp"
#include <string.h>
using namespace mhlik;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T> __global__ void fill(T *p, size_t n, unsigned seed)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (T)((x & 0xffff) / 65536.0 - 0.5);
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
  const int K = 64;
  const double flops = 2.0 * N * d * K;
  {
    const int tile_rows = 1024; const long ntiles = (N + tile_rows - 1) / tile_rows;
    float *X; CK(hipMalloc(&X, sizeof(float) * ntiles * tile_rows * d));
    fill<<<4096, 256>>>(X, (size_t)ntiles * tile_rows * d, 1u);
    std::vector<int> valid(ntiles, tile_rows); valid[ntiles - 1] = (int)(N - (ntiles - 1) * tile_rows);
    int *tv; CK(hipMalloc(&tv, sizeof(int) * ntiles)); CK(hipMemcpy(tv, valid.data(), sizeof(int) * ntiles, hipMemcpyHostToDevice));
    float *th; CK(hipMalloc(&th, sizeof(float) * d * 64)); fill<<<64, 256>>>(th, (size_t)d * 64, 7u);
    double *P; CK(hipMalloc(&P, sizeof(double) * ntiles * 64));
    TileGeom g{(int64_t)tile_rows * d, tile_rows, d};
    size_t lds = sizeof(float) * d * 64 + sizeof(double) * 4 * 2 * 64;
#define RUN32(EXPV, label) RUN32K(8, EXPV, label)
#define RUN32K(KUV, EXPV, label) { auto k = lab_lik_batched_f32_kernel<2, KUV, EXPV>; \
      CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(ntiles), dim3(256), lds, 0, X, g, tv, th, d, P, (int64_t)ntiles, K); }, reps); \
      printf("f32 %-28s %.3f ms  %.1f TF\n", label, ms, flops / ms / 1e9); }
    RUN32(0, "current")
    RUN32(1, "no transcendental epilogue")
    RUN32(2, "no global loads in k-loop")
    RUN32(3, "neither")
    RUN32K(4, 0, "KU=4")
    RUN32K(16, 0, "KU=16")
    RUN32(0, "current (again)")
    {
      std::vector<double> h1((size_t)ntiles * 64);
      auto k1 = lab_lik_batched_f32_kernel<2, 4, 0>;
      hipLaunchKernelGGL(k1, dim3(ntiles), dim3(256), lds, 0, X, g, tv, th, d, P, (int64_t)ntiles, K);
      CK(hipMemcpy(h1.data(), P, sizeof(double) * ntiles * 64, hipMemcpyDeviceToHost));
      double *P2; CK(hipMalloc(&P2, sizeof(double) * ntiles * 2 * 64));
      int cus = 256;
      const int64_t nunits = 2 * ntiles;
#define RUNV2X(KUV, EXPV, label) RUNV2Y(KUV, EXPV, 2, label)
#define RUNV2Y(KUV, EXPV, OCCV, label) { auto k = lik_batched2_f32_kernel<2, KUV, EXPV, OCCV>; \
        const int d_pad = (d + 2 * KUV - 1) / (2 * KUV) * (2 * KUV); \
        size_t lds2 = sizeof(float) * d_pad * 64 + sizeof(double) * 2 * 4 * 2 * 64 + ((EXPV & 32) ? 4 * 2 * KUV * 1024 : 0); \
        float *th2; CK(hipMalloc(&th2, sizeof(float) * d_pad * 64)); CK(hipMemset(th2, 0, sizeof(float) * d_pad * 64)); \
        CK(hipMemcpy(th2, th, sizeof(float) * d * 64, hipMemcpyDeviceToDevice)); \
        CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); \
        CK(hipMemset(P2, 0, sizeof(double) * nunits * 64)); \
        float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(OCCV * cus), dim3(256), lds2, 0, X, g, tv, th2, d_pad, P2, nunits, K, nunits); }, reps); \
        std::vector<double> h2((size_t)nunits * 64); \
        CK(hipMemcpy(h2.data(), P2, sizeof(double) * nunits * 64, hipMemcpyDeviceToHost)); \
        double worst = 0; \
        for (int c = 0; c < 64; c++) for (long t = 0; t < ntiles; t++) { \
          double a = h1[(size_t)c * ntiles + t], b2 = h2[(size_t)c * nunits + 2 * t] + h2[(size_t)c * nunits + 2 * t + 1]; \
          double e = fabs(a - b2) / fabs(a); if (e > worst) worst = e; } \
        printf("f32 %-28s %.3f ms  %.1f TF   max rel diff vs v1 per tile %.2e\n", label, ms, flops / ms / 1e9, worst); \
        CK(hipFree(th2)); }
      RUNV2X(4, 32, "v2 glds KU=4")
#define RUNV3(HINTV, label) if (d == 128) { auto k = lik_batched3_f32_kernel<8, HINTV>; \
        CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(ntiles), dim3(256), lds, 0, X, g, tv, th, d, P, (int64_t)ntiles, K); }, reps); \
        std::vector<double> h3((size_t)ntiles * 64); \
        CK(hipMemcpy(h3.data(), P, sizeof(double) * ntiles * 64, hipMemcpyDeviceToHost)); \
        double worst = 0; \
        for (size_t i = 0; i < h3.size(); i++) { double e = fabs(h1[i] - h3[i]) / fabs(h1[i]); if (e > worst) worst = e; } \
        printf("f32 %-28s %.3f ms  %.1f TF   max rel diff vs v1 per tile %.2e\n", label, ms, flops / ms / 1e9, worst); }
      RUNV3(0, "v3 interleaved epilogue")
      RUNV3(1, "v3 same + sched groups")
      RUN32(0, "current (again)")
      CK(hipFree(P2));
    }
    CK(hipFree(X)); CK(hipFree(tv)); CK(hipFree(th)); CK(hipFree(P));
  }
  {
    const int tile_rows = 512; const long ntiles = (N + tile_rows - 1) / tile_rows;
    double *X; CK(hipMalloc(&X, sizeof(double) * ntiles * tile_rows * d));
    fill<<<4096, 256>>>(X, (size_t)ntiles * tile_rows * d, 1u);
    std::vector<int> valid(ntiles, tile_rows); valid[ntiles - 1] = (int)(N - (ntiles - 1) * tile_rows);
    int *tv; CK(hipMalloc(&tv, sizeof(int) * ntiles)); CK(hipMemcpy(tv, valid.data(), sizeof(int) * ntiles, hipMemcpyHostToDevice));
    double *th; CK(hipMalloc(&th, sizeof(double) * d * 64)); fill<<<64, 256>>>(th, (size_t)d * 64, 7u);
    double *P; CK(hipMalloc(&P, sizeof(double) * ntiles * 64));
    TileGeom g{(int64_t)tile_rows * d, tile_rows, d};
    size_t lds = sizeof(double) * d * 64 + sizeof(double) * 4 * 4 * 64;
#define RUN64(EXPV, label) RUN64K(2, EXPV, label)
#define RUN64K(KUV, EXPV, label) { auto k = lab_lik_batched_f64_kernel<4, KUV, EXPV>; \
      CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(ntiles), dim3(256), lds, 0, X, g, tv, th, d, P, (int64_t)ntiles, K, d * 64); }, reps); \
      printf("f64 %-28s %.3f ms  %.1f TF\n", label, ms, flops / ms / 1e9); }
    std::vector<double> href((size_t)ntiles * 64);
#define RUNILV(NKV, EXPV, label) if (d == 4 * NKV) { auto k = lik_batched_f64_ilv_kernel<NKV, EXPV>; \
      CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(ntiles), dim3(256), lds, 0, X, g, tv, th, d, P, (int64_t)ntiles, K); }, reps); \
      std::vector<double> h((size_t)ntiles * 64); CK(hipMemcpy(h.data(), P, sizeof(double) * ntiles * 64, hipMemcpyDeviceToHost)); \
      size_t bad = 0; for (size_t i = 0; i < h.size(); i++) bad += memcmp(&h[i], &href[i], 8) != 0; \
      printf("f64 %-28s %.3f ms  %.1f TF   partials differing from the current kernel: %zu of %zu\n", label, ms, flops / ms / 1e9, bad, h.size()); }
    RUN64(0, "current (fast softplus)")
    CK(hipMemcpy(href.data(), P, sizeof(double) * ntiles * 64, hipMemcpyDeviceToHost));
    RUNILV(32, 0, "interleaved epilogue")
    RUNILV(32, 8, "interleaved + sched groups")
    RUNILV(16, 0, "interleaved epilogue")
    RUNILV(16, 8, "interleaved + sched groups")
    RUN64(0, "current (again)")
    RUN64(4, "ocml exp + log epilogue")
    RUN64(1, "no transcendental epilogue")
    RUN64(2, "no global loads in k-loop")
    RUN64(3, "neither")
    RUN64K(4, 0, "KU=4")
    RUN64(8, "product-form epilogue")
  }
  return 0;
}
