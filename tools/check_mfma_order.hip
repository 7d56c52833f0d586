// check_mfma_order.hip -- is v_mfma_f32_16x16x4_f32 a k-ordered fmaf chain, and which rows does a lane's
// result hold?  (development check behind mhlik_mp_batched.hpp's dense correction kernel)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// A: [16][K] row-major, B: [K][16]; out: [64 lanes][4]
__global__ void k32(const float *A, const float *B, int K, float *out)
{
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4) c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k0 + g], B[(k0 + g) * 16 + i], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}
__global__ void k64(const double *A, const double *B, int K, double *out)
{
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  f64x4 c = {0., 0., 0., 0.};
  for (int k0 = 0; k0 < K; k0 += 4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[i * K + k0 + g], B[(k0 + g) * 16 + i], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}

int main()
{
  const int K = 128;
  std::vector<float> A(16 * K), B(K * 16);
  std::vector<double> Ad(16 * K), Bd(K * 16);
  unsigned s = 7u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / 16777216.0 - 0.5; };
  for (int i = 0; i < 16 * K; i++) { Ad[i] = rnd() * 3.0; A[i] = (float)Ad[i]; }
  for (int i = 0; i < K * 16; i++) { Bd[i] = rnd(); B[i] = (float)Bd[i]; }
  float *dA, *dB, *dO; double *eA, *eB, *eO;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dO, 256 * 4));
  CK(hipMalloc(&eA, A.size() * 8)); CK(hipMalloc(&eB, B.size() * 8)); CK(hipMalloc(&eO, 256 * 8));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(eA, Ad.data(), A.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(eB, Bd.data(), B.size() * 8, hipMemcpyHostToDevice));
  k32<<<1, 64>>>(dA, dB, K, dO); k64<<<1, 64>>>(eA, eB, K, eO);
  CK(hipDeviceSynchronize());
  std::vector<float> O(256); std::vector<double> Od(256);
  CK(hipMemcpy(O.data(), dO, 256 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(Od.data(), eO, 256 * 8, hipMemcpyDeviceToHost));
  // references: sequential fmaf chain / fma chain / unfused chain, k = 0..K-1
  auto ref32 = [&](int i, int j) { float acc = 0.f; for (int k = 0; k < K; k++) acc = fmaf(A[i * K + k], B[k * 16 + j], acc); return acc; };
  auto ref64f = [&](int i, int j) { double acc = 0.; for (int k = 0; k < K; k++) acc = fma(Ad[i * K + k], Bd[k * 16 + j], acc); return acc; };
  auto ref64u = [&](int i, int j) { volatile double acc = 0.; for (int k = 0; k < K; k++) { volatile double p = Ad[i * K + k] * Bd[k * 16 + j]; acc = acc + p; } return (double)acc; };
  int rc = 0;
  for (int layout = 0; layout < 2; layout++) { // 0: row = 4*(l>>4) + r   1: row = (l>>4) + 4*r
    int ok32 = 0, ok64f = 0, ok64u = 0; double worst64 = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
      const int j = l & 15, i = layout == 0 ? 4 * (l >> 4) + r : (l >> 4) + 4 * r;
      float want = ref32(i, j); ok32 += memcmp(&want, &O[l * 4 + r], 4) == 0;
      double wf = ref64f(i, j), wu = ref64u(i, j);
      ok64f += memcmp(&wf, &Od[l * 4 + r], 8) == 0; ok64u += memcmp(&wu, &Od[l * 4 + r], 8) == 0;
      double e = fabs(Od[l * 4 + r] - wf) / fabs(wf); if (e > worst64) worst64 = e;
    }
    printf("layout %s: f32 16x16x4 == k-ordered fmaf chain on %d/256; f64 16x16x4 == fma chain on %d/256, == unfused chain on %d/256 (worst rel diff vs fma chain %.2e)\n",
           layout == 0 ? "row=4*(l>>4)+r" : "row=(l>>4)+4*r", ok32, ok64f, ok64u, worst64);
  }
  return rc;
}
