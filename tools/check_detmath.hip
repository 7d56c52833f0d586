// tools/check_detmath.hip -- are mhlik_expf_det / mhlik_sigmoidf_det bit-identical on host and device?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "mhlik_detmath.h"
__global__ void k(const float *x, float *e, float *s, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { e[i] = mhlik_expf_det(x[i]); s[i] = mhlik_sigmoidf_det(x[i]); }
}
int main()
{
  const int n = 1 << 20;
  std::vector<float> x(n), e(n), s(n);
  for (int i = 0; i < n; i++) x[i] = -30.0f + 60.0f * (float)i / n;
  float *dx, *de, *ds;
  (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&de, n * 4); (void)hipMalloc(&ds, n * 4);
  (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, de, ds, n);
  (void)hipMemcpy(e.data(), de, n * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
  long be = 0, bs = 0;
  for (int i = 0; i < n; i++) {
    float he = mhlik_expf_det(x[i]), hs = mhlik_sigmoidf_det(x[i]);
    if (memcmp(&he, &e[i], 4)) { if (be < 5) printf("exp x=%a host=%a dev=%a\n", x[i], he, e[i]); be++; }
    if (memcmp(&hs, &s[i], 4)) { if (bs < 5) printf("sig x=%a host=%a dev=%a\n", x[i], hs, s[i]); bs++; }
  }
  printf("exp mismatches %ld, sigmoid mismatches %ld of %d\n", be, bs, n);
  return 0;
}
