// tools/check_softplus.hip -- accuracy of mhlik_neg_softplus as compiled for
// gfx950 (rcp + Newton quotient, device ldexp / rint) against an 80-bit long
// double evaluation on the host.  Exit status 0 = within 6e-16 relative.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#include "mhlik_softplus.h"
__global__ void k(const double *x, double *f, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f[i] = mhlik_neg_softplus(x[i]);
}
int main()
{
  const int n = 1 << 20;
  std::vector<double> x(n), f(n);
  for (int i = 0; i < n; i++) x[i] = (i & 1) ? -40.0 + 80.0 * (double)i / n : -708.0 + 1417.0 * (double)i / n;
  x[0] = 0.0; x[2] = 710.0; x[4] = -800.0; x[6] = 709.0;
  double *dx, *df;
  if (hipMalloc(&dx, n * 8) != hipSuccess || hipMalloc(&df, n * 8) != hipSuccess) { printf("no device memory\n"); return 2; }
  (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, df, n);
  if (hipMemcpy(f.data(), df, n * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
  long double worst = 0; double at = 0; long bad = 0;
  for (int i = 0; i < n; i++) {
    if (x[i] > 709.782712893384) { if (!(isinf(f[i]) && f[i] < 0)) bad++; continue; }
    if (x[i] < -746) { if (f[i] != 0.0) bad++; continue; }
    long double xl = x[i], w = -(fmaxl(xl, 0) + log1pl(expl(-fabsl(xl))));
    long double e = fabsl((long double)f[i] - w) / fabsl(w);
    if (e > worst) { worst = e; at = x[i]; }
  }
  printf("max relative error %.3Lg at x=%g, %ld special-case mismatches\n", worst, at, bad);
  return (worst < 6e-16L && bad == 0) ? 0 : 1;
}
