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
// the product-form epilogue as a lane of lik_batched_f64_kernel runs it: groups of 32 elements
__global__ void kgroup(const double *x, double *f, int ngroups)
{
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngroups) return;
  mhlik_softplus_acc a;
  mhlik_softplus_acc_init(&a);
  for (int i = 0; i < 32; i++) mhlik_softplus_acc_add(&a, x[(size_t)g * 32 + i]);
  if (mhlik_softplus_acc_suspect(&a)) {
    mhlik_softplus_acc_init(&a);
    for (int i = 0; i < 32; i++) mhlik_softplus_acc_add_safe(&a, x[(size_t)g * 32 + i]);
  }
  f[g] = mhlik_softplus_acc_neg_sum(&a);
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
  // product form: groups of 32 consecutive finite arguments against the 80-bit sum of the 32 terms
  const int ng = n / 32;
  std::vector<double> fg(ng);
  for (int i = 0; i < n; i++)
    if (x[i] > 709.0 || x[i] < -745.0) x[i] = -40.0 + 80.0 * (double)((i * 2654435761u) >> 8) / 16777216.0;
  (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(kgroup, dim3((ng + 255) / 256), dim3(256), 0, 0, dx, df, ng);
  if (hipMemcpy(fg.data(), df, ng * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("group kernel failed\n"); return 2; }
  long double gworst = 0, gsum = 0;
  long gcnt = 0, gbad = 0;
  for (int g = 0; g < ng; g++) {
    long double w = 0;
    for (int i = 0; i < 32; i++) { long double xl = x[(size_t)g * 32 + i]; w -= fmaxl(xl, 0) + log1pl(expl(-fabsl(xl))); }
    // Every factor 1 + y is rounded to nearest, exactly as in the reference's own log(1 + exp(x)): 1.1e-16 ABSOLUTE per
    // element.  A group whose 32 terms are all tiny is therefore held to an absolute bar; the relative one applies
    // where the sum is of order 1e-3 or more (a tile's partial is of order 1e2).
    if (fabsl((long double)fg[g] - w) > 2e-15L * fabsl(w) + 4e-15L) gbad++;
    if (fabsl(w) < 1.0L) continue; // relative statistics where the sum is of order one or more
    long double e = ((long double)fg[g] - w) / fabsl(w);
    if (fabsl(e) > gworst) gworst = fabsl(e);
    gsum += e;
    gcnt++;
  }
  printf("product form: max relative error of a 32-element group sum %.3Lg, mean signed %.3Lg over %ld groups of |sum| >= 1, %ld groups outside 2e-15 |sum| + 4e-15\n",
         gworst, gsum / gcnt, gcnt, gbad);
  return (worst < 6e-16L && bad == 0 && gworst < 2e-15L && fabsl(gsum / gcnt) < 5e-17L && gbad == 0) ? 0 : 1;
}
