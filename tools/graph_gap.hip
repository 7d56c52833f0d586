// tools/graph_gap.hip -- what a HIP graph would buy the device-resident MH loop at small N: two dependent
// dispatches per iteration (a "tile" kernel over a few workgroups, a one-workgroup "step" kernel that reads its
// result and writes the next input), 2000 iterations, (a) launched one by one on a stream, the host far ahead,
// (b) captured once as a graph of 100 iterations and launched 20 times.  Prints microseconds per iteration.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void tile_kernel(const double *x, int n, const double *theta, double *partials)
{
  __shared__ double s[256];
  double acc = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc += x[i] * theta[0];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = s[0];
}
__global__ void step_kernel(const double *partials, int nb, double *theta)
{
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < nb; i++) t += partials[i];
    theta[0] = 1.0 + 1e-9 * t;
  }
}

int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 30000, nb = argc > 2 ? atoi(argv[2]) : 20, iters = 2000, per_graph = 100;
  double *x, *theta, *partials;
  CK(hipMalloc(&x, sizeof(double) * n)); CK(hipMemset(x, 0, sizeof(double) * n));
  CK(hipMalloc(&theta, 8)); CK(hipMemset(theta, 0, 8));
  CK(hipMalloc(&partials, 8 * nb));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto one = [&]() {
    hipLaunchKernelGGL(tile_kernel, dim3(nb), dim3(256), 0, st, x, n, theta, partials);
    hipLaunchKernelGGL(step_kernel, dim3(1), dim3(64), 0, st, partials, nb, theta);
  };
  for (int rep = 0; rep < 3; rep++) {
    for (int i = 0; i < 200; i++) one();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < iters; i++) one();
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("stream launches : %.2f us per iteration (n=%d, %d workgroups)\n", ms * 1e3 / iters, n, nb);
  }
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < per_graph; i++) one();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipGraphLaunch(ge, st)); CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < iters / per_graph; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("graph of %d     : %.2f us per iteration\n", per_graph, ms * 1e3 / iters);
  }
  return 0;
}
