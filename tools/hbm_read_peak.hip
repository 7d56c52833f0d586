// tools/hbm_read_peak.hip -- device-read microbenchmark: how fast can this
// GPU stream a buffer with 16-byte loads and no arithmetic to speak of?
// The likelihood kernel's GB/s is also quoted against this number.
//   build: hipcc --offload-arch=gfx950 -O3 tools/hbm_read_peak.hip -o tools/hbm_read_peak
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT> __global__ __launch_bounds__(256) void read_kernel(const f4 *__restrict__ p, size_t n16, float *out)
{
  // each workgroup streams a contiguous chunk of U*256 vectors per step
  size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  f4 acc = {0, 0, 0, 0};
  for (; base + (U - 1) * 256 < n16; base += stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      if constexpr (NT)
        v[u] = __builtin_nontemporal_load(p + base + u * 256);
      else
        v[u] = p[base + u * 256];
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u];
  }
  float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123.456f) out[0] = s; // keep the loads alive
}

template <int U, bool NT> void run(const f4 *buf, size_t bytes, float *out, int grid, const char *name)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> ms;
  for (int r = 0; r < 13; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((read_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, buf, bytes / 16, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float m;
    hipEventElapsedTime(&m, e0, e1);
    if (r >= 3) ms.push_back(m);
  }
  std::sort(ms.begin(), ms.end());
  printf("%-24s grid=%6d  min %.4f ms  med %.4f ms  %.1f GB/s (med)\n", name, grid, ms.front(), ms[ms.size() / 2],
         bytes / ms[ms.size() / 2] * 1e-6);
}

int main(int argc, char **argv)
{
  size_t bytes = (argc > 1 ? atof(argv[1]) : 5.12e9);
  bytes &= ~(size_t)0xFFFFF;
  f4 *buf;
  float *out;
  if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
  hipMalloc(&out, 4);
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  for (int grid : {2048, 4096, 8192, 16384, 65536}) {
    run<4, true>(buf, bytes, out, grid, "U4 nt");
    run<8, true>(buf, bytes, out, grid, "U8 nt");
    run<16, true>(buf, bytes, out, grid, "U16 nt");
    run<8, false>(buf, bytes, out, grid, "U8 plain");
  }
  return 0;
}
