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

// MAP: which chunk a workgroup takes.  Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8):
//   0  chunk b of every window of gridDim.x chunks (neighbouring chunks on different XCDs);
//   1  XCD x takes the x-th eighth of every window (neighbouring chunks on one XCD);
//   2  XCD x streams the x-th eighth of the whole buffer.
template <int U, bool NT, int MAP = 0>
__global__ __launch_bounds__(256) void read_kernel(const f4 *__restrict__ p, size_t n16, float *out)
{
  // each workgroup streams a contiguous chunk of U*256 vectors per step
  const size_t per = gridDim.x / 8;
  const size_t slot = MAP == 0 ? blockIdx.x : (blockIdx.x % 8) * per + blockIdx.x / 8;
  size_t base = slot * (256 * U) + threadIdx.x;
  size_t stride = (size_t)gridDim.x * 256 * U;
  size_t end = n16;
  if (MAP == 2) {
    const size_t eighth = n16 / 8 / (256 * U) * (256 * U);
    base = (blockIdx.x % 8) * eighth + (blockIdx.x / 8) * (256 * U) + threadIdx.x;
    stride = per * 256 * U;
    end = (blockIdx.x % 8 + 1) * eighth;
  }
  f4 acc = {0, 0, 0, 0};
  for (; base + (U - 1) * 256 < end; base += stride) {
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

template <int U, bool NT, int MAP = 0> void run(const f4 *buf, size_t bytes, float *out, int grid, const char *name)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> ms;
  for (int r = 0; r < 13; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((read_kernel<U, NT, MAP>), dim3(grid), dim3(256), 0, 0, buf, bytes / 16, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float m;
    hipEventElapsedTime(&m, e0, e1);
    if (r >= 3) ms.push_back(m);
  }
  std::sort(ms.begin(), ms.end());
  printf("%-36s grid=%6d  min %.4f ms  med %.4f ms  %.1f GB/s (med)\n", name, grid, ms.front(), ms[ms.size() / 2],
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
    run<16, true, 1>(buf, bytes, out, grid, "U16 nt, XCD eighths of a window");
    run<16, true, 2>(buf, bytes, out, grid, "U16 nt, XCD eighths of the buffer");
    run<8, true, 2>(buf, bytes, out, grid, "U8 nt, XCD eighths of the buffer");
  }
  return 0;
}
