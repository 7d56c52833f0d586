// check_f64_overlap.hip -- do v_mfma_f64_16x16x4_f64 of one wave and FP64 VALU work of another wave on the same
// SIMD overlap on gfx950?  (development check behind DESIGN.md section 3.4's statement about the f64 epilogue)
// Workgroup of 8 waves: waves 0-3 run MFMAs, waves 4-7 run v_fma_f64 chains (wave w and w+4 share a SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND> // 0: f64 MFMA + f64 FMA, 1: f32 16x16x4 MFMA + f64 FMA, 2: f64 MFMA + f32 FMA
__global__ __launch_bounds__(512) void k(int do_mfma, int do_valu, int iters, double *out)
{
  const int wave = threadIdx.x >> 6;
  double r = 0.0;
  if (wave < 4) {
    if (do_mfma) {
      if (KIND == 1) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
        for (int i = 0; i < iters; i++) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
      } else {
        f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const double a = 1.0 + threadIdx.x * 1e-3, b = 0.5;
        for (int i = 0; i < iters; i++) {
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
      }
    }
  } else if (do_valu) {
    if (KIND == 2) {
      float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
      const float m = 0.999999f, q = 1e-7f;
      for (int i = 0; i < iters * 8; i++) {
        x0 = __builtin_fmaf(x0, m, q); x1 = __builtin_fmaf(x1, m, q); x2 = __builtin_fmaf(x2, m, q); x3 = __builtin_fmaf(x3, m, q);
        x4 = __builtin_fmaf(x4, m, q); x5 = __builtin_fmaf(x5, m, q); x6 = __builtin_fmaf(x6, m, q); x7 = __builtin_fmaf(x7, m, q);
      }
      r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else {
      double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
      const double m = 0.999999, q = 1e-7;
      for (int i = 0; i < iters * 8; i++) {
        x0 = __builtin_fma(x0, m, q); x1 = __builtin_fma(x1, m, q); x2 = __builtin_fma(x2, m, q); x3 = __builtin_fma(x3, m, q);
        x4 = __builtin_fma(x4, m, q); x5 = __builtin_fma(x5, m, q); x6 = __builtin_fma(x6, m, q); x7 = __builtin_fma(x7, m, q);
      }
      r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
  }
  if (r == 12345.678) out[0] = r;
}

// the same mix inside every wave: 4 f64 MFMAs and 32 f32 (KIND 2) or f64 FMAs per iteration, independent of each other
template <int KIND> __global__ __launch_bounds__(256) void k_same(int do_mfma, int do_valu, int iters, double *out)
{
  f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double a = 1.0 + threadIdx.x * 1e-3, b = 0.5;
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float y0 = (float)x0, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3, y4 = y0 + 4, y5 = y0 + 5, y6 = y0 + 6, y7 = y0 + 7;
  const double m = 0.999999, q = 1e-7;
  const float mf = 0.999999f, qf = 1e-7f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (do_mfma) {
        if (u == 0) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        if (u == 1) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        if (u == 2) c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        if (u == 3) c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      }
      if (do_valu) {
        if (KIND == 2) {
          y0 = __builtin_fmaf(y0, mf, qf); y1 = __builtin_fmaf(y1, mf, qf); y2 = __builtin_fmaf(y2, mf, qf); y3 = __builtin_fmaf(y3, mf, qf);
          y4 = __builtin_fmaf(y4, mf, qf); y5 = __builtin_fmaf(y5, mf, qf); y6 = __builtin_fmaf(y6, mf, qf); y7 = __builtin_fmaf(y7, mf, qf);
        } else {
          x0 = __builtin_fma(x0, m, q); x1 = __builtin_fma(x1, m, q); x2 = __builtin_fma(x2, m, q); x3 = __builtin_fma(x3, m, q);
          x4 = __builtin_fma(x4, m, q); x5 = __builtin_fma(x5, m, q); x6 = __builtin_fma(x6, m, q); x7 = __builtin_fma(x7, m, q);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const double r = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
  if (r == 12345.678) out[0] = r;
}

template <int KIND> void run_same(const char *label, double *out, int block)
{
  const int iters = 4000;
  float t[3];
  for (int mode = 0; mode < 3; mode++) {
    const int dm = mode != 1, dv = mode != 0;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_same<KIND><<<256, block>>>(dm, dv, iters, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k_same<KIND><<<256, block>>>(dm, dv, iters, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&t[mode], a, b));
  }
  printf("%-28s MFMA only %.3f ms, VALU only %.3f ms, both in one wave %.3f ms  (sum %.3f, max %.3f)\n", label, t[0], t[1], t[2],
         t[0] + t[1], t[0] > t[1] ? t[0] : t[1]);
}

template <int KIND> void run(const char *label, double *out)
{
  const int iters = 4000;
  float t[3];
  for (int mode = 0; mode < 3; mode++) {
    const int dm = mode != 1, dv = mode != 0;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<KIND><<<256, 512>>>(dm, dv, iters, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k<KIND><<<256, 512>>>(dm, dv, iters, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&t[mode], a, b));
  }
  printf("%-28s MFMA waves alone %.3f ms, VALU waves alone %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", label, t[0], t[1], t[2],
         t[0] + t[1], t[0] > t[1] ? t[0] : t[1]);
}

__global__ void where(unsigned *ids) // HW_ID of every wave of workgroup 0: simd_id = bits 5:4, cu_id = bits 11:8
{
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) ids[threadIdx.x >> 6] = id;
}

int main()
{
  double *out;
  CK(hipMalloc(&out, 64));
  unsigned *ids, h[8];
  CK(hipMalloc(&ids, 64));
  where<<<256, 512>>>(ids);
  CK(hipMemcpy(h, ids, 32, hipMemcpyDeviceToHost));
  printf("workgroup 0 of a 512-thread launch: wave -> (simd, cu):");
  for (int w = 0; w < 8; w++) printf(" %d->(%u,%u)", w, (h[w] >> 4) & 3, (h[w] >> 8) & 15);
  printf("\n");
  run<0>("f64 MFMA | f64 FMA", out);
  run<1>("f32 16x16x4 MFMA | f64 FMA", out);
  run<2>("f64 MFMA | f32 FMA", out);
  run_same<0>("same wave, 1/SIMD: f64 MFMA + f64 FMA", out, 256);
  run_same<2>("same wave, 1/SIMD: f64 MFMA + f32 FMA", out, 256);
  return 0;
}
