// Probe: does VALU work (f64 adds / 32-bit DPP movs) overlap with v_mfma_f64_16x16x4_f64 on gfx950?
// 512-thread blocks = 2 waves per SIMD; waves 0-3 run MFMAs, waves 4-7 run VALU.  Times: MFMA only,
// VALU only, both.  both ~ max => separate pipes; both ~ sum => shared.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 1 = mfma waves active, 2 = valu waves active, 3 = both; VK: 0 = f64 add, 1 = f32 fma, 2 = same-wave mix
__global__ __launch_bounds__(512) void k(double* out, int iters, int vk) {
  const int wave = threadIdx.x >> 6;
  double r = 0;
  if (wave < 4) {
    if (MODE & 1) {
      f64x4 acc[4];
      for (int i = 0; i < 4; ++i) acc[i] = (f64x4){0, 0, 0, 0};
      double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else {
    if (MODE & 2) {
      if (vk == 0) {
        double x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
        for (int it = 0; it < iters * 16; ++it) { x0 += x1; x1 += x2; x2 += x3; x3 += x0; }
        r = x0 + x1 + x2 + x3;
      } else {
        float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
        for (int it = 0; it < iters * 32; ++it) { x0 = fmaf(x0, x1, x2); x1 = fmaf(x1, x2, x3); x2 = fmaf(x2, x3, x0); x3 = fmaf(x3, x0, x1); }
        r = x0 + x1 + x2 + x3;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> float run(double* out, int cus, int iters, int vk) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, cus, 512, 0, 0, out, 10, vk);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, cus, 512, 0, 0, out, iters, vk); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount; double* out; hipMalloc(&out, (size_t)cus * 512 * 8);
  for (int vk = 0; vk < 2; ++vk) {
    float a = run<1>(out, cus, 20000, vk), b = run<2>(out, cus, 20000, vk), c = run<3>(out, cus, 20000, vk);
    printf("valu kind %s: mfma-only %.3f ms, valu-only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n",
           vk == 0 ? "f64 add" : "f32 fma", a, b, c, a + b, a > b ? a : b);
  }
  return 0;
}
