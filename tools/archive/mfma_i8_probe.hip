// Probe for the "Ozaki" question of DESIGN.md 7: what the i8 matrix pipe of gfx950 delivers next to the vector work an int8-slice
// scheme of the f64 Gram matrix would need.
//   (a) issue rate of v_mfma_i32_16x16x64_i8 and v_mfma_i32_32x32x32_i8 (cycles per instruction and SIMD, TOP/s, held clock);
//   (b) co-execution: 512-thread workgroups = 2 waves per SIMD; waves 0-3 issue i8 MFMAs, waves 4-7 one kind of vector work --
//       f64 fma (the rescale-and-accumulate of the i32 partial sums), v_cvt_f64_i32 + f64 fma (the same with the conversion),
//       32-bit integer shifts / ands (the slicing of mantissas).  Times: MFMA only, VALU only, both.  both ~ max => the pipes
//       overlap; both ~ sum => they share issue (what tools/coexec_probe.hip found for f64 MFMA + any VALU);
//   (c) the same with the f64 MFMA in waves 0-3, for reference.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_i8_probe.hip -o /tmp/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// MK: 0 = i8 16x16x64, 1 = i8 32x32x32, 2 = f64 16x16x4.   VK: 0 = f64 fma, 1 = cvt_f64_i32 + f64 fma, 2 = int shift/and
template <int MODE, int MK, int VK>
__global__ __launch_bounds__(512) void k(double *out, int iters) {
  const int wave = threadIdx.x >> 6;
  double r = 0;
  if (wave < 4) {
    if (MODE & 1) {
      if (MK == 0) {
        i32x4 a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)threadIdx.x};
        i32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (i32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
      } else if (MK == 1) {
        i32x4 a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)threadIdx.x};
        i32x16 acc[2];
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 16; ++j) r += acc[i][j];
      } else {
        f64x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (f64x4){0, 0, 0, 0};
        double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
        for (int it = 0; it < iters; ++it)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
      }
    }
  } else if (MODE & 2) {
    if (VK == 0) {
      double x0 = threadIdx.x, x1 = 1.0000001, x2 = 0.5, x3 = 3;
      for (int it = 0; it < iters * 4; ++it) { x0 = fma(x0, x1, x2); x1 = fma(x1, x2, x3); x2 = fma(x2, x3, x0); x3 = fma(x3, x0, x1); }
      r = x0 + x1 + x2 + x3;
    } else if (VK == 1) {
      int q0 = threadIdx.x, q1 = 7, q2 = 9, q3 = 11;
      double x0 = 0, x1 = 0, x2 = 0, x3 = 0;
      for (int it = 0; it < iters * 2; ++it) {
        x0 = fma((double)q0, 1.5, x0); x1 = fma((double)q1, 1.5, x1); x2 = fma((double)q2, 1.5, x2); x3 = fma((double)q3, 1.5, x3);
        q0 += it; q1 ^= q0; q2 += q1; q3 ^= q2;
      }
      r = x0 + x1 + x2 + x3;
    } else {
      unsigned q0 = threadIdx.x * 2654435761u, q1 = 7, q2 = 9, q3 = 11;
      for (int it = 0; it < iters * 8; ++it) {
        q0 = (q0 >> 7) ^ (q1 & 0x7f7f7f7fu); q1 = (q1 << 3) + (q2 & 0xff00ffu); q2 = (q2 >> 5) ^ (q3 & 0x0f0f0f0fu); q3 = (q3 << 1) + (q0 & 0x3fu);
      }
      r = (double)(q0 + q1 + q2 + q3);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, int MK, int VK> float run(double *out, int cus, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, MK, VK>), cus, 512, 0, 0, out, 10);
  hipEventRecord(e0); hipLaunchKernelGGL((k<MODE, MK, VK>), cus, 512, 0, 0, out, iters); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

template <int MK, int VK> void triple(double *out, int cus, int iters, const char *mk, const char *vk) {
  const float a = run<1, MK, VK>(out, cus, iters), b = run<2, MK, VK>(out, cus, iters), c = run<3, MK, VK>(out, cus, iters);
  printf("  %-22s + %-26s: mfma-only %.3f ms, valu-only %.3f ms, both %.3f ms  (sum %.3f, max %.3f) -> %s\n", mk, vk, a, b, c,
         a + b, a > b ? a : b, c < 0.5f * ((a + b) + (a > b ? a : b)) ? "overlap" : "serialised");
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double *out; hipMalloc(&out, (size_t)cus * 512 * 8);
  const int iters = 40000;
  {
    const float ms = run<1, 0, 0>(out, cus, iters);     // 4 waves x 4 MFMAs x iters per CU, one wave per SIMD
    const double ops = (double)cus * 4 * 4 * iters * (2.0 * 16 * 16 * 64);
    printf("v_mfma_i32_16x16x64_i8: %.2f POP/s, %.1f cycles per instruction and SIMD at the nominal %d MHz\n", ops / ms / 1e12,
           (double)ms * 1e-3 * p.clockRate * 1e3 / (iters * 4.0), p.clockRate / 1000);
  }
  {
    const float ms = run<1, 1, 0>(out, cus, iters);
    const double ops = (double)cus * 4 * 2 * iters * (2.0 * 32 * 32 * 32);
    printf("v_mfma_i32_32x32x32_i8: %.2f POP/s, %.1f cycles per instruction and SIMD at the nominal %d MHz\n", ops / ms / 1e12,
           (double)ms * 1e-3 * p.clockRate * 1e3 / (iters * 2.0), p.clockRate / 1000);
  }
  {
    const float ms = run<1, 2, 0>(out, cus, iters / 4);
    const double fl = (double)cus * 4 * 4 * (iters / 4) * 2048.0;
    printf("v_mfma_f64_16x16x4_f64: %.2f TFLOP/s, %.1f cycles per instruction and SIMD\n", fl / ms / 1e9,
           (double)ms * 1e-3 * p.clockRate * 1e3 / (iters / 4 * 4.0));
  }
  printf("co-execution, waves 0-3 matrix pipe, waves 4-7 vector work (2 waves per SIMD):\n");
  triple<0, 0>(out, cus, iters, "i8 16x16x64", "f64 fma");
  triple<0, 1>(out, cus, iters, "i8 16x16x64", "cvt_f64_i32 + f64 fma");
  triple<0, 2>(out, cus, iters, "i8 16x16x64", "int shift / and");
  triple<1, 0>(out, cus, iters, "i8 32x32x32", "f64 fma");
  triple<1, 2>(out, cus, iters, "i8 32x32x32", "int shift / and");
  triple<2, 0>(out, cus, iters / 4, "f64 16x16x4", "f64 fma");
  triple<2, 2>(out, cus, iters / 4, "f64 16x16x4", "int shift / and");
  return 0;
}
