// Probe: lane maps and issue rate of v_mfma_f64_16x16x4_f64 on the box (run once, keep the numbers in DESIGN.md).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double* A, const double* B, double* D) {  // A 16x4, B 4x16 row-major
  const int l = threadIdx.x;
  f64x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[((l >> 4) + 4 * i) * 16 + (l & 15)] = acc[i];
}

template <int NACC>
__global__ void rate_kernel(double* out, int iters) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  std::vector<double> A(64), B(64), D(256), R(256);
  for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.5; B[i] = 3 - i * 0.25; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
  double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout_kernel, 1, 64, 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - R[i]));
  printf("layout max err %g (0 => A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4*reg col=l&15)\n", err);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount; double* out; hipMalloc(&out, (size_t)cus * 8 * 1024 * 8);
  for (int wpc : {4, 8}) {
    int iters = 20000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<8>, cus, wpc * 64, 0, 0, out, 100);
    hipEventRecord(e0); hipLaunchKernelGGL(rate_kernel<8>, cus, wpc * 64, 0, 0, out, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)cus * wpc * iters * 8 * 2048.0;
    printf("f64 mfma 16x16x4: %d CUs x %d waves: %.2f TFLOP/s (%.1f cycles/instr/SIMD at %d MHz)\n", cus, wpc,
           flops / ms / 1e9, (double)ms * 1e-3 * p.clockRate * 1e3 / (iters * 8.0 * wpc / 4.0), p.clockRate / 1000);
  }
  return 0;
}
