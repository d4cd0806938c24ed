#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}
int main() {
  double A[64], B[64], D[64];
  for (int i = 0; i < 64; ++i) { A[i] = 1.0 + 0.37 * i + 0.011 * i * i; B[i] = 2.0 - 0.23 * i + 0.007 * i * i; }
  double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
  hipMemcpy(dA, A, 512, hipMemcpyHostToDevice); hipMemcpy(dB, B, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, 1, 64, 0, 0, dA, dB, dD);
  hipMemcpy(D, dD, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("%.17g\n", D[i]);
  return 0;
}
