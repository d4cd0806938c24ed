// Probe: lane map and issue rate of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction) on the box --
// the candidate for the 16 diagonal tiles of the m = 256 Gram kernel, which v_mfma_f64_16x16x4_f64 computes in full
// although they are symmetric.  Prints the issue cost next to the 16x16x4 form's.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double* A, const double* B, double* D) {  // per block b: A_b 4x4 [i][k], B_b 4x4 [k][j]
  const int l = threadIdx.x, b = l >> 4, x = l & 3, y = (l >> 2) & 3;
  // hypothesis: A operand lane (b, k = y, i = x), B operand lane (b, k = y, j = x), D lane (b, i = y, j = x)
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[b * 16 + x * 4 + y], B[b * 16 + y * 4 + x], 0.0, 0, 0, 0);
  D[l] = d;
}

template <int NACC>
__global__ void rate4_kernel(double* out, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void rate16_kernel(double* out, int iters) {
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
  std::vector<double> A(64), B(64), D(64), R(64);
  for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.5; B[i] = 3 - i * 0.25; }
  for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
    double s = 0; for (int k = 0; k < 4; ++k) s += A[b * 16 + i * 4 + k] * B[b * 16 + k * 4 + j]; R[b * 16 + i * 4 + j] = s; }
  double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout_kernel, 1, 64, 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0;                       // D lane (b, i = y, j = x)  vs  D lane (b, i = x, j = y)
  for (int l = 0; l < 64; ++l) { int b = l >> 4, x = l & 3, y = (l >> 2) & 3;
    e1 = fmax(e1, fabs(D[l] - R[b * 16 + y * 4 + x])); e2 = fmax(e2, fabs(D[l] - R[b * 16 + x * 4 + y])); }
  printf("4x4x4_4b layout: err with D[b][i=y][j=x] %g, with D[b][i=x][j=y] %g  (x = l&3, y = (l>>2)&3, b = l>>4)\n", e1, e2);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount; double* out; hipMalloc(&out, (size_t)cus * 8 * 1024 * 8);
  for (int wpc : {4, 8}) {
    int iters = 20000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    hipLaunchKernelGGL(rate4_kernel<8>, cus, wpc * 64, 0, 0, out, 100);
    hipEventRecord(e0); hipLaunchKernelGGL(rate4_kernel<8>, cus, wpc * 64, 0, 0, out, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("f64 mfma 4x4x4_4b: %d waves/CU: %.2f TFLOP/s, %.1f cycles/instr/SIMD at %d MHz\n", wpc,
           (double)cus * wpc * iters * 8 * 512.0 / ms / 1e9, (double)ms * 1e-3 * p.clockRate * 1e3 / (iters * 8.0 * wpc / 4.0), p.clockRate / 1000);
    hipLaunchKernelGGL(rate16_kernel<8>, cus, wpc * 64, 0, 0, out, 100);
    hipEventRecord(e0); hipLaunchKernelGGL(rate16_kernel<8>, cus, wpc * 64, 0, 0, out, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("f64 mfma 16x16x4 : %d waves/CU: %.2f TFLOP/s, %.1f cycles/instr/SIMD\n", wpc,
           (double)cus * wpc * iters * 8 * 2048.0 / ms / 1e9, (double)ms * 1e-3 * p.clockRate * 1e3 / (iters * 8.0 * wpc / 4.0));
  }
  return 0;
}
