// Lab harness (GPU box): the W-stationary projection kernel (csrc/project_ws.hip) against the general one
// (spr_project_f64 of the shipped library) on random data -- same inputs, max difference, HIP-event times.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Iopenmeasure_amd/csrc -DPROJ_WS_LAB tools/archive/lab/proj_lab.hip \
//         openmeasure_amd/libspr_hip.so -Wl,-rpath,$PWD/openmeasure_amd -o build/proj_lab
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "project_ws.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(double *p, int64_t n, uint64_t seed, double scale, double shift) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    p[i] = ((double)(z >> 11) / 9007199254740992.0 - 0.5) * scale + shift;
  }
}

int main(int argc, char **argv) {
  const int64_t cells = argc > 1 ? atoll(argv[1]) : 1000000;
  const int F = 9, m = argc > 2 ? atoi(argv[2]) : 256, r = argc > 3 ? atoi(argv[3]) : 64;
  const int reps = 5;
  const int64_t n = cells * F;
  double *X, *mean, *W, *isc, *U0, *U1;
  CK(hipMalloc(&X, sizeof(double) * n * m)); CK(hipMalloc(&mean, sizeof(double) * n));
  CK(hipMalloc(&W, sizeof(double) * m * r)); CK(hipMalloc(&isc, sizeof(double) * F));
  CK(hipMalloc(&U0, sizeof(double) * n * r)); CK(hipMalloc(&U1, sizeof(double) * n * r));
  fill_kernel<<<2048, 256>>>(X, n * m, 1, 2.0, 3.0);
  fill_kernel<<<2048, 256>>>(mean, n, 2, 0.1, 3.0);
  fill_kernel<<<64, 256>>>(W, (int64_t)m * r, 3, 1.0, 0.0);
  fill_kernel<<<1, 64>>>(isc, F, 4, 0.5, 1.0);
  CK(hipMemset(U0, 0, sizeof(double) * n * r)); CK(hipMemset(U1, 0, sizeof(double) * n * r));
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto fn) {
    fn(); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); ts.push_back(t); }
    std::sort(ts.begin(), ts.end());
    return ts[reps / 2];
  };
  const float t0 = time([&] { if (spr_project_f64(X, n, m, m, 0, cells, F, 1, isc, mean, W, r, U0, r, 0, nullptr)) { printf("base: %s\n", spr_last_error()); exit(2); } });
  const float t1 = time([&] { if (spr_project_ws<double, double>(X, n, m, m, 0, cells, F, 1, isc, mean, W, r, U1, r, 0, nullptr)) { printf("ws: unsupported\n"); exit(3); } });
  // compare
  std::vector<double> h0((size_t)1 << 20), h1((size_t)1 << 20);
  double worst = 0.0, mag = 0.0;
  for (int64_t off : {(int64_t)0, n * r / 2 - (1 << 19), n * r - (1 << 20)}) {
    off = off < 0 ? 0 : off;
    const size_t cnt = std::min<int64_t>(1 << 20, n * r - off);
    CK(hipMemcpy(h0.data(), U0 + off, cnt * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), U1 + off, cnt * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < cnt; ++i) { worst = std::max(worst, std::fabs(h0[i] - h1[i])); mag = std::max(mag, std::fabs(h0[i])); }
  }
  const double gb = (double)n * m * 8 / 1e9, fl = 2.0 * n * m * r / 1e12;
  printf("n=%lld m=%d r=%d X=%.2f GB | general %.3f ms (%.1f TF, %.2f TB/s) | W-stationary %.3f ms (%.1f TF, %.2f TB/s) | max|diff| %.3e of %.3e\n",
         (long long)n, m, r, gb, t0, fl / t0 * 1e3, gb / t0, t1, fl / t1 * 1e3, gb / t1, worst, mag);
  return worst <= 1e-10 * mag ? 0 : 4;
}
