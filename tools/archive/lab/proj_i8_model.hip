// THROUGHPUT MODEL (not a correct kernel) of an int8-slice projection  Ur = X W  at the config-3 shape (m = 256, r = 64), to see
// whether the idea in DESIGN.md 7 can beat the f64 W-stationary kernel (52.5 ms per 90M rows) before anybody builds it for real.
// What is real: the HBM traffic (every row of X read once in the A-operand layout of v_mfma_i32_16x16x64_i8 -- 128 contiguous
// bytes per lane and k step --, 64 doubles written per row), the instruction mix of the slicing (row exponent, 64-bit alignment
// shift, eight sign-magnitude 7-bit digits per element, packing into bytes), the 36 slice pairs x 4 column tiles x 4 k steps of
// MFMAs per 16-row block with the B digits read from a 128 KB LDS image, and the combination of the eight digit groups in f64.
// What is not: the LDS image holds arbitrary bytes, the row scales in the epilogue come from the wrong lanes, nothing is centred.
// One wave per SIMD (the register budget: 128 row data + 32 digits + 128 accumulators + 16 B), 16-row blocks dealt round-robin.
// Build: hipcc -O3 --offload-arch=gfx950 tools/archive/lab/proj_i8_model.hip -o build/probes/proj_i8_model ; run: proj_i8_model [rows]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

#ifndef PAIRS_MAX
#define PAIRS_MAX 7          // slice pairs with s + t <= PAIRS_MAX (7: 36 pairs of 8 slices)
#endif
#ifndef MODEL_PIPE
#define MODEL_PIPE 0       // n > 0: slicing of the next k step interleaved with the MFMAs, n vector instructions per MFMA
#endif
#ifndef MODEL_ABLATE
#define MODEL_ABLATE 0       // 1: no MFMAs, 2: no slicing (digits = raw words), 3: neither (loads + stores only)
#endif
constexpr int M = 256, R = 64, WAVES = 4, NS = 8;

__device__ inline unsigned pack4(int d0, int d1, int d2, int d3) {
  return (unsigned)(d0 & 0xff) | ((unsigned)(d1 & 0xff) << 8) | ((unsigned)(d2 & 0xff) << 16) | ((unsigned)d3 << 24);
}

__global__ __launch_bounds__(WAVES * 64) void proj_i8_model(const double *__restrict__ X, int64_t n_rows,
                                                            const unsigned char *__restrict__ Wimg, double *__restrict__ U) {
  __shared__ i32x4 Wl[NS * 4 * 4 * 64];                      // [slice t][k step][column tile][lane] : 128 KB
  for (int e = threadIdx.x; e < NS * 4 * 4 * 64; e += WAVES * 64) Wl[e] = reinterpret_cast<const i32x4 *>(Wimg)[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kq = lane >> 4;
  const int64_t nblocks = n_rows / 16;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t b = (int64_t)blockIdx.x * WAVES + wave;
  if (b >= nblocks) return;
  f64x2 x[4][8];                                              // the lane's 64 values of its row: k step ks, pairs of doubles
  {
    const double *rp = X + (b * 16 + li) * M + 16 * kq;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int q = 0; q < 8; ++q) x[ks][q] = *reinterpret_cast<const f64x2 *>(rp + 64 * ks + 2 * q);
  }
  while (b < nblocks) {
    const int64_t bn = b + stride;
    const double *rpn = X + ((bn < nblocks ? bn : b) * 16 + li) * M + 16 * kq;
    // row exponent: the largest biased exponent among the row's 256 values (64 here, the rest in lanes li + 16, 32, 48)
    unsigned emax = 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned h0 = (unsigned)(__double_as_longlong(x[ks][q].x) >> 32), h1 = (unsigned)(__double_as_longlong(x[ks][q].y) >> 32);
        const unsigned e0 = (h0 >> 20) & 0x7ff, e1 = (h1 >> 20) & 0x7ff;
        emax = emax > e0 ? emax : e0;
        emax = emax > e1 ? emax : e1;
      }
    { unsigned o = __shfl_xor(emax, 16, 64); emax = emax > o ? emax : o; o = __shfl_xor(emax, 32, 64); emax = emax > o ? emax : o; }
    i32x4 acc[NS][4];
#pragma unroll
    for (int d = 0; d < NS; ++d)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[d][ct] = (i32x4){0, 0, 0, 0};
    auto slice = [&](int ks, unsigned (&dig)[NS][4]) {        // 16 values -> eight planes of 16 signed 7-bit digits
#pragma unroll
      for (int g = 0; g < 4; ++g) {                           // four values per packed dword
        int d[4][NS];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const double xv = (v & 1) ? x[ks][2 * g + (v >> 1)].y : x[ks][2 * g + (v >> 1)].x;
          const unsigned long long bits = (unsigned long long)__double_as_longlong(xv);
#if MODEL_ABLATE & 2
#pragma unroll
          for (int s = 0; s < NS; ++s) d[v][s] = (int)(bits >> (7 * s)) & 127;
#else
          const unsigned hi = (unsigned)(bits >> 32);
          const unsigned ex = (hi >> 20) & 0x7ff;
          unsigned long long mant = (bits & 0x000fffffffffffffull) | (ex ? 0x0010000000000000ull : 0ull);
          const unsigned delta = emax - ex;
          unsigned long long T = delta < 56 ? ((mant << 3) >> delta) : 0ull;      // 56-bit magnitude aligned to the row exponent
          const int sg = (int)hi >> 31;                                             // 0 or -1
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const int mag = (int)((T >> (7 * (7 - s))) & 127ull);
            d[v][s] = (mag ^ sg) - sg;
          }
#endif
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) dig[s][g] = pack4(d[0][s], d[1][s], d[2][s], d[3][s]);
      }
    };
    auto mfmas = [&](int ks, const unsigned (&dig)[NS][4]) {  // slice pairs (s, t), s + t <= PAIRS_MAX, four column tiles
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        i32x4 bf[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) bf[ct] = Wl[((t * 4 + ks) * 4 + ct) * 64 + lane];
#pragma unroll
        for (int s = 0; s + t <= PAIRS_MAX && s < NS; ++s) {
          const i32x4 af = {(int)dig[s][0], (int)dig[s][1], (int)dig[s][2], (int)dig[s][3]};
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
#if MODEL_ABLATE & 1
            asm volatile("" ::"v"(af), "v"(bf[ct]));
#else
            acc[s + t][ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf[ct], acc[s + t][ct], 0, 0, 0);
#endif
          }
        }
      }
    };
#if MODEL_PIPE
    // software pipeline: the digits of k step ks + 1 are cut while the MFMAs of step ks run (one instruction stream: the
    // scheduler is asked for {1 MFMA, MODEL_PIPE vector instructions} groups)
    unsigned digA[NS][4], digB[NS][4];
    slice(0, digA);
#pragma unroll
    for (int q = 0; q < 8; ++q) x[0][q] = *reinterpret_cast<const f64x2 *>(rpn + 2 * q);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      unsigned (&cur)[NS][4] = (ks & 1) ? digB : digA;
      unsigned (&nxt)[NS][4] = (ks & 1) ? digA : digB;
      if (ks < 3) slice(ks + 1, nxt);
      mfmas(ks, cur);
      if (ks < 3) {
#pragma unroll
        for (int q = 0; q < 8; ++q) x[ks + 1][q] = *reinterpret_cast<const f64x2 *>(rpn + 64 * (ks + 1) + 2 * q);
      }
#pragma unroll
      for (int rep = 0; rep < 144; ++rep) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, MODEL_PIPE, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      unsigned dig[NS][4];
      slice(ks, dig);
      __builtin_amdgcn_sched_barrier(0);
      // the registers of this k step are free: request the next block's piece
#pragma unroll
      for (int q = 0; q < 8; ++q) x[ks][q] = *reinterpret_cast<const f64x2 *>(rpn + 64 * ks + 2 * q);
      mfmas(ks, dig);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    // ---- combination of the digit groups in f64, row and column scales, store (result tile: row 4 (l >> 4) + reg, col l & 15)
    const double rscale = __longlong_as_double((long long)(emax ? emax : 1) << 52) * 0x1p-55;
    double *up = U + (b * 16 + 4 * kq) * R + li;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        double sacc = 0.0;
#pragma unroll
        for (int d = NS - 1; d >= 0; --d) sacc = fma((double)acc[d][ct][reg], 1.0, sacc * 0x1p-7);
        up[reg * R + 16 * ct] = sacc * rscale;
      }
    b = bn;
  }
}

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 11250000;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  double *X, *U; unsigned char *W;
  if (hipMalloc(&X, (size_t)n * M * 8) != hipSuccess || hipMalloc(&U, (size_t)n * R * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&W, NS * 4 * 4 * 64 * 16);
  hipMemset(W, 0x15, NS * 4 * 4 * 64 * 16);
  {  // X: a repeating pattern of finite doubles of mixed magnitude and sign (1 GB generated on the host, tiled)
    const size_t chunk = (size_t)1 << 27;
    double *h = (double *)malloc(chunk * 8);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < chunk; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = ((double)(int64_t)s) * 0x1p-63 * (1.0 + (double)(i & 7)); }
    for (size_t o = 0; o < (size_t)n * M; o += chunk) hipMemcpy(X + o, h, (o + chunk <= (size_t)n * M ? chunk : (size_t)n * M - o) * 8, hipMemcpyHostToDevice);
    free(h);
  }
  const int grid = p.multiProcessorCount;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(proj_i8_model, grid, WAVES * 64, 0, 0, X, n < 160000 ? n : 160000, W, U);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(proj_i8_model, grid, WAVES * 64, 0, 0, X, n, W, U); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
  const double gb = (double)n * (M + R) * 8 / 1e9;
  printf("proj_i8_model (pairs s+t <= %d, ablate %d, pipe %d): %lld rows: %.3f ms = %.2f TB/s of X + Ur  (f64 W-stationary kernel: 52.5 ms per 90M rows, 6.4-6.7 per 11.25M)\n",
         PAIRS_MAX, MODEL_ABLATE, MODEL_PIPE, (long long)n, best, gb / best);
  return 0;
}
