// Synthetic snapshot matrices for the benchmark:
//   X[i,j] = (f+1) * ( sum_k L[i,k] R[k,j] + eps * N[i,j] ) + 10 f,    f = feature of global row i
// L (n x k) and N (n x m) are standard normal, produced by a counter-based generator keyed
// by (seed, global row, column, stream), so a rank can generate exactly its rows.  R (k x
// ncols_total) carries the designed spectrum and is supplied by the caller.
//
// One wave produces SY_ROWS rows at a time: lanes first draw the rows' L entries (k <= 256,
// four per lane), then walk the columns 64 at a time, broadcasting L[i,kk] with a
// wave-uniform readlane and reading R[kk, col] (L2-resident) once for all SY_ROWS rows.
#include "common.hpp"

namespace {

constexpr int SY_THREADS = 256;
constexpr int SY_ROWS = 4;
constexpr int SY_MAXK = 256;

__device__ inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// standard normal from the counter (seed, row, col, stream)
__device__ inline double normal_at(uint64_t seed, int64_t row, int32_t col, uint32_t stream) {
  const uint64_t key = mix64(seed + 0x9E3779B97F4A7C15ULL * (uint64_t)(row + 1)) ^
                       (0xD1B54A32D192ED03ULL * (uint64_t)((uint32_t)col + 1u)) ^
                       (0x8CB92BA72F3D8DD7ULL * (uint64_t)(stream + 1u));
  const uint64_t h1 = mix64(key), h2 = mix64(key ^ 0xA24BAED4963EE407ULL);
  const double u1 = ((double)(h1 >> 11) + 0.5) * 0x1.0p-53;  // (0,1)
  const double u2 = (double)(h2 >> 11) * 0x1.0p-53;          // [0,1)
  return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}

template <typename TX>
__global__ __launch_bounds__(SY_THREADS) void synth_kernel(
    TX *__restrict__ X, int64_t n_rows, int ncols, int64_t ldx, int64_t row0, int64_t n_points,
    int col0, const double *__restrict__ R, int k, int ldr, double eps, uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * (SY_THREADS / 64) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (SY_THREADS / 64);
  const int64_t ngroups = (n_rows + SY_ROWS - 1) / SY_ROWS;
  for (int64_t g = wave_id; g < ngroups; g += n_waves) {
    double Lr[SY_ROWS][SY_MAXK / 64];
    double fa[SY_ROWS], fb[SY_ROWS];
#pragma unroll
    for (int a = 0; a < SY_ROWS; ++a) {
      const int64_t gi = row0 + g * SY_ROWS + a;
      const int64_t f = gi / n_points;
      fa[a] = (double)(f + 1);
      fb[a] = 10.0 * (double)f;
#pragma unroll
      for (int j = 0; j < SY_MAXK / 64; ++j) {
        const int kk = lane + 64 * j;
        Lr[a][j] = (kk < k) ? normal_at(seed, gi, kk, 0u) : 0.0;
      }
    }
    for (int c0 = 0; c0 < ncols; c0 += 64) {
      const int c = c0 + lane;
      const int cg = col0 + c;
      double acc[SY_ROWS];
#pragma unroll
      for (int a = 0; a < SY_ROWS; ++a) acc[a] = 0.0;
#pragma unroll
      for (int j = 0; j < SY_MAXK / 64; ++j) {
        for (int kk = 0; kk < 64 && 64 * j + kk < k; ++kk) {
          const double rv = (c < ncols) ? R[(int64_t)(64 * j + kk) * ldr + cg] : 0.0;
#pragma unroll
          for (int a = 0; a < SY_ROWS; ++a) acc[a] += __shfl(Lr[a][j], kk, 64) * rv;
        }
      }
      if (c < ncols) {
#pragma unroll
        for (int a = 0; a < SY_ROWS; ++a) {
          const int64_t li = g * SY_ROWS + a;
          if (li < n_rows) {
            const double nz = normal_at(seed, row0 + li, cg, 1u);
            X[li * ldx + c] = (TX)(fa[a] * (acc[a] + eps * nz) + fb[a]);   // f32 storage: the f64 value rounded once
          }
        }
      }
    }
  }
}

__global__ void synth_gather_kernel(const int64_t *__restrict__ rows, int n, int64_t n_points, int col,
                                    const double *__restrict__ R, int k, int ldr, double eps, uint64_t seed,
                                    double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t gi = rows[i];
  const int64_t f = gi / n_points;
  double acc = 0.0;
  for (int kk = 0; kk < k; ++kk) acc += normal_at(seed, gi, kk, 0u) * R[(int64_t)kk * ldr + col];
  out[i] = (double)(f + 1) * (acc + eps * normal_at(seed, gi, col, 1u)) + 10.0 * (double)f;
}

}  // namespace

template <typename TX>
static int synth_entry(const char *who, TX *d_X, int64_t n_rows, int32_t ncols, int64_t ldx, int64_t row0,
                       int64_t n_points, int32_t col0, const double *d_R, int32_t k, int32_t ldr, double eps,
                       uint64_t seed, void *stream) {
  SPR_REQUIRE(d_X && d_R, SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(n_rows > 0 && ncols > 0 && ldx >= ncols && row0 >= 0 && n_points > 0 && col0 >= 0, SPR_E_INVALID,
              "%s: bad shape", who);
  SPR_REQUIRE(k > 0 && k <= SY_MAXK && ldr >= col0 + ncols, SPR_E_INVALID, "%s: bad k=%d ldr=%d", who, k, ldr);
  const int64_t groups = (n_rows + SY_ROWS - 1) / SY_ROWS;
  int64_t blocks = (groups + (SY_THREADS / 64) - 1) / (SY_THREADS / 64);
  const int cus = spr_cached_cus();
  const int64_t cap = 8LL * (cus > 0 ? cus : 256);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(synth_kernel<TX>, dim3((int)blocks), dim3(SY_THREADS), 0, static_cast<hipStream_t>(stream), d_X,
                     n_rows, (int)ncols, ldx, row0, n_points, (int)col0, d_R, (int)k, (int)ldr, eps, seed);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

extern "C" int spr_synth_f64(double *d_X, int64_t n_rows, int32_t ncols, int64_t ldx, int64_t row0,
                             int64_t n_points, int32_t col0, const double *d_R, int32_t k, int32_t ldr, double eps,
                             uint64_t seed, void *stream) {
  return synth_entry("spr_synth_f64", d_X, n_rows, ncols, ldx, row0, n_points, col0, d_R, k, ldr, eps, seed, stream);
}

extern "C" int spr_synth_f32(float *d_X, int64_t n_rows, int32_t ncols, int64_t ldx, int64_t row0,
                             int64_t n_points, int32_t col0, const double *d_R, int32_t k, int32_t ldr, double eps,
                             uint64_t seed, void *stream) {
  return synth_entry("spr_synth_f32", d_X, n_rows, ncols, ldx, row0, n_points, col0, d_R, k, ldr, eps, seed, stream);
}

extern "C" int spr_synth_gather_f64(const int64_t *d_rows, int32_t n, int64_t n_points, int32_t col,
                                    const double *d_R, int32_t k, int32_t ldr, double eps, uint64_t seed,
                                    double *d_out, void *stream) {
  SPR_REQUIRE(d_rows && d_R && d_out, SPR_E_INVALID, "spr_synth_gather_f64: NULL pointer");
  SPR_REQUIRE(n > 0 && n_points > 0 && col >= 0 && k > 0 && k <= SY_MAXK && ldr > col, SPR_E_INVALID,
              "spr_synth_gather_f64: bad shape");
  hipLaunchKernelGGL(synth_gather_kernel, dim3((n + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream),
                     d_rows, (int)n, n_points, (int)col, d_R, (int)k, (int)ldr, eps, seed, d_out);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}
