// K7 + K8: Theta = C . Ur, cnt = C . X_cnt (and optionally scl = C . X_scl) for a CSR matrix C (s x n).
//
// One 256-thread workgroup per row of C.  The one-hot C of optimal_placement
// (sparse_sensing.py:741-743) has one entry per row, which makes this a row gather; a
// dense or camera-projection C (docs/ctc_doc.ipynb) is the general SpMM.  Waves take the
// row's entries round-robin, lanes take the r columns of Ur; the four partial rows are
// added in a fixed order through LDS, so the result is reproducible.  Entries whose
// column lies outside this rank's rows [row0, row0+n_rows) are skipped.
#include "common.hpp"

namespace {

constexpr int MS_THREADS = 256;
constexpr int MS_WAVES = MS_THREADS / 64;

template <typename TU>
__global__ __launch_bounds__(MS_THREADS) void measure_csr_kernel(
    const int64_t *__restrict__ indptr, const int64_t *__restrict__ indices, const double *__restrict__ vals,
    const TU *__restrict__ Ur, int64_t n_rows, int r, int64_t ldu, int64_t row0,
    const double *__restrict__ rowmean, const double *__restrict__ scale, int64_t n_points, int n_features,
    double *__restrict__ Theta, int ldt, double *__restrict__ cnt, double *__restrict__ scl) {
  __shared__ double part[MS_WAVES][SPR_MAX_R + 2];
  const int row = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t e0 = indptr[row], e1 = indptr[row + 1];
  double a0 = 0.0, a1 = 0.0, ac = 0.0, as = 0.0;
  for (int64_t e = e0 + wave; e < e1; e += MS_WAVES) {
    const int64_t col = indices[e] - row0;
    if (col < 0 || col >= n_rows) continue;
    const double v = vals[e];
    const TU *u = Ur + col * ldu;
    if (lane < r) a0 += v * (double)u[lane];
    if (lane + 64 < r) a1 += v * (double)u[lane + 64];
    if (lane == 0) {
      ac += v * rowmean[col];
      if (scl) {                                   // X_scl of that row = scale of its feature (:110, :115)
        int64_t f = (row0 + col) / n_points;
        if (f > n_features - 1) f = n_features - 1;
        as += v * scale[f];
      }
    }
  }
  part[wave][lane] = a0;
  part[wave][lane + 64] = a1;
  if (lane == 0) { part[wave][SPR_MAX_R] = ac; part[wave][SPR_MAX_R + 1] = as; }
  __syncthreads();
  for (int k = threadIdx.x; k <= SPR_MAX_R + 1; k += MS_THREADS) {
    double s = 0.0;
    for (int w = 0; w < MS_WAVES; ++w) s += part[w][k];
    if (k < r) Theta[(int64_t)row * ldt + k] = s;
    if (k == SPR_MAX_R) cnt[row] = s;
    if (k == SPR_MAX_R + 1 && scl) scl[row] = s;
  }
}

}  // namespace

template <typename TU>
static int measure_entry(const char *who, const int64_t *d_indptr, const int64_t *d_indices, const double *d_vals,
                         int32_t s, const TU *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                         const double *d_rowmean, const double *d_scale, int64_t n_points, int32_t n_features,
                         double *d_Theta, double *d_cnt, double *d_scl, void *stream) {
  SPR_REQUIRE(d_indptr && d_indices && d_vals && d_Ur && d_rowmean && d_Theta && d_cnt, SPR_E_INVALID,
              "%s: NULL pointer", who);
  SPR_REQUIRE(s > 0 && n_rows > 0 && r > 0 && ldu >= r && row0 >= 0, SPR_E_INVALID,
              "%s: bad shape s=%d n_rows=%lld r=%d", who, s, (long long)n_rows, r);
  SPR_REQUIRE(!d_scl || (d_scale && n_points > 0 && n_features > 0), SPR_E_INVALID,
              "%s: scl output needs the per-feature scale and layout", who);
  // 128 columns of Ur per launch; a wider basis goes in column groups (cnt / scl come out the same every time)
  for (int g0 = 0; g0 < r; g0 += SPR_MAX_R) {
    const int rg = (r - g0 < SPR_MAX_R) ? r - g0 : SPR_MAX_R;
    hipLaunchKernelGGL(measure_csr_kernel<TU>, dim3(s), dim3(MS_THREADS), 0, static_cast<hipStream_t>(stream), d_indptr,
                       d_indices, d_vals, d_Ur + g0, n_rows, rg, ldu, row0, d_rowmean, d_scale, n_points, (int)n_features,
                       d_Theta + g0, (int)r, d_cnt, d_scl);
    SPR_LAUNCH_CHECK();
  }
  return SPR_OK;
}

extern "C" int spr_measure_csr_f64(const int64_t *d_indptr, const int64_t *d_indices, const double *d_vals,
                                   int32_t s, const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu,
                                   int64_t row0, const double *d_rowmean, const double *d_scale, int64_t n_points,
                                   int32_t n_features, double *d_Theta, double *d_cnt, double *d_scl,
                                   void *stream) {
  return measure_entry("spr_measure_csr_f64", d_indptr, d_indices, d_vals, s, d_Ur, n_rows, r, ldu, row0, d_rowmean,
                       d_scale, n_points, n_features, d_Theta, d_cnt, d_scl, stream);
}

extern "C" int spr_measure_csr_u32(const int64_t *d_indptr, const int64_t *d_indices, const double *d_vals,
                                   int32_t s, const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu,
                                   int64_t row0, const double *d_rowmean, const double *d_scale, int64_t n_points,
                                   int32_t n_features, double *d_Theta, double *d_cnt, double *d_scl,
                                   void *stream) {
  return measure_entry("spr_measure_csr_u32", d_indptr, d_indices, d_vals, s, d_Ur, n_rows, r, ldu, row0, d_rowmean,
                       d_scale, n_points, n_features, d_Theta, d_cnt, d_scl, stream);
}
