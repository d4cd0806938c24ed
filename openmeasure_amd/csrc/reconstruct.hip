// K10 + K11: x = X_scl * (Ur a) + X_cnt  -- one streaming pass over the basis shard.
//
// HBM-bound (2 flops per 8 bytes).  A row of Ur (r doubles) is read by LPR consecutive
// lanes as 16-byte pieces (LPR = smallest power of two >= r/2), so a wave instruction
// covers 64/LPR whole rows of contiguous memory; each lane keeps UNR row groups in flight.
// The dot product is closed with a butterfly over the LPR lanes and the un-scaling
// (sparse_sensing.py:235) is applied before the store.  Workgroups are dealt to feature
// segments (common.hpp) so the per-feature scale is a workgroup constant.
#include "common.hpp"

namespace {

constexpr int RC_THREADS = 256;
constexpr int RC_UNR = 4;
constexpr int RC_PB = 4;  // coefficient vectors handled per pass over Ur

template <int LPR>
__global__ __launch_bounds__(RC_THREADS) void reconstruct_kernel(
    const double *__restrict__ Ur, int r, int64_t ldu, int vec_ok_i, SegPlan plan,
    const double *__restrict__ rowmean, const double *__restrict__ scale, const double *__restrict__ rowscale,
    const double *__restrict__ A, int np0, int npb, double *__restrict__ out, int64_t ldo) {
  constexpr int RPW = 64 / LPR;                        // rows per wave instruction
  constexpr int ROWS_IT = (RC_THREADS / 64) * RPW * RC_UNR;  // rows per workgroup step
  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / LPR, lig = lane % LPR;
  const bool vec_ok = vec_ok_i != 0;
  const double sc = scale[f];
  const int k0 = 2 * lig;

  double a0[RC_PB], a1[RC_PB];
#pragma unroll
  for (int p = 0; p < RC_PB; ++p) {
    a0[p] = (p < npb && k0 < r) ? A[(int64_t)(np0 + p) * r + k0] : 0.0;
    a1[p] = (p < npb && k0 + 1 < r) ? A[(int64_t)(np0 + p) * r + k0 + 1] : 0.0;
  }

  const int64_t nsteps = (hi - lo + ROWS_IT - 1) / ROWS_IT;
  for (int64_t s = wl; s < nsteps; s += wpf) {
    const int64_t rbase = lo + s * ROWS_IT + (wave * RC_UNR) * RPW + grp;
    f64x2 u[RC_UNR];
#pragma unroll
    for (int j = 0; j < RC_UNR; ++j) {
      const int64_t row = rbase + j * RPW;
      f64x2 t = {0.0, 0.0};
      if (row < hi) {
        const double *rp = Ur + row * ldu;
        if (vec_ok) {
          if (k0 < r) t = *reinterpret_cast<const f64x2 *>(rp + k0);
        } else {
          if (k0 < r) t.x = rp[k0];
          if (k0 + 1 < r) t.y = rp[k0 + 1];
        }
      }
      u[j] = t;
    }
#pragma unroll
    for (int j = 0; j < RC_UNR; ++j) {
      const int64_t row = rbase + j * RPW;
      const double mu = (row < hi && lig == 0) ? rowmean[row] : 0.0;
      const double rs = (rowscale && row < hi && lig == 0) ? rowscale[row] : sc;   // sampled rows carry their own scale
#pragma unroll
      for (int p = 0; p < RC_PB; ++p) {
        if (p < npb) {
          double d = u[j].x * a0[p] + u[j].y * a1[p];
          d = group_sum_t<LPR>(d);
          if (lig == 0 && row < hi) out[(int64_t)(np0 + p) * ldo + row] = rs * d + mu;
        }
      }
    }
  }
}

template <int LPR>
int launch(const double *Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0, int64_t n_points,
           int32_t n_features, const double *rowmean, const double *scale, const double *rowscale, const double *A,
           int32_t n_p, double *out, int64_t ldo, hipStream_t st) {
  constexpr int RPW = 64 / LPR;
  constexpr int ROWS_IT = (RC_THREADS / 64) * RPW * RC_UNR;
  const int cus = spr_cached_cus();
  SegPlan plan;
  plan.row0 = row0; plan.n_rows = n_rows; plan.n_points = n_points; plan.n_features = n_features;
  plan.total_wg = 8 * (cus > 0 ? cus : 256);
  plan.chunk_rows = ROWS_IT;
  const int grid = seg_total_wgs(plan);
  const int vec_ok = (r % 2 == 0) && (ldu % 2 == 0) && ((reinterpret_cast<uintptr_t>(Ur) & 15) == 0);
  for (int p0 = 0; p0 < n_p; p0 += RC_PB) {
    const int npb = (n_p - p0 < RC_PB) ? n_p - p0 : RC_PB;
    hipLaunchKernelGGL(reconstruct_kernel<LPR>, dim3(grid), dim3(RC_THREADS), 0, st, Ur, (int)r, ldu, vec_ok,
                       plan, rowmean, scale, rowscale, A, p0, npb, out, ldo);
    SPR_LAUNCH_CHECK();
  }
  return SPR_OK;
}

}  // namespace

extern "C" int spr_reconstruct_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                                   int64_t n_points, int32_t n_features, const double *d_rowmean,
                                   const double *d_scale, const double *d_rowscale, const double *d_A, int32_t n_p,
                                   double *d_Xrec, int64_t ldo, void *stream) {
  SPR_REQUIRE(d_Ur && d_rowmean && d_scale && d_A && d_Xrec, SPR_E_INVALID, "spr_reconstruct_f64: NULL pointer");
  SPR_REQUIRE(n_rows > 0 && r > 0 && ldu >= r && n_p > 0 && ldo >= n_rows, SPR_E_INVALID,
              "spr_reconstruct_f64: bad shape n_rows=%lld r=%d ldu=%lld n_p=%d ldo=%lld", (long long)n_rows, r,
              (long long)ldu, n_p, (long long)ldo);
  SPR_REQUIRE(n_points > 0 && n_features > 0 && row0 >= 0 &&
                  row0 + n_rows <= n_points * (int64_t)n_features,
              SPR_E_INVALID, "spr_reconstruct_f64: bad feature layout");
  SPR_REQUIRE(r <= SPR_MAX_R, SPR_E_UNSUPPORTED, "spr_reconstruct_f64: r=%d > %d not built", r, SPR_MAX_R);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int half = (r + 1) / 2;
#define RC(L) return launch<L>(d_Ur, n_rows, r, ldu, row0, n_points, n_features, d_rowmean, d_scale, d_rowscale, d_A, n_p, d_Xrec, ldo, st)
  if (half <= 1) RC(1);
  if (half <= 2) RC(2);
  if (half <= 4) RC(4);
  if (half <= 8) RC(8);
  if (half <= 16) RC(16);
  if (half <= 32) RC(32);
  RC(64);
#undef RC
}
