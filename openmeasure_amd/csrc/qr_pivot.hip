// K6: QR column pivoting of Ur^T -- greedy max-residual-norm row selection.
//
// dgeqp3 on the r x n matrix Ur^T (sparse_sensing.py:739) picks, at step j, the column
// with the largest residual norm after projecting out the j columns already chosen.
// Only the pivot ORDER is used by the reference (:740-743), so no Householder vectors or
// R factor are formed: per step the rank owning the winner contributes its row u_p, the
// residual direction q_j = (I - Q Q^T) u_p / |..| is built in the r-dimensional
// coefficient space (classical Gram-Schmidt, applied twice), and every row's squared
// residual norm is down-dated by (u_i . q_j)^2 in one streaming pass over Ur.  Ties go to
// the lowest global row index, as LAPACK's idamax does.
//
// Candidate record (one per rank, all-gathered between steps when sharded), r+3 doubles:
//   [0] best residual norm^2   [1] its global row (as double, exact below 2^53)
//   [2] runner-up norm^2 on this rank   [3..3+r) the row of Ur
#include "common.hpp"

namespace {

constexpr int QR_THREADS = 256;
constexpr int QR_UNR = 4;
constexpr int QR_MAX_PART = 4096;

struct Best {
  double v1; int64_t i1; double v2;
  __device__ inline void init() { v1 = -2.0; i1 = INT64_MAX; v2 = -2.0; }
  __device__ inline void push(double v, int64_t i) {
    if (v > v1 || (v == v1 && i < i1)) { v2 = v1; v1 = v; i1 = i; }
    else if (v > v2) v2 = v;
  }
  __device__ inline void merge(double ov1, int64_t oi1, double ov2) {
    if (ov1 > v1 || (ov1 == v1 && oi1 < i1)) {
      v2 = (v1 > ov2) ? v1 : ov2; v1 = ov1; i1 = oi1;
    } else {
      if (ov1 > v2) v2 = ov1;
    }
  }
};

__device__ inline void block_best(Best &b, double *sv1, long long *si1, double *sv2, double *part) {
  // wave butterfly, then across waves through LDS; thread 0 writes (v1, i1, v2) to part[0..2]
  for (int o = 32; o > 0; o >>= 1) {
    const double ov1 = __shfl_xor(b.v1, o, 64);
    const long long oi1 = __shfl_xor((long long)b.i1, o, 64);
    const double ov2 = __shfl_xor(b.v2, o, 64);
    b.merge(ov1, oi1, ov2);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sv1[wave] = b.v1; si1[wave] = b.i1; sv2[wave] = b.v2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    Best t; t.init();
    for (int w = 0; w < QR_THREADS / 64; ++w) t.merge(sv1[w], si1[w], sv2[w]);
    part[0] = t.v1; part[1] = (double)t.i1; part[2] = t.v2;
  }
}

// MODE 0: nrm = |u|^2.  MODE 1: nrm -= (u.q)^2, pivot row -> -1.  Both: per-block best.
template <int LPR, int MODE>
__global__ __launch_bounds__(QR_THREADS) void qr_sweep_kernel(
    const double *__restrict__ Ur, int64_t n_rows, int r, int64_t ldu, int vec_ok_i, int64_t row0,
    const double *__restrict__ q, const int64_t *__restrict__ piv_ptr, double *__restrict__ nrm,
    double *__restrict__ part) {
  constexpr int RPW = 64 / LPR;
  constexpr int ROWS_IT = (QR_THREADS / 64) * RPW * QR_UNR;
  __shared__ double sv1[QR_THREADS / 64], sv2[QR_THREADS / 64];
  __shared__ long long si1[QR_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / LPR, lig = lane % LPR;
  const bool vec_ok = vec_ok_i != 0;
  const int k0 = 2 * lig;
  double q0 = 0.0, q1 = 0.0;
  int64_t piv = -1;
  if (MODE == 1) {
    if (k0 < r) q0 = q[k0];
    if (k0 + 1 < r) q1 = q[k0 + 1];
    piv = *piv_ptr;
  }
  Best best; best.init();
  const int64_t nsteps = (n_rows + ROWS_IT - 1) / ROWS_IT;
  for (int64_t s = blockIdx.x; s < nsteps; s += gridDim.x) {
    const int64_t rbase = s * ROWS_IT + (wave * QR_UNR) * RPW + grp;
    f64x2 u[QR_UNR];
#pragma unroll
    for (int j = 0; j < QR_UNR; ++j) {
      const int64_t row = rbase + j * RPW;
      f64x2 t = {0.0, 0.0};
      if (row < n_rows) {
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
    for (int j = 0; j < QR_UNR; ++j) {
      const int64_t row = rbase + j * RPW;
      double d = (MODE == 0) ? (u[j].x * u[j].x + u[j].y * u[j].y) : (u[j].x * q0 + u[j].y * q1);
      d = group_sum_t<LPR>(d);
      if (lig == 0 && row < n_rows) {
        double v;
        if (MODE == 0) {
          v = d;
        } else {
          const double old = nrm[row];
          v = old - d * d;
          if (v < 0.0) v = 0.0;
          if (old < 0.0 || row0 + row == piv) v = -1.0;  // chosen rows leave the race
        }
        nrm[row] = v;
        best.push(v, row0 + row);
      }
    }
  }
  block_best(best, sv1, si1, sv2, part + 3 * (int64_t)blockIdx.x);
}

// one workgroup: reduce the per-block bests, emit this rank's candidate record
__global__ __launch_bounds__(QR_THREADS) void qr_candidate_kernel(
    const double *__restrict__ part, int n_part, const double *__restrict__ Ur, int r, int64_t ldu,
    int64_t row0, int64_t n_rows, double *__restrict__ cand) {
  __shared__ double sv1[QR_THREADS / 64], sv2[QR_THREADS / 64];
  __shared__ long long si1[QR_THREADS / 64];
  __shared__ double res[3];
  Best b; b.init();
  for (int p = threadIdx.x; p < n_part; p += QR_THREADS)
    b.merge(part[3 * p], (int64_t)part[3 * p + 1], part[3 * p + 2]);
  block_best(b, sv1, si1, sv2, res);
  __syncthreads();
  const int64_t gi = (int64_t)res[1];
  if (threadIdx.x == 0) { cand[0] = res[0]; cand[1] = res[1]; cand[2] = res[2]; }
  const int64_t li = gi - row0;
  for (int k = threadIdx.x; k < r; k += QR_THREADS)
    cand[3 + k] = (li >= 0 && li < n_rows) ? Ur[li * ldu + k] : 0.0;
}

// one workgroup: pick the winner among the ranks' candidates, orthogonalise, store q / pivot
__global__ __launch_bounds__(QR_THREADS) void qr_orth_kernel(
    const double *__restrict__ cands, int n_cand, int r, int step, double *__restrict__ Q,
    int64_t *__restrict__ piv, double *__restrict__ gap) {
  __shared__ double v[SPR_MAX_R], c[SPR_MAX_R];
  __shared__ double red[QR_THREADS / 64];
  __shared__ int win;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int stride = r + 3;
  if (threadIdx.x == 0) {
    int w = 0;
    for (int i = 1; i < n_cand; ++i) {
      const double vi = cands[(int64_t)i * stride], vw = cands[(int64_t)w * stride];
      if (vi > vw || (vi == vw && cands[(int64_t)i * stride + 1] < cands[(int64_t)w * stride + 1])) w = i;
    }
    win = w;
    const double bestv = cands[(int64_t)w * stride];
    double second = cands[(int64_t)w * stride + 2];
    for (int i = 0; i < n_cand; ++i)
      if (i != w && cands[(int64_t)i * stride] > second) second = cands[(int64_t)i * stride];
    piv[step] = (int64_t)cands[(int64_t)w * stride + 1];
    if (gap) gap[step] = (bestv > 0.0) ? (bestv - second) / bestv : 0.0;
  }
  __syncthreads();
  const double *row = cands + (int64_t)win * stride + 3;
  for (int k = threadIdx.x; k < r; k += QR_THREADS) v[k] = row[k];
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {
    for (int t = wave; t < step; t += QR_THREADS / 64) {
      double d = 0.0;
      for (int k = lane; k < r; k += 64) d += Q[(int64_t)t * r + k] * v[k];
      d = group_sum(d, 64);
      if (lane == 0) c[t] = d;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < r; k += QR_THREADS) {
      double acc = v[k];
      for (int t = 0; t < step; ++t) acc -= c[t] * Q[(int64_t)t * r + k];
      v[k] = acc;
    }
    __syncthreads();
  }
  double s = 0.0;
  for (int k = threadIdx.x; k < r; k += QR_THREADS) s += v[k] * v[k];
  s = group_sum(s, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  double nn = 0.0;
  for (int w = 0; w < QR_THREADS / 64; ++w) nn += red[w];
  const double inv = (nn > 0.0) ? 1.0 / sqrt(nn) : 0.0;
  for (int k = threadIdx.x; k < r; k += QR_THREADS) Q[(int64_t)step * r + k] = v[k] * inv;
}

__global__ void mask_rows_kernel(double *__restrict__ Ur, int64_t n_rows, int r, int64_t ldu,
                                 const uint8_t *__restrict__ mask) {
  const int64_t total = n_rows * r;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / r;
    if (!mask[row]) Ur[row * ldu + (e - row * r)] = 0.0;
  }
}

int sweep_grid(int64_t n_rows, int lpr) {
  const int rows_it = (QR_THREADS / 64) * (64 / lpr) * QR_UNR;
  int64_t steps = (n_rows + rows_it - 1) / rows_it;
  const int cus = spr_cached_cus();
  int64_t cap = 8LL * (cus > 0 ? cus : 256);
  if (cap > QR_MAX_PART) cap = QR_MAX_PART;
  return (int)(steps < cap ? steps : cap);
}

int pick_lpr(int r) {
  const int half = (r + 1) / 2;
  int l = 1;
  while (l < half) l *= 2;
  return l;
}

template <int MODE>
int launch_sweep(int lpr, int grid, hipStream_t st, const double *Ur, int64_t n_rows, int r, int64_t ldu,
                 int vec_ok, int64_t row0, const double *q, const int64_t *piv, double *nrm, double *part) {
#define SW(L) hipLaunchKernelGGL((qr_sweep_kernel<L, MODE>), dim3(grid), dim3(QR_THREADS), 0, st, Ur, n_rows, r, ldu, vec_ok, row0, q, piv, nrm, part); break
  switch (lpr) {
    case 1: SW(1);
    case 2: SW(2);
    case 4: SW(4);
    case 8: SW(8);
    case 16: SW(16);
    case 32: SW(32);
    default: SW(64);
  }
#undef SW
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

int check_ur(const char *who, const double *Ur, int64_t n_rows, int32_t r, int64_t ldu) {
  SPR_REQUIRE(Ur != nullptr, SPR_E_INVALID, "%s: Ur is NULL", who);
  SPR_REQUIRE(n_rows > 0 && r > 0 && ldu >= r, SPR_E_INVALID, "%s: bad shape n_rows=%lld r=%d ldu=%lld", who,
              (long long)n_rows, r, (long long)ldu);
  SPR_REQUIRE(r <= SPR_MAX_R, SPR_E_UNSUPPORTED, "%s: r=%d > %d not built", who, r, SPR_MAX_R);
  return SPR_OK;
}

}  // namespace

extern "C" size_t spr_qr_workspace(int64_t n_rows) {
  (void)n_rows;
  return (size_t)QR_MAX_PART * 3 * sizeof(double);
}

extern "C" int spr_mask_rows_f64(double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, const uint8_t *d_mask,
                                 void *stream) {
  int rc = check_ur("spr_mask_rows_f64", d_Ur, n_rows, r, ldu);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_mask != nullptr, SPR_E_INVALID, "spr_mask_rows_f64: mask is NULL");
  const int64_t total = n_rows * r;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mask_rows_kernel, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_Ur,
                     n_rows, (int)r, ldu, d_mask);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

extern "C" int spr_qr_init_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                               double *d_nrm, double *d_cand, void *d_workspace, size_t workspace_bytes,
                               void *stream) {
  int rc = check_ur("spr_qr_init_f64", d_Ur, n_rows, r, ldu);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_nrm && d_cand && d_workspace, SPR_E_INVALID, "spr_qr_init_f64: NULL pointer");
  SPR_REQUIRE(workspace_bytes >= spr_qr_workspace(n_rows), SPR_E_WORKSPACE, "spr_qr_init_f64: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int lpr = pick_lpr(r), grid = sweep_grid(n_rows, lpr);
  const int vec_ok = (r % 2 == 0) && (ldu % 2 == 0) && ((reinterpret_cast<uintptr_t>(d_Ur) & 15) == 0);
  double *part = static_cast<double *>(d_workspace);
  rc = launch_sweep<0>(lpr, grid, st, d_Ur, n_rows, r, ldu, vec_ok, row0, nullptr, nullptr, d_nrm, part);
  if (rc != SPR_OK) return rc;
  hipLaunchKernelGGL(qr_candidate_kernel, dim3(1), dim3(QR_THREADS), 0, st, part, grid, d_Ur, (int)r, ldu, row0,
                     n_rows, d_cand);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

extern "C" int spr_qr_step_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                               int32_t step, const double *d_cands, int32_t n_cand, double *d_Q, int64_t *d_piv,
                               double *d_nrm, double *d_cand, double *d_gap, void *d_workspace,
                               size_t workspace_bytes, void *stream) {
  int rc = check_ur("spr_qr_step_f64", d_Ur, n_rows, r, ldu);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_cands && d_Q && d_piv && d_nrm && d_cand && d_workspace, SPR_E_INVALID,
              "spr_qr_step_f64: NULL pointer");
  SPR_REQUIRE(step >= 0 && step < r && n_cand >= 1, SPR_E_INVALID, "spr_qr_step_f64: bad step=%d n_cand=%d", step,
              n_cand);
  SPR_REQUIRE(workspace_bytes >= spr_qr_workspace(n_rows), SPR_E_WORKSPACE, "spr_qr_step_f64: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int lpr = pick_lpr(r), grid = sweep_grid(n_rows, lpr);
  const int vec_ok = (r % 2 == 0) && (ldu % 2 == 0) && ((reinterpret_cast<uintptr_t>(d_Ur) & 15) == 0);
  double *part = static_cast<double *>(d_workspace);
  hipLaunchKernelGGL(qr_orth_kernel, dim3(1), dim3(QR_THREADS), 0, st, d_cands, (int)n_cand, (int)r, (int)step,
                     d_Q, d_piv, d_gap);
  SPR_LAUNCH_CHECK();
  rc = launch_sweep<1>(lpr, grid, st, d_Ur, n_rows, r, ldu, vec_ok, row0, d_Q + (int64_t)step * r, d_piv + step,
                       d_nrm, part);
  if (rc != SPR_OK) return rc;
  hipLaunchKernelGGL(qr_candidate_kernel, dim3(1), dim3(QR_THREADS), 0, st, part, grid, d_Ur, (int)r, ldu, row0,
                     n_rows, d_cand);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}
