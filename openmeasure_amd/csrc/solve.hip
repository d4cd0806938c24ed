// K8 + K9: scale_vector + (weighted) ordinary least squares for a batch of measurement
// vectors -- one 256-thread workgroup per vector.
//
// Reference (sparse_sensing.py:866-878): y0 = ((y-cnt)/scl, sigma/scl); W = I when every
// sigma is zero, else diag(1/y0_sigma); a = pinv(W Theta) (W y0), sigma_a = |pinv(W Theta) y0_sigma|.
// For a full-column-rank W Theta the pseudo-inverse solution is the normal-equations
// solution, so the kernel forms the augmented Gram matrix of [W Theta | W y0 | y0_sigma]
// (s x (r+2)) with v_mfma_f64_16x16x4_f64 -- the same 16x16-tile scheme as the snapshot
// Gram kernel, with the sensors as the contraction index -- equilibrates it to unit
// diagonal, factors N' = L L^T in LDS and solves both right-hand sides, then takes ONE step of iterative refinement
// with the residual formed from W Theta itself (corrected semi-normal equations: x += N^-1 A^T (b - A x)), which
// brings the error from cond(A)^2 eps back to the cond(A) eps of a QR solve as long as cond(A)^2 eps < 1.
// info[1] = (max L_jj / min L_jj)^2 is returned so the caller can refuse a system beyond that range instead of
// losing digits silently.
#include "common.hpp"

namespace {

constexpr int SV_THREADS = 256;
constexpr int SV_WAVES = SV_THREADS / 64;

template <int NT> struct SolveCfg {
  static constexpr int NAP = 16 * NT;                            // padded augmented width
  static constexpr int CP = NAP + ((NT % 2 == 0) ? 16 : 0);      // LDS row stride (== 16 mod 32)
  static constexpr int SC = (NT >= 9) ? 16 : 32;                 // sensors per panel
  static constexpr int RMAX = (NAP - 2 > SPR_MAX_R) ? SPR_MAX_R : NAP - 2;
  static constexpr int LDN = RMAX + 1;
  static constexpr int T = NT * (NT + 1) / 2;
  static constexpr int TPW = (T + SV_WAVES - 1) / SV_WAVES;
};

template <int NT>
__device__ inline void tri_coords(int idx, int &ti, int &tj) {
  ti = 0;
  while (ti < NT - 1 && idx >= NT - ti) { idx -= NT - ti; ++ti; }
  tj = ti + idx;
  if (tj > NT - 1) tj = NT - 1;
}

template <int NT>
__global__ __launch_bounds__(SV_THREADS) void solve_ols_kernel(
    const double *__restrict__ Theta, int s, int r, const double *__restrict__ cnt,
    const double *__restrict__ scale, int n_features, const double *__restrict__ y_all,
    double *__restrict__ Ar, double *__restrict__ Ar_sigma, double *__restrict__ y0_all,
    double *__restrict__ info) {
  using C = SolveCfg<NT>;
  constexpr int CP = C::CP, SC = C::SC, LDN = C::LDN, T = C::T, TPW = C::TPW, NAP = C::NAP;
  __shared__ double panel[SC * CP];
  __shared__ double N[C::RMAX * LDN];
  __shared__ double rhs[2][C::RMAX];
  __shared__ double sol[2][C::RMAX];
  __shared__ double dsc[C::RMAX];
  __shared__ double xs[2][C::RMAX];
  __shared__ double sw[SC], sv[SC], ss[SC];
  __shared__ int flags[3];  // [0] any sigma != 0, [1] Cholesky breakdown, [2] a non-finite weight 1/sigma

  const int p = blockIdx.x;
  const double *y = y_all + (int64_t)p * s * 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  if (tid < 3) flags[tid] = 0;
  __syncthreads();
  {
    int any = 0;
    for (int k = tid; k < s; k += SV_THREADS) any |= (y[3 * k + 1] != 0.0);
    if (any) flags[0] = 1;
  }
  __syncthreads();
  const bool weighted = flags[0] != 0;

  int offA[TPW], offB[TPW];
  int nt = T - wave * TPW;
  if (nt > TPW) nt = TPW;
  if (nt < 0) nt = 0;
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    int ti, tj;
    tri_coords<NT>(wave * TPW + u, ti, tj);
    offA[u] = ti * 16;
    offB[u] = tj * 16;
  }
  f64x4 acc[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) acc[u] = (f64x4){0.0, 0.0, 0.0, 0.0};

  const int frag = (lane >> 4) * CP + (lane & 15);
  for (int c0 = 0; c0 < s; c0 += SC) {
    __syncthreads();  // previous panel fully consumed
    if (tid < SC) {
      const int k = c0 + tid;
      double w = 0.0, v0 = 0.0, s0 = 0.0;
      if (k < s) {
        int f = (int)y[3 * k + 2];
        if (f < 0) f = 0;
        if (f > n_features - 1) f = n_features - 1;
        const double scl = scale[f];
        v0 = (y[3 * k] - cnt[k]) / scl;
        s0 = y[3 * k + 1] / scl;
        w = weighted ? 1.0 / s0 : 1.0;
        if (!isfinite(w)) flags[2] = 1;     // sigma zero or NaN for SOME sensors: W = diag(1/0) in the reference (:872)
        if (y0_all) {
          y0_all[((int64_t)p * s + k) * 2] = v0;
          y0_all[((int64_t)p * s + k) * 2 + 1] = s0;
        }
      }
      sw[tid] = w; sv[tid] = v0; ss[tid] = s0;
    }
    __syncthreads();
    for (int e = tid; e < SC * NAP; e += SV_THREADS) {
      const int kk = e / NAP, c = e - kk * NAP;
      const int k = c0 + kk;
      double val = 0.0;
      if (k < s) {
        if (c < r) val = sw[kk] * Theta[(int64_t)k * r + c];
        else if (c == r) val = sw[kk] * sv[kk];
        else if (c == r + 1) val = weighted ? ss[kk] : 0.0;
      }
      panel[kk * CP + c] = val;
    }
    __syncthreads();
#pragma unroll 1
    for (int k0 = 0; k0 < SC; k0 += 4) {
#pragma unroll
      for (int u = 0; u < TPW; ++u) {
        if (u < nt) {
          const double a = panel[frag + k0 * CP + offA[u]];
          const double b = panel[frag + k0 * CP + offB[u]];
          acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
        }
      }
    }
  }

  // accumulators -> N (both triangles) and the two right-hand sides
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    if (u < nt) {
      const double vals[4] = {acc[u].x, acc[u].y, acc[u].z, acc[u].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gi = offA[u] + (lane >> 4) + 4 * i, gj = offB[u] + (lane & 15);
        if (gi < r) {
          if (gj < r) { N[gi * LDN + gj] = vals[i]; N[gj * LDN + gi] = vals[i]; }
          else if (gj == r) rhs[0][gi] = vals[i];
          else if (gj == r + 1) rhs[1][gi] = vals[i];
        }
      }
    }
  }
  __syncthreads();

  // Jacobi equilibration: unit-diagonal N' = D N D, D = diag(N)^-1/2.  Removes the effect of
  // column scaling (e.g. a retained mode whose basis column is tiny) on the conditioning.
  for (int j = tid; j < r; j += SV_THREADS) {
    const double d = N[j * LDN + j];
    dsc[j] = (d > 0.0) ? 1.0 / sqrt(d) : 1.0;
  }
  __syncthreads();
  for (int e = tid; e < r * r; e += SV_THREADS) {
    const int i = e / r, j = e - i * r;
    N[i * LDN + j] *= dsc[i] * dsc[j];
  }
  if (tid < r) { rhs[0][tid] *= dsc[tid]; rhs[1][tid] *= dsc[tid]; }
  __syncthreads();

  // Cholesky N' = L L^T (lower, in place)
  for (int j = 0; j < r; ++j) {
    if (tid == 0) {
      double d = N[j * LDN + j];
      if (!(d > 0.0)) { flags[1] = 1; d = 1e-300; }
      N[j * LDN + j] = sqrt(d);
    }
    __syncthreads();
    const double djj = N[j * LDN + j];
    for (int i = j + 1 + tid; i < r; i += SV_THREADS) N[i * LDN + j] /= djj;
    __syncthreads();
    const int cntj = r - j - 1;
    for (int e = tid; e < cntj * cntj; e += SV_THREADS) {
      const int a = e / cntj, b = e - a * cntj;
      if (b <= a) {
        const int i = j + 1 + a, k = j + 1 + b;
        N[i * LDN + k] -= N[i * LDN + j] * N[k * LDN + j];
      }
    }
    __syncthreads();
  }
  // forward substitution L z = rhs (both right-hand sides), then L^T x = z; the solution ends up in rhs
  auto chol_solve = [&]() {
    for (int j = 0; j < r; ++j) {
      if (tid < 2) sol[tid][j] = rhs[tid][j] / N[j * LDN + j];
      __syncthreads();
      for (int i = j + 1 + tid; i < r; i += SV_THREADS) {
        const double l = N[i * LDN + j];
        rhs[0][i] -= l * sol[0][j];
        rhs[1][i] -= l * sol[1][j];
      }
      __syncthreads();
    }
    for (int j = r - 1; j >= 0; --j) {
      if (tid < 2) rhs[tid][j] = sol[tid][j] / N[j * LDN + j];
      __syncthreads();
      for (int i = tid; i < j; i += SV_THREADS) {
        const double l = N[j * LDN + i];
        sol[0][i] -= l * rhs[0][j];
        sol[1][i] -= l * rhs[1][j];
      }
      __syncthreads();
    }
  };
  chol_solve();
  // x = D x' so far; one refinement step: e = b - A x per sensor, g = A^T e, x' += N'^-1 (D g)
  if (tid < r) { xs[0][tid] = rhs[0][tid]; xs[1][tid] = rhs[1][tid]; rhs[0][tid] = 0.0; rhs[1][tid] = 0.0; }
  __syncthreads();
  for (int c0 = 0; c0 < s; c0 += SC) {
    if (tid < SC) {
      const int k = c0 + tid;
      double w = 0.0, v0 = 0.0, s0 = 0.0;
      if (k < s) {
        int f = (int)y[3 * k + 2];
        if (f < 0) f = 0;
        if (f > n_features - 1) f = n_features - 1;
        const double scl = scale[f];
        v0 = (y[3 * k] - cnt[k]) / scl;
        s0 = y[3 * k + 1] / scl;
        w = weighted ? 1.0 / s0 : 1.0;
        if (!isfinite(w)) flags[2] = 1;     // sigma zero or NaN for SOME sensors: W = diag(1/0) in the reference (:872)
      }
      sw[tid] = w; sv[tid] = v0; ss[tid] = s0;
    }
    __syncthreads();
    for (int e = tid; e < SC * NAP; e += SV_THREADS) {
      const int kk = e / NAP, c = e - kk * NAP;
      const int k = c0 + kk;
      panel[kk * CP + c] = (k < s && c < r) ? sw[kk] * Theta[(int64_t)k * r + c] : 0.0;
    }
    __syncthreads();
    // residuals of the panel's sensors: wave w takes sensors w, w+4, ...
    for (int kk = wave; kk < SC; kk += SV_WAVES) {
      double d0 = 0.0, d1 = 0.0;
      for (int c = lane; c < r; c += 64) {
        const double a = panel[kk * CP + c];
        d0 += a * xs[0][c] * dsc[c];
        d1 += a * xs[1][c] * dsc[c];
      }
      d0 = group_sum(d0, 64);
      d1 = group_sum(d1, 64);
      if (lane == 0) {
        const bool live = c0 + kk < s;
        sv[kk] = live ? sw[kk] * sv[kk] - d0 : 0.0;            // b = W y0
        ss[kk] = (live && weighted) ? ss[kk] - d1 : 0.0;       // b = y0_sigma
      }
    }
    __syncthreads();
    if (tid < r) {
      double g0 = 0.0, g1 = 0.0;
      for (int kk = 0; kk < SC; ++kk) {
        const double a = panel[kk * CP + tid];
        g0 += a * sv[kk];
        g1 += a * ss[kk];
      }
      rhs[0][tid] += g0;
      rhs[1][tid] += g1;
    }
    __syncthreads();
  }
  if (tid < r) { rhs[0][tid] *= dsc[tid]; rhs[1][tid] *= dsc[tid]; }
  __syncthreads();
  chol_solve();
  if (tid < r) { rhs[0][tid] += xs[0][tid]; rhs[1][tid] += xs[1][tid]; }
  __syncthreads();
  for (int k = tid; k < r; k += SV_THREADS) {
    Ar[(int64_t)p * r + k] = rhs[0][k] * dsc[k];
    Ar_sigma[(int64_t)p * r + k] = weighted ? fabs(rhs[1][k] * dsc[k]) : 0.0;
  }
  if (tid == 0) {
    double dmax = 0.0, dmin = 1e300;
    for (int j = 0; j < r; ++j) {
      const double d = N[j * LDN + j];
      if (d > dmax) dmax = d;
      if (d < dmin) dmin = d;
    }
    info[2 * p] = flags[2] ? 2.0 : (double)flags[1];     // 2: non-finite weights (np.linalg.pinv raises), 1: Cholesky breakdown
    info[2 * p + 1] = (dmax / dmin) * (dmax / dmin);
  }
}

// ---- the same algorithm for r > 128 (r <= SPR_MAX_R_WIDE): the normal matrix no longer fits LDS, so the scaled augmented
// matrix A = [W Theta | W y0 | y0_sigma] (s x (r+2)) and N = A^T A (r x r) live in a global workspace (L2-resident) and one
// 1024-thread workgroup per vector works on them with plain FMA loops -- a basis this wide is rare (the reference keeps any
// r <= m modes, :336), the point is that predict() stays in milliseconds there instead of falling to the SVD path.
constexpr int SW_THREADS = 1024;

__global__ __launch_bounds__(SW_THREADS) void solve_ols_wide_kernel(
    const double *__restrict__ Theta, int s, int r, const double *__restrict__ cnt, const double *__restrict__ scale,
    int n_features, const double *__restrict__ y_all, double *__restrict__ Ar, double *__restrict__ Ar_sigma,
    double *__restrict__ y0_all, double *__restrict__ info, double *ws, int64_t ws_stride) {
  const int nc = r + 2, LDN = r + 1;
  double *A = ws + (int64_t)blockIdx.x * ws_stride;      // s x nc
  double *N = A + (int64_t)s * nc;                        // r x LDN
  double *rhs0 = N + (int64_t)r * LDN, *rhs1 = rhs0 + r, *sol0 = rhs1 + r, *sol1 = sol0 + r, *dsc = sol1 + r;
  double *xs0 = dsc + r, *xs1 = xs0 + r, *res0 = xs1 + r, *res1 = res0 + s;   // residuals: s entries each
  __shared__ int flags[3];
  __shared__ double piv[2];
  const int p = blockIdx.x, tid = threadIdx.x;
  const double *y = y_all + (int64_t)p * s * 3;
  if (tid < 3) flags[tid] = 0;
  __syncthreads();
  {
    int any = 0;
    for (int k = tid; k < s; k += SW_THREADS) any |= (y[3 * k + 1] != 0.0);
    if (any) flags[0] = 1;
  }
  __syncthreads();
  const bool weighted = flags[0] != 0;
  // scale_vector (:571-582) and the rows of A
  for (int k = tid; k < s; k += SW_THREADS) {
    int f = (int)y[3 * k + 2];
    f = f < 0 ? 0 : (f > n_features - 1 ? n_features - 1 : f);
    const double scl = scale[f];
    const double v0 = (y[3 * k] - cnt[k]) / scl, s0 = y[3 * k + 1] / scl;
    const double w = weighted ? 1.0 / s0 : 1.0;
    if (!isfinite(w)) flags[2] = 1;         // sigma zero or NaN for SOME sensors: W = diag(1/0) in the reference (:872)
    if (y0_all) {
      y0_all[((int64_t)p * s + k) * 2] = v0;
      y0_all[((int64_t)p * s + k) * 2 + 1] = s0;
    }
    A[(int64_t)k * nc + r] = w * v0;
    A[(int64_t)k * nc + r + 1] = weighted ? s0 : 0.0;
    res0[k] = w;                                          // the row weight, parked until the rows are scaled
  }
  __syncthreads();
  for (int64_t e = tid; e < (int64_t)s * r; e += SW_THREADS) {
    const int k = (int)(e / r), c = (int)(e - (int64_t)k * r);
    A[(int64_t)k * nc + c] = res0[k] * Theta[e];
  }
  __syncthreads();
  // N = A^T A (upper triangle computed, mirrored), rhs = A^T b for both right-hand sides
  for (int64_t e = tid; e < (int64_t)r * nc; e += SW_THREADS) {
    const int i = (int)(e / nc), j = (int)(e - (int64_t)i * nc);
    if (j >= i) {
      double acc = 0.0;
      for (int k = 0; k < s; ++k) acc += A[(int64_t)k * nc + i] * A[(int64_t)k * nc + j];
      if (j < r) { N[(int64_t)i * LDN + j] = acc; N[(int64_t)j * LDN + i] = acc; }
      else if (j == r) rhs0[i] = acc;
      else rhs1[i] = acc;
    }
  }
  __syncthreads();
  for (int j = tid; j < r; j += SW_THREADS) {
    const double d = N[(int64_t)j * LDN + j];
    dsc[j] = (d > 0.0) ? 1.0 / sqrt(d) : 1.0;
  }
  __syncthreads();
  for (int64_t e = tid; e < (int64_t)r * r; e += SW_THREADS) {
    const int i = (int)(e / r), j = (int)(e - (int64_t)i * r);
    N[(int64_t)i * LDN + j] *= dsc[i] * dsc[j];
  }
  for (int j = tid; j < r; j += SW_THREADS) { rhs0[j] *= dsc[j]; rhs1[j] *= dsc[j]; }
  __syncthreads();
  // Cholesky N' = L L^T, right-looking, in place (lower)
  for (int j = 0; j < r; ++j) {
    if (tid == 0) {
      double d = N[(int64_t)j * LDN + j];
      if (!(d > 0.0)) { flags[1] = 1; d = 1e-300; }
      piv[0] = sqrt(d);
      N[(int64_t)j * LDN + j] = piv[0];
    }
    __syncthreads();
    const double djj = piv[0];
    for (int i = j + 1 + tid; i < r; i += SW_THREADS) N[(int64_t)i * LDN + j] /= djj;
    __syncthreads();
    const int cntj = r - j - 1;
    for (int64_t e = tid; e < (int64_t)cntj * cntj; e += SW_THREADS) {
      const int a = (int)(e / cntj), b = (int)(e - (int64_t)a * cntj);
      if (b <= a) {
        const int i = j + 1 + a, k = j + 1 + b;
        N[(int64_t)i * LDN + k] -= N[(int64_t)i * LDN + j] * N[(int64_t)k * LDN + j];
      }
    }
    __syncthreads();
  }
  auto chol_solve = [&]() {     // L z = rhs, L^T x = z for both right-hand sides; the solution ends up in rhs
    for (int j = 0; j < r; ++j) {
      if (tid < 2) (tid ? sol1 : sol0)[j] = (tid ? rhs1 : rhs0)[j] / N[(int64_t)j * LDN + j];
      __syncthreads();
      for (int i = j + 1 + tid; i < r; i += SW_THREADS) {
        const double l = N[(int64_t)i * LDN + j];
        rhs0[i] -= l * sol0[j];
        rhs1[i] -= l * sol1[j];
      }
      __syncthreads();
    }
    for (int j = r - 1; j >= 0; --j) {
      if (tid < 2) (tid ? rhs1 : rhs0)[j] = (tid ? sol1 : sol0)[j] / N[(int64_t)j * LDN + j];
      __syncthreads();
      for (int i = tid; i < j; i += SW_THREADS) {
        const double l = N[(int64_t)j * LDN + i];
        sol0[i] -= l * rhs0[j];
        sol1[i] -= l * rhs1[j];
      }
      __syncthreads();
    }
  };
  chol_solve();
  // one refinement step with the residual formed from A itself (corrected semi-normal equations)
  for (int j = tid; j < r; j += SW_THREADS) { xs0[j] = rhs0[j]; xs1[j] = rhs1[j]; }
  __syncthreads();
  for (int k = tid; k < s; k += SW_THREADS) {
    double d0 = 0.0, d1 = 0.0;
    for (int c = 0; c < r; ++c) {
      const double a = A[(int64_t)k * nc + c];
      d0 += a * xs0[c] * dsc[c];
      d1 += a * xs1[c] * dsc[c];
    }
    res0[k] = A[(int64_t)k * nc + r] - d0;
    res1[k] = weighted ? A[(int64_t)k * nc + r + 1] - d1 : 0.0;
  }
  __syncthreads();
  for (int c = tid; c < r; c += SW_THREADS) {
    double g0 = 0.0, g1 = 0.0;
    for (int k = 0; k < s; ++k) {
      const double a = A[(int64_t)k * nc + c];
      g0 += a * res0[k];
      g1 += a * res1[k];
    }
    rhs0[c] = g0 * dsc[c];
    rhs1[c] = g1 * dsc[c];
  }
  __syncthreads();
  chol_solve();
  for (int k = tid; k < r; k += SW_THREADS) {
    Ar[(int64_t)p * r + k] = (rhs0[k] + xs0[k]) * dsc[k];
    Ar_sigma[(int64_t)p * r + k] = weighted ? fabs((rhs1[k] + xs1[k]) * dsc[k]) : 0.0;
  }
  if (tid == 0) {
    double dmax = 0.0, dmin = 1e300;
    for (int j = 0; j < r; ++j) {
      const double d = N[(int64_t)j * LDN + j];
      if (d > dmax) dmax = d;
      if (d < dmin) dmin = d;
    }
    info[2 * p] = flags[2] ? 2.0 : (double)flags[1];     // 2: non-finite weights (np.linalg.pinv raises), 1: Cholesky breakdown
    info[2 * p + 1] = (dmax / dmin) * (dmax / dmin);
  }
}

}  // namespace

extern "C" int spr_solve_ols_f64(const double *d_Theta, int32_t s, int32_t r, const double *d_cnt,
                                 const double *d_scale, int32_t n_features, const double *d_y, int32_t n_p,
                                 double *d_Ar, double *d_Ar_sigma, double *d_y0, double *d_info, void *stream) {
  SPR_REQUIRE(d_Theta && d_cnt && d_scale && d_y && d_Ar && d_Ar_sigma && d_info, SPR_E_INVALID,
              "spr_solve_ols_f64: NULL pointer");
  SPR_REQUIRE(s > 0 && r > 0 && n_p > 0 && n_features > 0, SPR_E_INVALID, "spr_solve_ols_f64: bad shape");
  SPR_REQUIRE(r <= SPR_MAX_R, SPR_E_UNSUPPORTED, "spr_solve_ols_f64: r=%d > %d not built", r, SPR_MAX_R);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define SV(NTV)                                                                                              \
  hipLaunchKernelGGL(solve_ols_kernel<NTV>, dim3(n_p), dim3(SV_THREADS), 0, st, d_Theta, (int)s, (int)r,     \
                     d_cnt, d_scale, (int)n_features, d_y, d_Ar, d_Ar_sigma, d_y0, d_info)
  const int need = (r + 2 + 15) / 16;
  if (need <= 1) SV(1);
  else if (need <= 2) SV(2);
  else if (need <= 3) SV(3);
  else if (need <= 5) SV(5);
  else SV(9);
#undef SV
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

// r > SPR_MAX_R: same semantics and outputs, matrices in a workspace of spr_solve_ols_workspace(s, r, n_p) bytes
extern "C" size_t spr_solve_ols_workspace(int32_t s, int32_t r, int32_t n_p) {
  if (r <= SPR_MAX_R || r > SPR_MAX_R_WIDE || s <= 0 || n_p <= 0) return 0;
  return sizeof(double) * (size_t)n_p * ((size_t)s * (r + 2) + (size_t)r * (r + 1) + 7 * (size_t)r + 2 * (size_t)s);
}

extern "C" int spr_solve_ols_wide_f64(const double *d_Theta, int32_t s, int32_t r, const double *d_cnt,
                                      const double *d_scale, int32_t n_features, const double *d_y, int32_t n_p,
                                      double *d_Ar, double *d_Ar_sigma, double *d_y0, double *d_info, void *d_workspace,
                                      size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(d_Theta && d_cnt && d_scale && d_y && d_Ar && d_Ar_sigma && d_info && d_workspace, SPR_E_INVALID,
              "spr_solve_ols_wide_f64: NULL pointer");
  SPR_REQUIRE(s > 0 && r > 0 && n_p > 0 && n_features > 0, SPR_E_INVALID, "spr_solve_ols_wide_f64: bad shape");
  SPR_REQUIRE(r <= SPR_MAX_R_WIDE, SPR_E_UNSUPPORTED, "spr_solve_ols_wide_f64: r=%d > %d not built", r, SPR_MAX_R_WIDE);
  const size_t per = (size_t)s * (r + 2) + (size_t)r * (r + 1) + 7 * (size_t)r + 2 * (size_t)s;
  SPR_REQUIRE(workspace_bytes >= sizeof(double) * per * (size_t)n_p, SPR_E_WORKSPACE,
              "spr_solve_ols_wide_f64: workspace too small");
  hipLaunchKernelGGL(solve_ols_wide_kernel, dim3(n_p), dim3(SW_THREADS), 0, static_cast<hipStream_t>(stream), d_Theta,
                     (int)s, (int)r, d_cnt, d_scale, (int)n_features, d_y, d_Ar, d_Ar_sigma, d_y0, d_info,
                     static_cast<double *>(d_workspace), (int64_t)per);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}
