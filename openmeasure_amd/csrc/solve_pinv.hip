// K9b: minimum-norm least squares with the reference's pseudo-inverse semantics.
//
// Reference (sparse_sensing.py:873-878): a = np.linalg.pinv(W Theta) (W y0), sigma_a = |pinv(W Theta) y0_sigma|,
// pinv through an SVD with rcond = 1e-15: singular values <= rcond * sigma_max are dropped, so a rank-deficient or
// underdetermined (s < r) system gets its minimum-norm solution.  The fast path (solve.hip: MFMA normal equations +
// Cholesky + one refinement step) is only valid for full column rank and cond(W Theta)^2 eps < 1; this kernel is what
// predict() runs for everything else -- s < r (every GEM placement), Cholesky breakdown, cond^2 > 1e13 -- so that no
// input the reference accepts is refused.  One 256-thread workgroup per measurement vector, everything in LDS:
//
//   1. scale_vector (:571-582) exactly as in solve.hip, rows of M = [W Theta | W y0 | y0_sigma]  (s x (r+2)).
//   2. s > r: M is reduced to its r x (r+2) triangular factor [R | Q^T b] by Householder reflections applied to
//      row panels as they stream in (stacked [R; panel] re-triangularised per panel: backward stable, one read of
//      Theta).  s <= r: the rows are used as they are.
//   3. One-sided (Hestenes) Jacobi on the ROWS of the q = min(s, r) x r factor: plane rotations G make the rows
//      mutually orthogonal, G R = S U^T, and are applied to the right-hand-side columns on the fly, so no
//      orthogonal factor is stored: R x = c  <=>  (S U^T) x = G c =: c', and the minimum-norm solution is
//          x = sum_{sigma_i > rcond sigma_max} row_i * c'_i / sigma_i^2,      sigma_i = |row_i|.
//      One-sided Jacobi finds every singular value to high relative accuracy, which is what the rcond = 1e-15
//      cut needs (the eigenvalues of the normal matrix would only resolve sigma_i/sigma_1 > 1e-8).
//   info (n_p x 4): [0] Jacobi sweeps used (negative: not converged after the maximum), [1] numerical rank kept,
//                   [2] sigma_max, [3] smallest singular value kept.
#include "common.hpp"

namespace {

constexpr int PV_THREADS = 256;
constexpr int PV_MAX_SWEEPS = 60;

template <int RMAX> struct PinvCfg {
  static constexpr int SC = RMAX >= 128 ? 16 : (RMAX >= 64 ? 64 : 128);   // panel rows of the streaming QR
  static constexpr int LDR = RMAX + 3;                                      // r + 2 columns, odd stride
};

template <int RMAX>
__global__ __launch_bounds__(PV_THREADS) void solve_pinv_kernel(
    const double *__restrict__ Theta, int s, int r, const double *__restrict__ cnt, const double *__restrict__ scale,
    int n_features, const double *__restrict__ y_all, double rcond, double *__restrict__ Ar,
    double *__restrict__ Ar_sigma, double *__restrict__ y0_all, double *__restrict__ info) {
  using C = PinvCfg<RMAX>;
  constexpr int SC = C::SC, LDR = C::LDR;
  __shared__ double Rm[RMAX * LDR];
  __shared__ double P[SC * LDR];
  __shared__ double sw[SC], sv[SC], ss[SC];
  __shared__ double hv[2];
  __shared__ double sig2[RMAX];
  __shared__ int flags[3];   // [0] any sigma != 0, [1] rotations in the current sweep, [2] a non-finite weight 1/sigma

  const int p = blockIdx.x;
  const double *y = y_all + (int64_t)p * s * 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nc = r + 2;

  if (tid < 3) flags[tid] = 0;
  __syncthreads();
  {
    int any = 0;
    for (int k = tid; k < s; k += PV_THREADS) any |= (y[3 * k + 1] != 0.0);
    if (any) flags[0] = 1;
  }
  for (int e = tid; e < RMAX * LDR; e += PV_THREADS) Rm[e] = 0.0;
  __syncthreads();
  const bool weighted = flags[0] != 0;

  // scale_vector for sensors [c0, c0 + rows): weights and scaled values into sw / sv / ss
  auto scale_rows = [&](int c0, int rows) {
    if (tid < rows) {
      const int k = c0 + tid;
      double w = 0.0, v0 = 0.0, s0 = 0.0;
      if (k < s) {
        int f = (int)y[3 * k + 2];
        f = f < 0 ? 0 : (f > n_features - 1 ? n_features - 1 : f);   // the host has already rejected ids out of range
        const double scl = scale[f];
        v0 = (y[3 * k] - cnt[k]) / scl;
        s0 = y[3 * k + 1] / scl;
        w = weighted ? 1.0 / s0 : 1.0;
        // an uncertainty that is zero (or NaN) for SOME sensors: W = diag(1/0) in the reference (:872), whose pinv raises
        if (!isfinite(w)) flags[2] = 1;
        if (y0_all) {
          y0_all[((int64_t)p * s + k) * 2] = v0;
          y0_all[((int64_t)p * s + k) * 2 + 1] = s0;
        }
      }
      sw[tid] = w; sv[tid] = v0; ss[tid] = s0;
    }
  };
  // rows [c0, c0 + rows) of M into dst (row stride LDR); rows past s are zero
  auto fill = [&](double *dst, int c0, int rows) {
    for (int e = tid; e < rows * nc; e += PV_THREADS) {
      const int kk = e / nc, c = e - kk * nc;
      const int k = c0 + kk;
      double val = 0.0;
      if (k < s) {
        if (c < r) val = sw[kk] * Theta[(int64_t)k * r + c];
        else if (c == r) val = sw[kk] * sv[kk];
        else val = weighted ? ss[kk] : 0.0;
      }
      dst[kk * LDR + c] = val;
    }
  };

  int q;   // rows of the factor
  if (s <= r) {
    q = s;
    for (int c0 = 0; c0 < s; c0 += SC) {
      const int rows = (s - c0 < SC) ? s - c0 : SC;
      __syncthreads();
      scale_rows(c0, rows);
      __syncthreads();
      fill(Rm + c0 * LDR, c0, rows);
    }
    __syncthreads();
  } else {
    q = r;
    for (int c0 = 0; c0 < s; c0 += SC) {
      __syncthreads();
      scale_rows(c0, SC);
      __syncthreads();
      fill(P, c0, SC);
      __syncthreads();
      // re-triangularise [R; P]: reflector j annihilates column j of the panel against R[j][j] (dlarfg/dlarf)
      for (int j = 0; j < r; ++j) {
        if (wave == 0) {
          double ssq = 0.0;
          for (int i = lane; i < SC; i += 64) { const double x = P[i * LDR + j]; ssq += x * x; }
          ssq = group_sum(ssq, 64);
          if (lane == 0) {
            const double alpha = Rm[j * LDR + j];
            double tau = 0.0, scal = 0.0;
            if (ssq > 0.0) {
              const double beta = -copysign(sqrt(alpha * alpha + ssq), alpha);
              tau = (beta - alpha) / beta;
              scal = 1.0 / (alpha - beta);
              Rm[j * LDR + j] = beta;
            }
            hv[0] = tau; hv[1] = scal;
          }
        }
        __syncthreads();
        const double tau = hv[0], scal = hv[1];
        if (tau != 0.0) {
          for (int k = j + 1 + tid; k < nc; k += PV_THREADS) {
            double dot = Rm[j * LDR + k];
            for (int i = 0; i < SC; ++i) dot += (P[i * LDR + j] * scal) * P[i * LDR + k];
            const double t = tau * dot;
            Rm[j * LDR + k] -= t;
            for (int i = 0; i < SC; ++i) P[i * LDR + k] -= t * (P[i * LDR + j] * scal);
          }
        }
        __syncthreads();
      }
    }
  }

  // ---- one-sided Jacobi on the q rows (length r; the two right-hand-side columns ride along) ----
  const int qe = q + (q & 1);
  const int npairs = qe / 2;
  int tpp = 64;
  while (tpp * npairs > PV_THREADS && tpp > 1) tpp >>= 1;
  const int pair = tid / tpp, lip = tid % tpp;
  int sweeps = 0;
  bool converged = (q <= 1);
  const double tol = 2.3e-16 * sqrt((double)r);   // rounding floor of the r-term inner product (dgesvj: sqrt(m) eps)
  while (!converged && sweeps < PV_MAX_SWEEPS) {
    if (tid == 0) flags[1] = 0;
    __syncthreads();
    for (int t = 0; t < qe - 1; ++t) {
      if (pair < npairs) {
        // round-robin tournament (circle method): slot 0 plays the fixed row qe-1
        int a, b;
        if (pair == 0) { a = qe - 1; b = t; }
        else { a = (t + pair) % (qe - 1); b = (t - pair + (qe - 1)) % (qe - 1); }
        if (a > b) { const int x = a; a = b; b = x; }
        if (b < q) {
          double *ra = Rm + a * LDR, *rb = Rm + b * LDR;
          double al = 0.0, be = 0.0, ga = 0.0;
          for (int c = lip; c < r; c += tpp) {
            const double xa = ra[c], xb = rb[c];
            al += xa * xa; be += xb * xb; ga += xa * xb;
          }
          for (int o = tpp >> 1; o > 0; o >>= 1) {
            al += __shfl_xor(al, o, 64); be += __shfl_xor(be, o, 64); ga += __shfl_xor(ga, o, 64);
          }
          if (al > 0.0 && be > 0.0 && fabs(ga) > tol * sqrt(al) * sqrt(be)) {
            const double zeta = (be - al) / (2.0 * ga);
            const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
            for (int c = lip; c < nc; c += tpp) {
              const double xa = ra[c], xb = rb[c];
              ra[c] = cs * xa - sn * xb;
              rb[c] = sn * xa + cs * xb;
            }
            if (lip == 0) flags[1] = 1;
          }
        }
      }
      __syncthreads();
    }
    ++sweeps;
    converged = (flags[1] == 0);
    __syncthreads();
  }

  // ---- x = sum_i row_i c'_i / sigma_i^2 over the singular values above rcond * sigma_max ----
  for (int i = wave; i < q; i += PV_THREADS / 64) {
    double a = 0.0;
    for (int c = lane; c < r; c += 64) { const double x = Rm[i * LDR + c]; a += x * x; }
    a = group_sum(a, 64);
    if (lane == 0) sig2[i] = a;
  }
  __syncthreads();
  double s2max = 0.0;
  for (int i = 0; i < q; ++i) s2max = sig2[i] > s2max ? sig2[i] : s2max;
  const double cut = rcond * sqrt(s2max);
  for (int c = tid; c < r; c += PV_THREADS) {
    double x0 = 0.0, x1 = 0.0;
    for (int i = 0; i < q; ++i) {
      const double sg2 = sig2[i];
      if (sqrt(sg2) > cut) {
        const double g = Rm[i * LDR + c] / sg2;
        x0 += g * Rm[i * LDR + r];
        x1 += g * Rm[i * LDR + r + 1];
      }
    }
    Ar[(int64_t)p * r + c] = x0;
    Ar_sigma[(int64_t)p * r + c] = weighted ? fabs(x1) : 0.0;
  }
  if (tid == 0) {
    int rank = 0;
    double smin = 0.0;
    for (int i = 0; i < q; ++i)
      if (sqrt(sig2[i]) > cut) { ++rank; smin = (rank == 1 || sig2[i] < smin) ? sig2[i] : smin; }
    info[4 * p] = (converged && !flags[2]) ? (double)sweeps : -(double)(sweeps > 0 ? sweeps : 1);   // < 0: LinAlgError on the host
    info[4 * p + 1] = (double)rank;
    info[4 * p + 2] = sqrt(s2max);
    info[4 * p + 3] = sqrt(smin);
  }
}

// ---- the same algorithm for r > 128 (r <= SPR_MAX_R_WIDE): the factor no longer fits LDS, so it lives in a global
// workspace (L2-resident: (r + 16) x (r + 3) doubles per measurement vector) and one 1024-thread workgroup per vector
// works on it; workgroup barriers order the global accesses (one CU, one L1).  A rarely taken path -- the reference
// accepts any r <= m (:336) -- written for correctness, not speed.
constexpr int PW_THREADS = 1024;
constexpr int PW_SC = 16;

__global__ __launch_bounds__(PW_THREADS) void solve_pinv_wide_kernel(
    const double *__restrict__ Theta, int s, int r, const double *__restrict__ cnt, const double *__restrict__ scale,
    int n_features, const double *__restrict__ y_all, double rcond, double *__restrict__ Ar,
    double *__restrict__ Ar_sigma, double *__restrict__ y0_all, double *__restrict__ info, double *ws, int64_t ws_stride) {
  constexpr int SC = PW_SC;
  const int LDR = r + 3;
  double *Rm = ws + (int64_t)blockIdx.x * ws_stride;         // r x LDR
  double *P = Rm + (int64_t)r * LDR;                          // SC x LDR
  __shared__ double sw[SC], sv[SC], ss[SC];
  __shared__ double hv[2];
  __shared__ double sig2[SPR_MAX_R_WIDE];
  __shared__ int flags[3];

  const int p = blockIdx.x;
  const double *y = y_all + (int64_t)p * s * 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nc = r + 2;

  if (tid < 3) flags[tid] = 0;
  __syncthreads();
  {
    int any = 0;
    for (int k = tid; k < s; k += PW_THREADS) any |= (y[3 * k + 1] != 0.0);
    if (any) flags[0] = 1;
  }
  for (int64_t e = tid; e < (int64_t)r * LDR; e += PW_THREADS) Rm[e] = 0.0;
  __syncthreads();
  const bool weighted = flags[0] != 0;

  auto scale_rows = [&](int c0, int rows) {
    if (tid < rows) {
      const int k = c0 + tid;
      double w = 0.0, v0 = 0.0, s0 = 0.0;
      if (k < s) {
        int f = (int)y[3 * k + 2];
        f = f < 0 ? 0 : (f > n_features - 1 ? n_features - 1 : f);
        const double scl = scale[f];
        v0 = (y[3 * k] - cnt[k]) / scl;
        s0 = y[3 * k + 1] / scl;
        w = weighted ? 1.0 / s0 : 1.0;
        // an uncertainty that is zero (or NaN) for SOME sensors: W = diag(1/0) in the reference (:872), whose pinv raises
        if (!isfinite(w)) flags[2] = 1;
        if (y0_all) {
          y0_all[((int64_t)p * s + k) * 2] = v0;
          y0_all[((int64_t)p * s + k) * 2 + 1] = s0;
        }
      }
      sw[tid] = w; sv[tid] = v0; ss[tid] = s0;
    }
  };
  auto fill = [&](double *dst, int c0, int rows) {
    for (int e = tid; e < rows * nc; e += PW_THREADS) {
      const int kk = e / nc, c = e - kk * nc;
      const int k = c0 + kk;
      double val = 0.0;
      if (k < s) {
        if (c < r) val = sw[kk] * Theta[(int64_t)k * r + c];
        else if (c == r) val = sw[kk] * sv[kk];
        else val = weighted ? ss[kk] : 0.0;
      }
      dst[(int64_t)kk * LDR + c] = val;
    }
  };

  int q;
  if (s <= r) {
    q = s;
    for (int c0 = 0; c0 < s; c0 += SC) {
      const int rows = (s - c0 < SC) ? s - c0 : SC;
      __syncthreads();
      scale_rows(c0, rows);
      __syncthreads();
      fill(Rm + (int64_t)c0 * LDR, c0, rows);
    }
    __syncthreads();
  } else {
    q = r;
    for (int c0 = 0; c0 < s; c0 += SC) {
      __syncthreads();
      scale_rows(c0, SC);
      __syncthreads();
      fill(P, c0, SC);
      __syncthreads();
      for (int j = 0; j < r; ++j) {
        if (wave == 0) {
          double ssq = 0.0;
          for (int i = lane; i < SC; i += 64) { const double x = P[(int64_t)i * LDR + j]; ssq += x * x; }
          ssq = group_sum(ssq, 64);
          if (lane == 0) {
            const double alpha = Rm[(int64_t)j * LDR + j];
            double tau = 0.0, scal = 0.0;
            if (ssq > 0.0) {
              const double beta = -copysign(sqrt(alpha * alpha + ssq), alpha);
              tau = (beta - alpha) / beta;
              scal = 1.0 / (alpha - beta);
              Rm[(int64_t)j * LDR + j] = beta;
            }
            hv[0] = tau; hv[1] = scal;
          }
        }
        __syncthreads();
        const double tau = hv[0], scal = hv[1];
        if (tau != 0.0) {
          for (int k = j + 1 + tid; k < nc; k += PW_THREADS) {
            double dot = Rm[(int64_t)j * LDR + k];
            for (int i = 0; i < SC; ++i) dot += (P[(int64_t)i * LDR + j] * scal) * P[(int64_t)i * LDR + k];
            const double t = tau * dot;
            Rm[(int64_t)j * LDR + k] -= t;
            for (int i = 0; i < SC; ++i) P[(int64_t)i * LDR + k] -= t * (P[(int64_t)i * LDR + j] * scal);
          }
        }
        __syncthreads();
      }
    }
  }

  const int qe = q + (q & 1);
  const int npairs = qe / 2;
  int tpp = 64;
  while (tpp * npairs > PW_THREADS && tpp > 1) tpp >>= 1;
  const int pair = tid / tpp, lip = tid % tpp;
  int sweeps = 0;
  bool converged = (q <= 1);
  const double tol = 2.3e-16 * sqrt((double)r);
  while (!converged && sweeps < PV_MAX_SWEEPS) {
    if (tid == 0) flags[1] = 0;
    __syncthreads();
    for (int t = 0; t < qe - 1; ++t) {
      if (pair < npairs) {
        int a, b;
        if (pair == 0) { a = qe - 1; b = t; }
        else { a = (t + pair) % (qe - 1); b = (t - pair + (qe - 1)) % (qe - 1); }
        if (a > b) { const int x = a; a = b; b = x; }
        if (b < q) {
          double *ra = Rm + (int64_t)a * LDR, *rb = Rm + (int64_t)b * LDR;
          double al = 0.0, be = 0.0, ga = 0.0;
          for (int c = lip; c < r; c += tpp) {
            const double xa = ra[c], xb = rb[c];
            al += xa * xa; be += xb * xb; ga += xa * xb;
          }
          for (int o = tpp >> 1; o > 0; o >>= 1) {
            al += __shfl_xor(al, o, 64); be += __shfl_xor(be, o, 64); ga += __shfl_xor(ga, o, 64);
          }
          if (al > 0.0 && be > 0.0 && fabs(ga) > tol * sqrt(al) * sqrt(be)) {
            const double zeta = (be - al) / (2.0 * ga);
            const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
            for (int c = lip; c < nc; c += tpp) {
              const double xa = ra[c], xb = rb[c];
              ra[c] = cs * xa - sn * xb;
              rb[c] = sn * xa + cs * xb;
            }
            if (lip == 0) flags[1] = 1;
          }
        }
      }
      __syncthreads();
    }
    ++sweeps;
    converged = (flags[1] == 0);
    __syncthreads();
  }

  for (int i = wave; i < q; i += PW_THREADS / 64) {
    double a = 0.0;
    for (int c = lane; c < r; c += 64) { const double x = Rm[(int64_t)i * LDR + c]; a += x * x; }
    a = group_sum(a, 64);
    if (lane == 0) sig2[i] = a;
  }
  __syncthreads();
  double s2max = 0.0;
  for (int i = 0; i < q; ++i) s2max = sig2[i] > s2max ? sig2[i] : s2max;
  const double cut = rcond * sqrt(s2max);
  for (int c = tid; c < r; c += PW_THREADS) {
    double x0 = 0.0, x1 = 0.0;
    for (int i = 0; i < q; ++i) {
      const double sg2 = sig2[i];
      if (sqrt(sg2) > cut) {
        const double g = Rm[(int64_t)i * LDR + c] / sg2;
        x0 += g * Rm[(int64_t)i * LDR + r];
        x1 += g * Rm[(int64_t)i * LDR + r + 1];
      }
    }
    Ar[(int64_t)p * r + c] = x0;
    Ar_sigma[(int64_t)p * r + c] = weighted ? fabs(x1) : 0.0;
  }
  if (tid == 0) {
    int rank = 0;
    double smin = 0.0;
    for (int i = 0; i < q; ++i)
      if (sqrt(sig2[i]) > cut) { ++rank; smin = (rank == 1 || sig2[i] < smin) ? sig2[i] : smin; }
    info[4 * p] = (converged && !flags[2]) ? (double)sweeps : -(double)(sweeps > 0 ? sweeps : 1);   // < 0: LinAlgError on the host
    info[4 * p + 1] = (double)rank;
    info[4 * p + 2] = sqrt(s2max);
    info[4 * p + 3] = sqrt(smin);
  }
}

}  // namespace

extern "C" int spr_solve_pinv_f64(const double *d_Theta, int32_t s, int32_t r, const double *d_cnt, int32_t s_cnt,
                                  const double *d_scale, int32_t n_features, const double *d_y, int32_t n_p,
                                  double rcond, double *d_Ar, double *d_Ar_sigma, double *d_y0, double *d_info,
                                  void *stream) {
  SPR_REQUIRE(d_Theta && d_cnt && d_scale && d_y && d_Ar && d_Ar_sigma && d_info, SPR_E_INVALID,
              "spr_solve_pinv_f64: NULL pointer");
  SPR_REQUIRE(s > 0 && r > 0 && n_p > 0 && n_features > 0, SPR_E_INVALID, "spr_solve_pinv_f64: bad shape");
  SPR_REQUIRE(s_cnt == s, SPR_E_INVALID, "spr_solve_pinv_f64: cnt holds %d entries, Theta has %d rows", s_cnt, s);
  SPR_REQUIRE(rcond >= 0.0, SPR_E_INVALID, "spr_solve_pinv_f64: rcond < 0");
  SPR_REQUIRE(r <= SPR_MAX_R, SPR_E_UNSUPPORTED, "spr_solve_pinv_f64: r=%d > %d not built", r, SPR_MAX_R);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define PV(RM)                                                                                                  \
  hipLaunchKernelGGL(solve_pinv_kernel<RM>, dim3(n_p), dim3(PV_THREADS), 0, st, d_Theta, (int)s, (int)r, d_cnt, \
                     d_scale, (int)n_features, d_y, rcond, d_Ar, d_Ar_sigma, d_y0, d_info)
  if (r <= 16) PV(16);
  else if (r <= 32) PV(32);
  else if (r <= 64) PV(64);
  else PV(128);
#undef PV
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

// r > SPR_MAX_R (up to SPR_MAX_R_WIDE): same semantics, the factor in a caller-provided workspace of
// spr_solve_pinv_workspace(r, n_p) bytes (0 when r <= SPR_MAX_R: spr_solve_pinv_f64 needs none)
extern "C" size_t spr_solve_pinv_workspace(int32_t r, int32_t n_p) {
  if (r <= SPR_MAX_R || r > SPR_MAX_R_WIDE || n_p <= 0) return 0;
  return sizeof(double) * (size_t)n_p * (size_t)(r + PW_SC) * (size_t)(r + 3);
}

extern "C" int spr_solve_pinv_wide_f64(const double *d_Theta, int32_t s, int32_t r, const double *d_cnt, int32_t s_cnt,
                                       const double *d_scale, int32_t n_features, const double *d_y, int32_t n_p,
                                       double rcond, double *d_Ar, double *d_Ar_sigma, double *d_y0, double *d_info,
                                       void *d_workspace, size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(d_Theta && d_cnt && d_scale && d_y && d_Ar && d_Ar_sigma && d_info && d_workspace, SPR_E_INVALID,
              "spr_solve_pinv_wide_f64: NULL pointer");
  SPR_REQUIRE(s > 0 && r > 0 && n_p > 0 && n_features > 0, SPR_E_INVALID, "spr_solve_pinv_wide_f64: bad shape");
  SPR_REQUIRE(s_cnt == s, SPR_E_INVALID, "spr_solve_pinv_wide_f64: cnt holds %d entries, Theta has %d rows", s_cnt, s);
  SPR_REQUIRE(rcond >= 0.0, SPR_E_INVALID, "spr_solve_pinv_wide_f64: rcond < 0");
  SPR_REQUIRE(r <= SPR_MAX_R_WIDE, SPR_E_UNSUPPORTED, "spr_solve_pinv_wide_f64: r=%d > %d not built", r, SPR_MAX_R_WIDE);
  const size_t per = (size_t)(r + PW_SC) * (size_t)(r + 3);
  SPR_REQUIRE(workspace_bytes >= sizeof(double) * per * (size_t)n_p, SPR_E_WORKSPACE,
              "spr_solve_pinv_wide_f64: workspace too small");
  hipLaunchKernelGGL(solve_pinv_wide_kernel, dim3(n_p), dim3(PW_THREADS), 0, static_cast<hipStream_t>(stream), d_Theta,
                     (int)s, (int)r, d_cnt, d_scale, (int)n_features, d_y, rcond, d_Ar, d_Ar_sigma, d_y0, d_info,
                     static_cast<double *>(d_workspace), (int64_t)per);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}
