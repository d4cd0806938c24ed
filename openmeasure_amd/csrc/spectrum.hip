// Device-side spectrum of the scaled Gram matrix for small snapshot counts (m <= 64).
//
// The reference's thin SVD (sparse_sensing.py:272) reduces, on the Gram route, to the m x m
// symmetric eigen-problem  G = sum_f G_f / scl_f^2 = V diag(S^2) V^T.  For the snapshot counts the
// reference is actually used with (its own data set has m = 41) the host LAPACK call plus the
// device->host->device round trip around it costs more than all the kernels of fit() together.
// This kernel keeps the whole step on the device so that fit() issues no host synchronisation.
// Measured on MI355X: 0.07 ms at m = 12, 0.39 ms at m = 41, 0.98 ms at m = 64 (11-14 sweeps of m-1
// rounds, two workgroup barriers each: latency-bound, ~1.1 us per round) against ~0.3-0.5 ms for LAPACK dsyevd on
// the host plus its round trip: fit() is faster through this kernel up to m = 24-28 (tools/spectrum_crossover.py),
// so the Python layer takes this path for m <= 24 and keeps dsyevd above that:
//   1. Chan-merge the per-rank feature statistics (rank order), block variance from trace(G_f) and
//      M2, the per-feature scale of the chosen scale_type (:114-161);
//   2. G = sum_f G_f / scl_f^2 in LDS;
//   3. cyclic two-sided Jacobi with the round-robin pairing (m/2 independent rotations per round,
//      one 1024-thread workgroup, G and V both in LDS), until off(G)^2 <= 1e-31 ||diag||^2;
//   4. eigenvalues sorted descending, eigenvector signs fixed (largest-magnitude entry positive,
//      the convention of the host path), S, explained variance, W = V_r S_r^-1, A_r = V_r S_r.
// Deterministic (fixed pairing, fixed reduction order), so every rank of a sharded run computes
// bit-identical factors from the all-reduced Gram blocks and no broadcast is needed.
#include "common.hpp"

namespace {

constexpr int SP_MAXM = 64;
constexpr int SP_THREADS = 1024;
constexpr int SP_MAX_SWEEPS = 15;

// scale_type codes shared with openmeasure_amd/rom.py / engine.py
enum { SC_STD = 0, SC_NONE = 1, SC_PARETO = 2, SC_VAST = 3, SC_LEVEL = 4, SC_VARIANCE = 5, SC_POISSON = 6, SC_L2 = 7 };

__device__ inline double block_sum_all(double v, double *red) {
  v = group_sum_t<64>(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < SP_THREADS / 64; ++w) s += red[w];
  return s;
}

// pair k of round rd in the round-robin tournament over mp (even) indices, p < q
__device__ inline void pair_of(int k, int rd, int mp, int &p, int &q) {
  int a, b;
  if (k == 0) { a = mp - 1; b = rd % (mp - 1); }
  else { a = (rd + k) % (mp - 1); b = (rd - k + (mp - 1)) % (mp - 1); }
  p = a < b ? a : b;
  q = a < b ? b : a;
}

// 1/x and 1/sqrt(x) from the hardware approximations (v_rcp_f64 / v_rsq_f64) plus Newton steps: full double
// accuracy without the long division / square-root sequences
__device__ inline double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}
__device__ inline double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}

// Jacobi rotation that annihilates a_pq:  J = [[c, s], [-s, c]],  t = sign(tau) / (|tau| + sqrt(1 + tau^2)).
// With u = 2 a_pq and w = a_qq - a_pp (tau = w / u):  t = sign(w u) |u| / (|w| + sqrt(w^2 + u^2)) -- no division
// by the possibly tiny a_pq.
__device__ inline void rotation(double app, double aqq, double apq, double &c, double &s) {
  c = 1.0; s = 0.0;
  if (fabs(apq) > 1e-300) {
    const double u = 2.0 * apq, w = aqq - app;
    const double h2 = fma(w, w, u * u);
    const double h = h2 * fast_rsqrt(h2);                   // sqrt(w^2 + u^2)
    double t = fabs(u) * fast_rcp(fabs(w) + h);
    t = ((w >= 0.0) == (u >= 0.0)) ? t : -t;
    c = fast_rsqrt(fma(t, t, 1.0));
    s = t * c;
  }
}

__global__ __launch_bounds__(SP_THREADS) void spectrum_kernel(
    const double *__restrict__ gram, const double *__restrict__ fstats_all, int n_ranks, int n_features, int m,
    int scale_code, int r, double *__restrict__ feat_out /*[F][5]: cnt, mu, var, scl, fluctuation variance*/,
    double *__restrict__ scale, double *__restrict__ inv_scale, double *__restrict__ lam_out,
    double *__restrict__ S_out, double *__restrict__ expvar_out, double *__restrict__ V_out,
    double *__restrict__ W_out, double *__restrict__ Ar_out, double *__restrict__ info) {
  constexpr int LD = SP_MAXM + 1;
  __shared__ double A[SP_MAXM * LD];
  __shared__ double V[SP_MAXM * LD];
  __shared__ double cs[SP_MAXM / 2], sn[SP_MAXM / 2];
  __shared__ int pp[SP_MAXM / 2], qq[SP_MAXM / 2];
  __shared__ double lam[SP_MAXM], red[SP_THREADS / 64];
  __shared__ double inv2[64];           // 1/scl_f^2 (chunks of 64 features)
  __shared__ int order[SP_MAXM];
  __shared__ double sgn[SP_MAXM];
  const int tid = threadIdx.x;
  const int mp = m + (m & 1);           // even size for the round-robin pairing (extra row/col is zero)

  // ---- 1 + 2: feature scales and the scaled Gram matrix ------------------------------------
  for (int e = tid; e < SP_MAXM * LD; e += SP_THREADS) { A[e] = 0.0; V[e] = 0.0; }
  __syncthreads();
  for (int f0 = 0; f0 < n_features; f0 += 64) {
    const int f = f0 + tid;
    if (tid < 64 && f < n_features) {
      double n = 0.0, mu = 0.0, m2 = 0.0;
      for (int w = 0; w < n_ranks; ++w) {   // Chan merge in rank order
        const double *q = fstats_all + ((int64_t)w * n_features + f) * 3;
        const double nb = q[0], mb = q[1], sb = q[2];
        if (nb > 0.0) {
          const double tot = n + nb, d = mb - mu;
          mu += d * nb / tot;
          m2 += sb + d * d * n * nb / tot;
          n = tot;
        }
      }
      double tr = 0.0;
      const double *G = gram + (int64_t)f * m * m;
      for (int i = 0; i < m; ++i) tr += G[(int64_t)i * m + i];
      // population variance of the raw block (:115); a feature without rows (partial row group) takes no part
      const bool present = n > 0.0;
      const double var = present ? (tr + m * m2) / (n * m) : 0.0;
      const double sd = sqrt(var);
      double scl;
      switch (scale_code) {
        case SC_NONE: scl = 1.0; break;
        case SC_PARETO: scl = sqrt(sd); break;
        case SC_VAST: scl = var / mu; break;
        case SC_LEVEL: scl = mu; break;
        case SC_VARIANCE: scl = var; break;
        case SC_POISSON: scl = sqrt(mu); break;
        case SC_L2: scl = sqrt(n * m * (var + mu * mu)); break;
        default: scl = sd; break;
      }
      if (!present) scl = 1.0;
      feat_out[5 * f] = n; feat_out[5 * f + 1] = mu; feat_out[5 * f + 2] = var;
      feat_out[5 * f + 3] = scl; feat_out[5 * f + 4] = present ? tr / (n * m) : 0.0;   // variance of the row-centred values (combine.hip)
      scale[f] = scl;
      inv_scale[f] = 1.0 / scl;
      inv2[tid] = 1.0 / (scl * scl);
    }
    __syncthreads();
    const int nf = (n_features - f0 < 64) ? n_features - f0 : 64;
    for (int e = tid; e < m * m; e += SP_THREADS) {
      const int i = e / m, j = e - i * m;
      double acc = A[i * LD + j];
      for (int ff = 0; ff < nf; ++ff) acc += gram[((int64_t)(f0 + ff) * m + i) * m + j] * inv2[ff];
      A[i * LD + j] = acc;
    }
    __syncthreads();
  }
  for (int i = tid; i < m; i += SP_THREADS) V[i * LD + i] = 1.0;
  __syncthreads();

  // ---- 3: cyclic Jacobi, round-robin pairing, two short phases per round ------------------------
  // The mp/2 rotations of a round act on disjoint index pairs, so A' = J^T A J splits into (mp/2)^2 independent
  // 2x2 blocks: block (a, b) = rows {p_a, q_a} x columns {p_b, q_b} needs only its own four entries and the two
  // rotations (c_a, s_a), (c_b, s_b).  Phase A: P lanes of wave 0 compute the rotations (reciprocal / reciprocal
  // square root instructions + Newton steps instead of the division and square-root sequences: this chain is the
  // critical path of a round).  Phase B: thread t = a P + b updates its block of A and two rows of V in place.
  const int P = mp / 2;
  const int ba = tid / P, bb = tid - ba * P;               // this thread's block (a, b); idle when tid >= P * P
  const bool active = tid < P * P;
  int sweeps = 0;
  double off2 = 0.0, diag2 = 0.0;
  for (; sweeps < SP_MAX_SWEEPS; ++sweeps) {
    double o = 0.0, d = 0.0;
    for (int e = tid; e < m * m; e += SP_THREADS) {
      const int i = e / m, j = e - i * m;
      const double a = A[i * LD + j];
      if (i == j) d += a * a; else o += a * a;
    }
    off2 = block_sum_all(o, red);
    diag2 = block_sum_all(d, red);
    if (off2 <= 1e-31 * diag2) break;
    for (int rd = 0; rd < mp - 1; ++rd) {
      if (tid < P) {
        int p, q;
        pair_of(tid, rd, mp, p, q);
        double c, sv;
        rotation(A[p * LD + p], A[q * LD + q], A[p * LD + q], c, sv);
        cs[tid] = c; sn[tid] = sv; pp[tid] = p; qq[tid] = q;
      }
      __syncthreads();
      if (active) {
        const int pa = pp[ba], qa = qq[ba], pb = pp[bb], qb = qq[bb];
        const double ca = cs[ba], sa = sn[ba], cb = cs[bb], sb = sn[bb];
        // B = [[x00, x01], [x10, x11]] on rows (pa, qa), columns (pb, qb);  B' = Ja^T B Jb,  J = [[c, s], [-s, c]]
        const double x00 = A[pa * LD + pb], x01 = A[pa * LD + qb], x10 = A[qa * LD + pb], x11 = A[qa * LD + qb];
        const double y00 = ca * x00 - sa * x10, y01 = ca * x01 - sa * x11;     // rows: J^T from the left
        const double y10 = sa * x00 + ca * x10, y11 = sa * x01 + ca * x11;
        A[pa * LD + pb] = cb * y00 - sb * y01;  A[pa * LD + qb] = sb * y00 + cb * y01;   // columns: J from the right
        A[qa * LD + pb] = cb * y10 - sb * y11;  A[qa * LD + qb] = sb * y10 + cb * y11;
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                            // V <- V Jb on rows 2a, 2a+1
          const int i = 2 * ba + h;
          const double vp = V[i * LD + pb], vq = V[i * LD + qb];
          V[i * LD + pb] = cb * vp - sb * vq;
          V[i * LD + qb] = sb * vp + cb * vq;
        }
      }
      __syncthreads();
    }
  }

  // ---- 4: sort, sign convention, derived quantities -----------------------------------------
  for (int i = tid; i < m; i += SP_THREADS) lam[i] = A[i * LD + i];
  __syncthreads();
  if (tid < m) {
    int rank = 0;
    const double li = lam[tid];
    for (int j = 0; j < m; ++j) rank += (lam[j] > li) || (lam[j] == li && j < tid);
    order[rank] = tid;
  }
  __syncthreads();
  if (tid < m) {                              // sign of sorted column tid: its largest-magnitude entry positive
    const int col = order[tid];
    double best = -1.0, val = 1.0;
    for (int i = 0; i < m; ++i) {
      const double v = V[i * LD + col];
      if (fabs(v) > best) { best = fabs(v); val = v; }
    }
    sgn[tid] = val < 0.0 ? -1.0 : 1.0;
  }
  __syncthreads();
  const double l0 = lam[order[0]] > 0.0 ? lam[order[0]] : 0.0;
  const double s0 = sqrt(l0);
  double floor_s = s0 * sqrt((double)m * 2.220446049250313e-16);
  if (!(floor_s > 0.0)) floor_s = 1.0;
  if (tid < m) {
    const double l = lam[order[tid]];
    lam_out[tid] = l;
    S_out[tid] = sqrt(l > 0.0 ? l : 0.0);
  }
  if (tid == 0) {
    double tot = 0.0;
    for (int k = 0; k < m; ++k) { const double l = lam[order[k]]; tot += l > 0.0 ? l : 0.0; }
    double run = 0.0;
    for (int k = 0; k < m; ++k) {
      const double l = lam[order[k]];
      run += l > 0.0 ? l : 0.0;
      expvar_out[k] = 100.0 * run / tot;      // :274-275
    }
    info[0] = (double)sweeps; info[1] = off2; info[2] = diag2;
  }
  for (int e = tid; e < m * m; e += SP_THREADS) {
    const int i = e / m, k = e - i * m;
    const double v = V[i * LD + order[k]] * sgn[k];
    V_out[e] = v;
    if (k < r) {
      const double l = lam[order[k]];
      const double sk = sqrt(l > 0.0 ? l : 0.0);
      W_out[i * r + k] = v / (sk > floor_s ? sk : floor_s);
      Ar_out[i * r + k] = v * sk;             // A = (diag(S) Vt)^T  (:273)
    }
  }
}

}  // namespace

extern "C" int32_t spr_spectrum_max_m(void) { return SP_MAXM; }

extern "C" int spr_spectrum_f64(const double *d_gram, const double *d_fstats_all, int32_t n_ranks,
                                int32_t n_features, int32_t m, int32_t scale_code, int32_t r, double *d_feat,
                                double *d_scale, double *d_inv_scale, double *d_lam, double *d_S, double *d_expvar,
                                double *d_V, double *d_W, double *d_Ar, double *d_info, void *stream) {
  SPR_REQUIRE(d_gram && d_fstats_all && d_feat && d_scale && d_inv_scale && d_lam && d_S && d_expvar && d_V && d_W &&
                  d_Ar && d_info,
              SPR_E_INVALID, "spr_spectrum_f64: NULL pointer");
  SPR_REQUIRE(m >= 1 && n_features >= 1 && n_ranks >= 1 && r >= 1 && r <= m, SPR_E_INVALID,
              "spr_spectrum_f64: bad shape m=%d F=%d ranks=%d r=%d", m, n_features, n_ranks, r);
  SPR_REQUIRE(m <= SP_MAXM, SPR_E_UNSUPPORTED, "spr_spectrum_f64: m=%d > %d (use the host eigen-solver)", m, SP_MAXM);
  SPR_REQUIRE(scale_code >= 0 && scale_code <= SC_L2, SPR_E_UNSUPPORTED, "spr_spectrum_f64: scale code %d", scale_code);
  hipLaunchKernelGGL(spectrum_kernel, dim3(1), dim3(SP_THREADS), 0, static_cast<hipStream_t>(stream), d_gram,
                     d_fstats_all, (int)n_ranks, (int)n_features, (int)m, (int)scale_code, (int)r, d_feat, d_scale,
                     d_inv_scale, d_lam, d_S, d_expvar, d_V, d_W, d_Ar, d_info);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}
