// Host-side helper of fit(): eigenvectors of a symmetric tridiagonal matrix for a LIST of eigenvalues, all at once.
//
// fit() eigen-solves the m x m Gram matrix on the host (the reference's np.linalg.svd call site, sparse_sensing.py:272; only
// the r vectors reduction() keeps are needed, :336): dsytrd -> dsterf -> vectors of the r largest eigenvalues -> dormqr.  LAPACK's
// dstein finds those vectors one eigenvalue after the other, each by a few scalar recurrences of length m (factor T - lambda I
// with partial pivoting, forward / backward solve): 0.77 ms for 64 of 256 on the GPU host, latency-bound.  The recurrences
// of DIFFERENT eigenvalues are independent, so this routine runs them side by side: every array carries the eigenvalue index
// as its fastest dimension and the compiler vectorises over it (AVX-512: 8 eigenvalues per instruction).  Same algorithm as
// dlagtf / dlagts(job = -1) / dstein's iteration, WITHOUT dstein's re-orthogonalisation inside clusters of close eigenvalues:
// the caller checks the result for orthonormality and takes dstein when that fails (openmeasure_amd/_eigen.py,
// _eigvecs_top).  Plain host code: no device memory, no stream.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace {

// counter-based uniform(-1, 1): the start vectors must not depend on the thread or the call
inline double start_value(uint32_t i, uint32_t j) {
  uint32_t x = i * 0x9E3779B1u + j * 0x85EBCA77u + 0x165667B1u;
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return (double)x * (2.0 / 4294967296.0) - 1.0;
}

}  // namespace

// The loops over the eigenvalue index are what the compiler vectorises: one clone per vector ISA, picked at load time (the
// library is built without -march, and the GPU hosts have AVX-512).
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define SPR_HOST_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define SPR_HOST_CLONES
#endif

namespace {
// work: 5 m r + 3 r doubles
SPR_HOST_CLONES void tridiag_vectors_core(const double *h_d, const double *h_e, int m, const double *h_lam, int r, double *h_Z,
                                          int iterations, double *work) {
  const size_t mr = (size_t)m * r;
  // U = (a: diagonal, b: first superdiagonal, u2: second superdiagonal), l: multipliers, sw: 1.0 where rows k, k+1 were swapped
  double *a = work, *b = a + mr, *u2 = b + mr, *l = u2 + mr, *sw = l + mr;
  double *scale1 = sw + mr, *nrm = scale1 + r, *tmp = nrm + r;

  double tnorm = fabs(h_d[0]) + fabs(h_e[0]);                     // ||T||_1
  for (int i = 1; i < m; ++i) {
    const double v = fabs(h_d[i]) + fabs(h_e[i - 1]) + (i < m - 1 ? fabs(h_e[i]) : 0.0);
    tnorm = v > tnorm ? v : tnorm;
  }
  const double eps = 2.220446049250313e-16;
  const double tol = (tnorm > 0.0 ? tnorm : 1.0) * eps;             // smallest pivot the back substitution divides by

  // ---- dlagtf, batched: T - lam_j I = P L U -------------------------------------------------------------------
  for (int j = 0; j < r; ++j) {
    a[j] = h_d[0] - h_lam[j];
    scale1[j] = fabs(a[j]) + fabs(h_e[0]);
  }
  for (int k = 0; k < m - 1; ++k) {
    double *ak = a + (size_t)k * r, *ak1 = a + (size_t)(k + 1) * r, *bk = b + (size_t)k * r, *bk1 = b + (size_t)(k + 1) * r;
    double *uk = u2 + (size_t)k * r, *lk = l + (size_t)k * r, *sk = sw + (size_t)k * r;
    const double ck = h_e[k];                                       // sub-diagonal entry (k+1, k)
    const double ek = h_e[k];                                       // super-diagonal entry (k, k+1)
    const double ek1 = (k < m - 2) ? h_e[k + 1] : 0.0;              // super-diagonal entry (k+1, k+2)
    const double dk1 = h_d[k + 1];
#pragma clang loop vectorize(enable)
    for (int j = 0; j < r; ++j) {
      const double akk = ak[j];
      const double bkk = (k == 0) ? ek : bk[j];                     // row k's first superdiagonal as left by the previous step
      const double a1 = dk1 - h_lam[j];
      const double scale2 = fabs(ck) + fabs(a1) + fabs(ek1);
      const double piv1 = (akk == 0.0) ? 0.0 : fabs(akk) / scale1[j];
      const double piv2 = (ck == 0.0) ? 0.0 : fabs(ck) / scale2;
      const bool swap = (ck != 0.0) && (piv2 > piv1);
      // no interchange: l = c / a(k), a(k+1) = a1 - l b(k), row k keeps (a, b, 0), row k+1 keeps its own superdiagonal
      const double l_n = (ck == 0.0 || akk == 0.0) ? 0.0 : ck / akk;
      const double a1_n = a1 - l_n * bkk;
      // interchange: row k <- old row k+1 = (c, a1, e(k+1)); l = a(k) / c; row k+1 <- old row k - l * old row k+1
      const double l_s = (ck == 0.0) ? 0.0 : akk / ck;
      ak[j] = swap ? ck : akk;
      bk[j] = swap ? a1 : bkk;
      uk[j] = swap ? ek1 : 0.0;
      ak1[j] = swap ? bkk - l_s * a1 : a1_n;
      bk1[j] = swap ? -l_s * ek1 : ek1;
      lk[j] = swap ? l_s : l_n;
      sk[j] = swap ? 1.0 : 0.0;
      scale1[j] = swap ? scale1[j] : scale2;
    }
  }
  {
    double *uk = u2 + (size_t)(m - 1) * r, *bk = b + (size_t)(m - 1) * r;
    for (int j = 0; j < r; ++j) { uk[j] = 0.0; bk[j] = 0.0; }
    if (m >= 2) {
      double *u2m = u2 + (size_t)(m - 2) * r;
      for (int j = 0; j < r; ++j) u2m[j] = 0.0;                     // row m-2 has no second superdiagonal
    }
  }

  // ---- inverse iteration: (T - lam_j I) x <- x, normalised, a fixed number of times ------------------------------
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < r; ++j) h_Z[(size_t)i * r + j] = start_value((uint32_t)i, (uint32_t)j);
  for (int it = 0; it < iterations; ++it) {
    // forward elimination with the recorded interchanges (dlagts, job = -1)
    for (int k = 1; k < m; ++k) {
      double *yk = h_Z + (size_t)k * r, *yk1 = h_Z + (size_t)(k - 1) * r;
      const double *lk = l + (size_t)(k - 1) * r, *sk = sw + (size_t)(k - 1) * r;
#pragma clang loop vectorize(enable)
      for (int j = 0; j < r; ++j) {
        const double y0 = yk1[j], y1 = yk[j];
        const bool s = sk[j] != 0.0;
        yk1[j] = s ? y1 : y0;
        yk[j] = s ? y0 - lk[j] * y1 : y1 - lk[j] * y0;
      }
    }
    // back substitution; a pivot below tol is replaced by +-tol (dlagts' perturbation)
    for (int k = m - 1; k >= 0; --k) {
      double *yk = h_Z + (size_t)k * r;
      const double *ak = a + (size_t)k * r, *bk = b + (size_t)k * r, *uk = u2 + (size_t)k * r;
      const double *y1 = (k + 1 < m) ? h_Z + (size_t)(k + 1) * r : nullptr;
      const double *y2 = (k + 2 < m) ? h_Z + (size_t)(k + 2) * r : nullptr;
#pragma clang loop vectorize(enable)
      for (int j = 0; j < r; ++j) {
        double t = yk[j];
        if (y1) t -= bk[j] * y1[j];
        if (y2) t -= uk[j] * y2[j];
        double piv = ak[j];
        piv = (fabs(piv) < tol) ? (piv < 0.0 ? -tol : tol) : piv;
        yk[j] = t / piv;
      }
    }
    // normalise every column (also keeps the next solve far from overflow)
    for (int j = 0; j < r; ++j) nrm[j] = 0.0;
    for (int i = 0; i < m; ++i) {
      const double *zi = h_Z + (size_t)i * r;
#pragma clang loop vectorize(enable)
      for (int j = 0; j < r; ++j) { const double v = fabs(zi[j]); nrm[j] = v > nrm[j] ? v : nrm[j]; }
    }
    for (int j = 0; j < r; ++j) tmp[j] = (nrm[j] > 0.0 && isfinite(nrm[j])) ? 1.0 / nrm[j] : 0.0;
    for (int j = 0; j < r; ++j) nrm[j] = 0.0;
    for (int i = 0; i < m; ++i) {
      double *zi = h_Z + (size_t)i * r;
#pragma clang loop vectorize(enable)
      for (int j = 0; j < r; ++j) { zi[j] *= tmp[j]; nrm[j] += zi[j] * zi[j]; }
    }
    for (int j = 0; j < r; ++j) tmp[j] = (nrm[j] > 0.0) ? 1.0 / sqrt(nrm[j]) : 0.0;
    for (int i = 0; i < m; ++i) {
      double *zi = h_Z + (size_t)i * r;
#pragma clang loop vectorize(enable)
      for (int j = 0; j < r; ++j) zi[j] *= tmp[j];
    }
  }
}
}  // namespace

// d[m], e[m-1]: the tridiagonal matrix; lam[r]: eigenvalues (any order); Z: m x r row-major, column j = unit eigenvector of lam[j].
// iterations: inverse-iteration steps (dstein stops two steps after the growth criterion; 4 fixed steps cover that for
// eigenvalues separated by more than ~1e5 eps ||T||, which the caller's check enforces after the fact).
extern "C" int spr_host_tridiag_vectors(const double *h_d, const double *h_e, int32_t m, const double *h_lam, int32_t r,
                                        double *h_Z, int32_t iterations) {
  SPR_REQUIRE(h_d && h_lam && h_Z && (h_e || m == 1), SPR_E_INVALID, "spr_host_tridiag_vectors: NULL pointer");
  SPR_REQUIRE(m >= 1 && r >= 1 && r <= m && iterations >= 1 && iterations <= 16, SPR_E_INVALID,
              "spr_host_tridiag_vectors: bad shape m=%d r=%d iterations=%d", m, r, iterations);
  if (m == 1) {
    for (int j = 0; j < r; ++j) h_Z[j] = 1.0;
    return SPR_OK;
  }
  double *work = static_cast<double *>(malloc(sizeof(double) * (5 * (size_t)m * r + 3 * (size_t)r)));
  SPR_REQUIRE(work != nullptr, SPR_E_WORKSPACE, "spr_host_tridiag_vectors: out of host memory");
  tridiag_vectors_core(h_d, h_e, (int)m, h_lam, (int)r, h_Z, (int)iterations, work);
  free(work);
  return SPR_OK;
}


// ---- the whole top-r route in ONE host call ---------------------------------------------------------------------------------
// fit()'s host gap at small m is mostly call overhead around four short LAPACK routines (m = 64, r = 32: dsytrd 33 us, dsterf 24,
// the batched vectors 27, the back-transformation 18 -- and 140-207 us once Python has glued them together, against 173 us for
// dsyevd; tools/eigvec_pieces_probe.py).  The caller hands over the addresses of its LAPACK's dsytrd / dsterf / dormtr (SciPy
// exports them as C function pointers, scipy.linalg.cython_lapack.__pyx_capi__: the library itself links no LAPACK) and gets
// all eigenvalues and the r leading eigenvectors back.  Same checks as the Python route (_eigvecs_top): orthonormality of the
// tridiagonal vectors to 1e-8 (then one symmetric correction), of the result to 1e-12; status 2 tells the caller to take
// dstein / dsyevd instead.
namespace {
typedef void (*dsytrd_fn)(char *, int *, double *, int *, double *, double *, double *, double *, int *, int *);
typedef void (*dsterf_fn)(int *, double *, double *, int *);
typedef void (*dormtr_fn)(char *, char *, char *, int *, int *, double *, int *, double *, double *, int *, double *, int *, int *);

// max |A^T A - I| for the m x r row-major A; optionally E = A^T A - I (r x r)
SPR_HOST_CLONES double gram_defect(const double *A, int m, int r, double *E) {
  for (int i = 0; i < r * r; ++i) E[i] = 0.0;
  for (int k = 0; k < m; ++k) {
    const double *ak = A + (size_t)k * r;
    for (int i = 0; i < r; ++i) {
      const double aki = ak[i];
      double *ei = E + (size_t)i * r;
#pragma clang loop vectorize(enable)
      for (int j = 0; j < r; ++j) ei[j] += aki * ak[j];
    }
  }
  double worst = 0.0;
  for (int i = 0; i < r; ++i) {
    E[(size_t)i * r + i] -= 1.0;
    for (int j = 0; j < r; ++j) {
      const double v = fabs(E[(size_t)i * r + j]);
      if (!(v <= worst)) worst = v;          // NaN-propagating maximum
    }
  }
  return worst;
}
}  // namespace

extern "C" int spr_host_eig_top(const double *h_G, int32_t m, int32_t r, double *h_lam, double *h_V, void *fn_dsytrd,
                                void *fn_dsterf, void *fn_dormtr) {
  SPR_REQUIRE(h_G && h_lam && h_V && fn_dsytrd && fn_dsterf && fn_dormtr, SPR_E_INVALID, "spr_host_eig_top: NULL pointer");
  SPR_REQUIRE(m >= 2 && r >= 1 && r <= m && m <= 4096, SPR_E_INVALID, "spr_host_eig_top: bad shape m=%d r=%d", m, r);
  const size_t mm = (size_t)m * m, mr = (size_t)m * r;
  int lwork = m * 64;
  double *buf = static_cast<double *>(malloc(sizeof(double) * (mm + 5 * (size_t)m + (size_t)lwork + 2 * mr + 2 * (size_t)r * r + 5 * mr + 3 * (size_t)r)));
  SPR_REQUIRE(buf != nullptr, SPR_E_WORKSPACE, "spr_host_eig_top: out of host memory");
  double *A = buf, *d = A + mm, *e = d + m, *tau = e + m, *d2 = tau + m, *e2 = d2 + m, *work = e2 + m;
  double *Z = work + lwork, *Zc = Z + mr, *E = Zc + mr, *E2 = E + (size_t)r * r, *vwork = E2 + (size_t)r * r;
  memcpy(A, h_G, sizeof(double) * mm);                       // symmetric: row-major = column-major
  char L = 'L', N = 'N';
  int mi = m, ri = r, info = 0;
  reinterpret_cast<dsytrd_fn>(fn_dsytrd)(&L, &mi, A, &mi, d, e, tau, work, &lwork, &info);
  if (info == 0) {
    memcpy(d2, d, sizeof(double) * m);
    memcpy(e2, e, sizeof(double) * (m - 1));
    reinterpret_cast<dsterf_fn>(fn_dsterf)(&mi, d2, e2, &info);
  }
  if (info != 0) { free(buf); return 1; }
  for (int i = 0; i < m; ++i) h_lam[i] = d2[m - 1 - i];      // descending
  int rc = 0;
  tridiag_vectors_core(d, e, m, d2 + (m - r), r, Z, 4, vwork);   // column j <-> ascending eigenvalue m - r + j
  double defect = gram_defect(Z, m, r, E);
  if (!(defect <= 1e-8)) rc = 2;
  if (rc == 0) {
    // Z <- Z (I - E / 2), written column-major for dormtr
    for (int k = 0; k < m; ++k) {
      const double *zk = Z + (size_t)k * r;
      for (int j = 0; j < r; ++j) {
        double acc = zk[j];
        for (int i = 0; i < r; ++i) acc -= 0.5 * zk[i] * E[(size_t)i * r + j];
        Zc[(size_t)j * m + k] = acc;
      }
    }
    reinterpret_cast<dormtr_fn>(fn_dormtr)(&L, &L, &N, &mi, &ri, A, &mi, tau, Zc, &mi, work, &lwork, &info);
    if (info != 0) rc = 1;
  }
  if (rc == 0) {
    for (int k = 0; k < m; ++k)                               // row-major, columns in DESCENDING order of eigenvalue
      for (int j = 0; j < r; ++j) h_V[(size_t)k * r + j] = Zc[(size_t)(r - 1 - j) * m + k];
    defect = gram_defect(h_V, m, r, E2);
    if (!(defect <= 1e-12)) rc = 2;
  }
  free(buf);
  return rc;
}


// ---- the r leading RIGHT singular vectors and all singular values of a square matrix ------------------------------------------
// fit()'s conditioning refinement (openmeasure_amd/rom.py, _refine_spectrum) ends with the SVD of an m x m factor M of which only the
// singular values and the r retained right singular vectors are used; LAPACK's dgesdd computes all 2 m^2 vector entries (6.1 of the
// 7 host milliseconds of a refinement pass at m = 256).  Here: dgebrd (M^T = Q B P^T in LAPACK's column-major reading of the
// row-major M), dbdsdc for the singular values of the bidiagonal B alone, the r leading LEFT vectors of B by the batched inverse
// iteration above on B's Golub-Kahan form -- the 2m x 2m tridiagonal matrix with zero diagonal and off-diagonal d1, e1, d2, e2, ...,
// whose eigenvector for +sigma is (v1, u1, v2, u2, ...) / sqrt(2): it inherits the relative accuracy of the small singular values,
// which B^T B would lose -- and dormbr for Q times those r vectors: left vectors of M^T = right vectors of M.  Same checks as
// spr_host_eig_top (orthonormality 1e-8 before the symmetric correction, 1e-12 after); status 2 tells the caller to take dgesdd.
namespace {
typedef void (*dgebrd_fn)(int *, int *, double *, int *, double *, double *, double *, double *, double *, int *, int *);
typedef void (*dbdsdc_fn)(char *, char *, int *, double *, double *, double *, int *, double *, int *, double *, int *, double *,
                          int *, int *);
typedef void (*dormbr_fn)(char *, char *, char *, int *, int *, int *, double *, int *, double *, double *, int *, double *, int *,
                          int *);
}  // namespace

extern "C" int spr_host_svd_top(const double *h_M, int32_t m, int32_t r, double *h_S, double *h_V, void *fn_dgebrd,
                                void *fn_dbdsdc, void *fn_dormbr) {
  SPR_REQUIRE(h_M && h_S && h_V && fn_dgebrd && fn_dbdsdc && fn_dormbr, SPR_E_INVALID, "spr_host_svd_top: NULL pointer");
  SPR_REQUIRE(m >= 2 && r >= 1 && r <= m && m <= 4096, SPR_E_INVALID, "spr_host_svd_top: bad shape m=%d r=%d", m, r);
  const size_t mm = (size_t)m * m, mr = (size_t)m * r, m2 = 2 * (size_t)m;
  int lwork = m * 128;
  const size_t n_dbl = mm + 6 * (size_t)m + (size_t)lwork + 2 * m2 + m2 * r + 2 * mr + 2 * (size_t)r * r + (5 * m2 * r + 3 * (size_t)r) + 8;
  double *buf = static_cast<double *>(malloc(sizeof(double) * n_dbl + sizeof(int) * 8 * (size_t)m));
  SPR_REQUIRE(buf != nullptr, SPR_E_WORKSPACE, "spr_host_svd_top: out of host memory");
  double *A = buf, *d = A + mm, *e = d + m, *tauq = e + m, *taup = tauq + m, *sv = taup + m, *se = sv + m, *work = se + m;
  double *dT = work + lwork, *eT = dT + m2, *Z = eT + m2, *Ub = Z + m2 * r, *Uc = Ub + mr, *E = Uc + mr, *E2 = E + (size_t)r * r;
  double *vwork = E2 + (size_t)r * r;
  int *iwork = reinterpret_cast<int *>(vwork + (5 * m2 * r + 3 * (size_t)r) + 8);
  memcpy(A, h_M, sizeof(double) * mm);                       // row-major M = column-major M^T
  int mi = m, ri = r, info = 0, one = 1;
  reinterpret_cast<dgebrd_fn>(fn_dgebrd)(&mi, &mi, A, &mi, d, e, tauq, taup, work, &lwork, &info);
  int rc = 0;
  if (info == 0) {
    memcpy(sv, d, sizeof(double) * m);
    memcpy(se, e, sizeof(double) * (m - 1));
    char U = 'U', N = 'N';
    double dummy = 0.0;
    int idummy = 0;
    // singular values only: workspace 4 m doubles, 8 m integers
    reinterpret_cast<dbdsdc_fn>(fn_dbdsdc)(&U, &N, &mi, sv, se, &dummy, &one, &dummy, &one, &dummy, &idummy, work, iwork, &info);
  }
  if (info != 0) { free(buf); return 1; }
  for (int i = 0; i < m; ++i) h_S[i] = sv[i];                  // dbdsdc leaves them in decreasing order
  // Golub-Kahan form of the (upper) bidiagonal B: zero diagonal, off-diagonal d1, e1, d2, e2, ..., d_m
  for (size_t i = 0; i < m2; ++i) dT[i] = 0.0;
  for (int k = 0; k < m; ++k) {
    eT[2 * k] = d[k];
    if (k < m - 1) eT[2 * k + 1] = e[k];
  }
  tridiag_vectors_core(dT, eT, (int)m2, sv, r, Z, 4, vwork);   // column j <-> +sigma_j, j = 0 .. r-1 (the r largest)
  // left vectors of B: the entries 1, 3, 5, ... of the Golub-Kahan vector, normalised (row-major m x r)
  for (int j = 0; j < r; ++j) {
    double nn = 0.0;
    for (int k = 0; k < m; ++k) { const double v = Z[(size_t)(2 * k + 1) * r + j]; nn += v * v; }
    const double inv = nn > 0.0 ? 1.0 / sqrt(nn) : 0.0;
    for (int k = 0; k < m; ++k) Ub[(size_t)k * r + j] = Z[(size_t)(2 * k + 1) * r + j] * inv;
  }
  double defect = gram_defect(Ub, m, r, E);
  if (!(defect <= 1e-8)) rc = 2;
  if (rc == 0) {
    for (int k = 0; k < m; ++k) {                             // Ub (I - E / 2), column-major for dormbr
      const double *zk = Ub + (size_t)k * r;
      for (int j = 0; j < r; ++j) {
        double acc = zk[j];
        for (int i = 0; i < r; ++i) acc -= 0.5 * zk[i] * E[(size_t)i * r + j];
        Uc[(size_t)j * m + k] = acc;
      }
    }
    char Qc = 'Q', L = 'L', N = 'N';
    reinterpret_cast<dormbr_fn>(fn_dormbr)(&Qc, &L, &N, &mi, &ri, &mi, A, &mi, tauq, Uc, &mi, work, &lwork, &info);
    if (info != 0) rc = 1;
  }
  if (rc == 0) {
    for (int k = 0; k < m; ++k)
      for (int j = 0; j < r; ++j) h_V[(size_t)k * r + j] = Uc[(size_t)j * m + k];
    defect = gram_defect(h_V, m, r, E2);
    if (!(defect <= 1e-12)) rc = 2;
  }
  free(buf);
  return rc;
}
