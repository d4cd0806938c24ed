// K10 + K11: x = X_scl * (Ur a) + X_cnt  -- one streaming pass over the basis shard.
//
// HBM-bound (2 flops per 8 bytes).  A row of Ur (r doubles) is read by LPR consecutive
// lanes as 16-byte pieces (LPR = smallest power of two >= r/2), so a wave instruction
// covers 64/LPR whole rows of contiguous memory; each lane keeps UNR row groups in flight.
// The dot product is closed with a butterfly over the LPR lanes and the un-scaling
// (sparse_sensing.py:235) is applied before the store.  Workgroups are dealt to feature
// segments (common.hpp) so the per-feature scale is a workgroup constant.
//
// Two forms.  The MFMA form (default) treats the pass as the skinny product A . Ur^T with up to 16 coefficient
// vectors at once: 64-row panels of Ur are staged raw in LDS by the rowtile.hpp loader (double-buffered, the
// loads of the panel after next issued behind the stores), wave w multiplies the vectors (MFMA A operand, in
// registers for the whole kernel) with its 16-row block (B operand, strided ds_read_b64): the result tile has
// the vector index down the rows and the panel row across the 16 lanes of a group, so each store instruction
// writes 16 consecutive field values per vector and no cross-lane reduction is needed at all.  r/4 MFMAs per
// 16 rows are far below the HBM time of the panel.  The VALU form (dot product closed with a DPP butterfly,
// 4 vectors per pass) is kept for cross-checks (SPR_RECONSTRUCT_VALU=1 at build time selects it).
#include <stdlib.h>

#include <type_traits>

#include "rowtile.hpp"

#ifndef SPR_RECONSTRUCT_VALU
#define SPR_RECONSTRUCT_VALU 0
#endif

namespace {

constexpr int RM_THREADS = 256;
constexpr int RM_PB = 16;   // coefficient vectors per pass (one MFMA tile of rows)

template <int MTR, int VEC, typename TU>
__global__ __launch_bounds__(RM_THREADS) void reconstruct_mfma_kernel(
    const TU *__restrict__ Ur, int r, int64_t ldu, SegPlan plan, const double *__restrict__ rowmean,
    const double *__restrict__ scale, const double *__restrict__ rowscale, const double *__restrict__ A, int64_t lda,
    int np0, int npb, double *__restrict__ out, int64_t ldo, int accumulate) {
  constexpr int NW = RM_THREADS / 64, R = 64;
  constexpr int MPAD = 16 * MTR, MP = MPAD + 2, KSTEPS = MPAD / 4;
  using RT = RowTile<MTR, R, MP, NW, 16, TU>;
  __shared__ double smem[2 * R * MP];
  double *const lds0 = smem, *const lds1 = smem + R * MP;
  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const double sc = scale[f];

  double vfrag[KSTEPS];          // MFMA A operand: A[i = lane & 15][k = lane >> 4] = vector np0+i, entry 4 ks + k
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int k = 4 * ks + (lane >> 4), j = lane & 15;
    vfrag[ks] = (j < npb && k < r) ? A[(int64_t)(np0 + j) * lda + k] : 0.0;
  }
  RT tile;
  const int64_t npanels = (hi - lo + R - 1) / R;
  int64_t c = wl;
  if (c >= npanels) return;                       // workgroup-uniform
  tile.template load<VEC>(Ur, ldu, r, lo + c * R, hi, wave, lane);
  tile.raw_store(lds0, r, lo + c * R, hi, wave, lane);
  int64_t cn = c + wpf;
  int64_t nrow0 = (cn < npanels) ? lo + cn * R : hi;
  tile.template load<VEC>(Ur, ldu, r, nrow0, hi, wave, lane);
  int buf = 0;
  const int ufrag = (lane & 15) * MP + (lane >> 4);   // B[k = lane >> 4][j = lane & 15] = panel[16 w + j][k0 + k]
  while (c < npanels) {
    const double *cur = buf ? lds1 : lds0;
    double *nxt = buf ? lds0 : lds1;
    const int64_t c2 = cn + wpf;
    const int64_t n2row0 = (c2 < npanels) ? lo + c2 * R : hi;
    __syncthreads();
    const int64_t row = lo + c * R + wave * 16 + (lane & 15);   // the panel row this lane's results belong to
    const int64_t rc = row < hi ? row : hi - 1;
    const double mu = rowmean[rc];                               // requested before the MFMAs
    const double rs = rowscale ? rowscale[rc] : sc;              // sampled rows carry their own scale
    const double *p = cur + wave * 16 * MP + ufrag;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vfrag[ks], p[4 * ks], acc, 0, 0, 0);
      if (ks == 0) {
#pragma unroll
        for (int it = 0; it < RT::IT; ++it) {
          tile.raw_store_pass(it, nxt, r, nrow0, hi, wave, lane);
          tile.template load_pass<VEC>(it, Ur, ldu, r, n2row0, hi, wave, lane);
        }
      }
    }
    // D[i = (lane >> 4) + 4 q][j = lane & 15] = a_{np0+i} . u_row
    const double d[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pv = 4 * q + (lane >> 4);
      if (pv < npb && row < hi) {
        // accumulate: a further column group of a basis wider than one launch handles -- the centre is already in
        double *o = out + (int64_t)(np0 + pv) * ldo + row;
        *o = rs * d[q] + (accumulate ? *o : mu);
      }
    }
    buf ^= 1;
    c = cn;
    cn = c2;
    nrow0 = n2row0;
  }
}

template <int MTR, typename TU>
int launch_mfma(const TU *Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0, int64_t n_points,
                int32_t n_features, const double *rowmean, const double *scale, const double *rowscale,
                const double *A, int64_t lda, int32_t n_p, double *out, int64_t ldo, int accumulate, hipStream_t st) {
  const int cus = spr_cached_cus();
  SegPlan plan;
  plan.row0 = row0; plan.n_rows = n_rows; plan.n_points = n_points; plan.n_features = n_features;
  // LDS: 2 x 64 x (16 MTR + 2) doubles per workgroup -> 6 / 4 / 3 / 2 / 1 / 1 workgroups per CU
  constexpr int PER_CU = MTR <= 1 ? 6 : MTR == 2 ? 4 : MTR == 3 ? 3 : MTR == 4 ? 2 : 1;
  plan.total_wg = PER_CU * (cus > 0 ? cus : 256);
  plan.chunk_rows = 64;
  const int grid = seg_total_wgs(plan);
  const int vec_ok = (r % 2 == 0) && (ldu % 2 == 0) && ((reinterpret_cast<uintptr_t>(Ur) & (2 * sizeof(TU) - 1)) == 0);
  const int lm = vec_ok ? ((r == 16 * MTR) ? 2 : 1) : 0;
  for (int p0 = 0; p0 < n_p; p0 += RM_PB) {
    const int npb = (n_p - p0 < RM_PB) ? n_p - p0 : RM_PB;
#define RM(LM) hipLaunchKernelGGL((reconstruct_mfma_kernel<MTR, LM, TU>), dim3(grid), dim3(RM_THREADS), 0, st, Ur, (int)r, ldu, plan, rowmean, scale, rowscale, A, lda, p0, npb, out, ldo, accumulate)
    if (lm == 2) RM(2);
    else if (lm == 1) RM(1);
    else RM(0);
#undef RM
    SPR_LAUNCH_CHECK();
  }
  return SPR_OK;
}

// Register-direct form for an f32-stored basis (BASELINE config 5): its rows carry half the bytes for the same MFMAs and the
// same fixed cost per LDS panel, which held the panel form at 3.1 TB/s (133 KB of LDS per workgroup at r = 128: one
// workgroup, four waves, per CU).  Here -- as in qr_refresh_direct_kernel -- lane (j = l & 15, kk = l >> 4) of a wave owns row
// j of the wave's 16-row block and loads the 16-byte pieces [16 g + 4 kk, +4) of it straight into the MFMA B operand
// (contraction index permuted accordingly, the coefficient fragments are permuted the same way once per kernel): no LDS,
// no barrier, every wave independent, the next block requested before this one is multiplied.
template <int NG, typename TU>
__global__ __launch_bounds__(RM_THREADS) void reconstruct_direct_kernel(
    const TU *__restrict__ Ur, int r, int64_t ldu, SegPlan plan, const double *__restrict__ rowmean,
    const double *__restrict__ scale, const double *__restrict__ rowscale, const double *__restrict__ A, int64_t lda,
    int np0, int npb, double *__restrict__ out, int64_t ldo, int accumulate) {
  constexpr int R = 64;
  using P4 = typename std::conditional<std::is_same<TU, float>::value, float4, double4>::type;
  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kk = lane >> 4;
  const double sc = scale[f];
  double vfrag[4 * NG];          // A[i = li][k]: vector np0 + li, entry 16 g + 4 kk + t at MFMA step 4 g + t
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int t = 0; t < 4; ++t) vfrag[4 * g + t] = (li < npb) ? A[(int64_t)(np0 + li) * lda + 16 * g + 4 * kk + t] : 0.0;
  const int64_t npanels = (hi - lo + R - 1) / R;
  auto load_block = [&](int64_t c, P4 (&dst)[NG]) {
    int64_t row = lo + c * R + wave * 16 + li;
    row = row < hi ? row : hi - 1;                             // past the segment: a harmless re-read, never stored
    const TU *rp = Ur + row * ldu + 4 * kk;
#pragma unroll
    for (int g = 0; g < NG; ++g) dst[g] = *reinterpret_cast<const P4 *>(rp + 16 * g);
  };
  int64_t c = wl;
  if (c >= npanels) return;
  P4 cur[NG], nxt[NG];
  load_block(c, cur);
  while (c < npanels) {
    const int64_t cn = c + wpf;
    load_block(cn < npanels ? cn : c, nxt);
    const int64_t row = lo + c * R + wave * 16 + li;           // the row this lane's results belong to
    const int64_t rc = row < hi ? row : hi - 1;
    const double mu = rowmean[rc];
    const double rs = rowscale ? rowscale[rc] : sc;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const double b4[4] = {(double)cur[g].x, (double)cur[g].y, (double)cur[g].z, (double)cur[g].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vfrag[4 * g + t], b4[t], acc, 0, 0, 0);
    }
    const double d[4] = {acc.x, acc.y, acc.z, acc.w};          // D[i = kk + 4 q][j = li] = a_{np0+i} . u_row
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pv = 4 * q + kk;
      if (pv < npb && row < hi) {
        double *o = out + (int64_t)(np0 + pv) * ldo + row;
        *o = rs * d[q] + (accumulate ? *o : mu);
      }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) cur[g] = nxt[g];
    c = cn;
  }
}

template <int NG, typename TU>
int launch_direct(const TU *Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0, int64_t n_points,
                  int32_t n_features, const double *rowmean, const double *scale, const double *rowscale,
                  const double *A, int64_t lda, int32_t n_p, double *out, int64_t ldo, int accumulate, hipStream_t st) {
  const int cus = spr_cached_cus();
  SegPlan plan;
  plan.row0 = row0; plan.n_rows = n_rows; plan.n_points = n_points; plan.n_features = n_features;
  plan.total_wg = 4 * (cus > 0 ? cus : 256);                   // no LDS: registers decide (16 waves per CU)
  plan.chunk_rows = 64;
  const int grid = seg_total_wgs(plan);
  for (int p0 = 0; p0 < n_p; p0 += RM_PB) {
    const int npb = (n_p - p0 < RM_PB) ? n_p - p0 : RM_PB;
    hipLaunchKernelGGL((reconstruct_direct_kernel<NG, TU>), dim3(grid), dim3(RM_THREADS), 0, st, Ur, (int)r, ldu, plan,
                       rowmean, scale, rowscale, A, lda, p0, npb, out, ldo, accumulate);
    SPR_LAUNCH_CHECK();
  }
  return SPR_OK;
}

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

// r <= 128 per launch; a wider basis (r <= m in the reference, :336) goes in column groups of 128 that accumulate
template <typename TU>
int reconstruct_groups(const TU *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0, int64_t n_points,
                       int32_t n_features, const double *d_rowmean, const double *d_scale, const double *d_rowscale,
                       const double *d_A, int32_t n_p, double *d_Xrec, int64_t ldo, hipStream_t st) {
  for (int g0 = 0; g0 < r; g0 += SPR_MAX_R) {
    const int rg = (r - g0 < SPR_MAX_R) ? r - g0 : SPR_MAX_R;
    int rc = SPR_OK;
    {
      // whole 16-column groups, 16-byte aligned rows: the register-direct form for an f32 basis and for f64 bases wider than
      // 96 columns (whose LDS panels leave room for one workgroup per CU only: 3.5 -> 2.8 ms at the c5s shape).
      // SPR_RECONSTRUCT_DIRECT=0: panels everywhere; =3: direct everywhere (A/B switches)
      static const int direct_mode = [] { const char *e = getenv("SPR_RECONSTRUCT_DIRECT"); return e ? atoi(e) : 1; }();
      const bool direct_on = direct_mode == 3 || (direct_mode != 0 && (std::is_same<TU, float>::value || rg > 96));
      if (direct_on && rg % 16 == 0 && (ldu * sizeof(TU)) % 16 == 0 && (reinterpret_cast<uintptr_t>(d_Ur + g0) & 15) == 0) {
#define RDF(NGV) rc = launch_direct<NGV, TU>(d_Ur + g0, n_rows, rg, ldu, row0, n_points, n_features, d_rowmean, d_scale, d_rowscale, d_A + g0, r, n_p, d_Xrec, ldo, g0 > 0, st); break
        switch (rg / 16) {
          case 1: RDF(1);
          case 2: RDF(2);
          case 3: RDF(3);
          case 4: RDF(4);
          case 5: RDF(5);
          case 6: RDF(6);
          case 7: RDF(7);
          default: RDF(8);
        }
#undef RDF
        if (rc != SPR_OK) return rc;
        continue;
      }
    }
#define RMF(MTV) rc = launch_mfma<MTV, TU>(d_Ur + g0, n_rows, rg, ldu, row0, n_points, n_features, d_rowmean, d_scale, d_rowscale, d_A + g0, r, n_p, d_Xrec, ldo, g0 > 0, st); break
    switch (spr_round_mt(rg)) {      // padded width of the group in 16-column tiles
      case 1: RMF(1);
      case 2: RMF(2);
      case 3: RMF(3);
      case 4: RMF(4);
      case 6: RMF(6);
      default: RMF(8);
    }
#undef RMF
    if (rc != SPR_OK) return rc;
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
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!SPR_RECONSTRUCT_VALU || r > SPR_MAX_R)
    return reconstruct_groups<double>(d_Ur, n_rows, r, ldu, row0, n_points, n_features, d_rowmean, d_scale, d_rowscale,
                                      d_A, n_p, d_Xrec, ldo, st);
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

// Sharded reconstruct() with several coefficient vectors: the ONE all-gather of the ranks' (n_p, n_loc) blocks leaves
// stage[q][v][i]; the reference's result has the vectors as columns of the whole field (:371-375), i.e. out[v][q n_loc + i].
namespace {
__global__ __launch_bounds__(256) void field_unstage_kernel(const double *__restrict__ stage, int world, int n_p,
                                                            int64_t n_loc, double *__restrict__ out, int64_t ldo) {
  // one (q, v) block per blockIdx.y; 16-byte pieces where both sides are aligned, scalar tail otherwise
  const int q = blockIdx.y / n_p, v = blockIdx.y - q * n_p;
  const double *src = stage + ((int64_t)q * n_p + v) * n_loc;
  double *dst = out + (int64_t)v * ldo + (int64_t)q * n_loc;
  const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (vec) {
    const int64_t n2 = n_loc / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride)
      reinterpret_cast<f64x2 *>(dst)[i] = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(src) + i);
    if ((n_loc & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n_loc - 1] = src[n_loc - 1];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_loc; i += stride) dst[i] = src[i];
  }
}
}  // namespace

extern "C" int spr_field_unstage_f64(const double *d_stage, int32_t world, int32_t n_p, int64_t n_loc, double *d_out,
                                     int64_t ldo, void *stream) {
  SPR_REQUIRE(d_stage && d_out, SPR_E_INVALID, "spr_field_unstage_f64: NULL pointer");
  SPR_REQUIRE(world > 0 && n_p > 0 && n_loc > 0 && ldo >= (int64_t)world * n_loc && (int64_t)world * n_p <= 65535,
              SPR_E_INVALID, "spr_field_unstage_f64: bad shape world=%d n_p=%d n_loc=%lld ldo=%lld", world, n_p,
              (long long)n_loc, (long long)ldo);
  int64_t bx = (n_loc / 2 + 255) / 256;
  if (bx > 1024) bx = 1024;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(field_unstage_kernel, dim3((unsigned)bx, (unsigned)(world * n_p)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d_stage, (int)world, (int)n_p, n_loc, d_out, ldo);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

namespace {
__global__ __launch_bounds__(256) void field_unstage_blocks_kernel(const double *__restrict__ stage, int n_p, int64_t n_max,
                                                                   const int64_t *__restrict__ layout,
                                                                   double *__restrict__ out, int64_t ldo) {
  // one (q, v) block per blockIdx.y, rows_q of its n_max padded entries; same copy loop as field_unstage_kernel
  const int q = blockIdx.y / n_p, v = blockIdx.y - q * n_p;
  const int64_t off = layout[2 * q], rows = layout[2 * q + 1];
  const double *src = stage + ((int64_t)q * n_p + v) * n_max;
  double *dst = out + (int64_t)v * ldo + off;
  const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (vec) {
    const int64_t n2 = rows / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride)
      reinterpret_cast<f64x2 *>(dst)[i] = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(src) + i);
    if ((rows & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[rows - 1] = src[rows - 1];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += stride) dst[i] = src[i];
  }
}
}  // namespace

extern "C" int spr_field_unstage_blocks_f64(const double *d_stage, int32_t world, int32_t n_p, int64_t n_max,
                                            const int64_t *d_layout, double *d_out, int64_t ldo, void *stream) {
  SPR_REQUIRE(d_stage && d_out && d_layout, SPR_E_INVALID, "spr_field_unstage_blocks_f64: NULL pointer");
  SPR_REQUIRE(world > 0 && n_p > 0 && n_max > 0 && ldo > 0 && (int64_t)world * n_p <= 65535, SPR_E_INVALID,
              "spr_field_unstage_blocks_f64: bad shape world=%d n_p=%d n_max=%lld ldo=%lld", world, n_p, (long long)n_max,
              (long long)ldo);
  int64_t bx = (n_max / 2 + 255) / 256;
  if (bx > 1024) bx = 1024;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(field_unstage_blocks_kernel, dim3((unsigned)bx, (unsigned)(world * n_p)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), d_stage, (int)n_p, n_max, d_layout, d_out, ldo);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

// basis stored as f32 (the dtype of an f32 snapshot shard's U), arithmetic and output f64
extern "C" int spr_reconstruct_u32(const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                                   int64_t n_points, int32_t n_features, const double *d_rowmean,
                                   const double *d_scale, const double *d_rowscale, const double *d_A, int32_t n_p,
                                   double *d_Xrec, int64_t ldo, void *stream) {
  SPR_REQUIRE(d_Ur && d_rowmean && d_scale && d_A && d_Xrec, SPR_E_INVALID, "spr_reconstruct_u32: NULL pointer");
  SPR_REQUIRE(n_rows > 0 && r > 0 && ldu >= r && n_p > 0 && ldo >= n_rows, SPR_E_INVALID,
              "spr_reconstruct_u32: bad shape n_rows=%lld r=%d ldu=%lld n_p=%d ldo=%lld", (long long)n_rows, r,
              (long long)ldu, n_p, (long long)ldo);
  SPR_REQUIRE(n_points > 0 && n_features > 0 && row0 >= 0 &&
                  row0 + n_rows <= n_points * (int64_t)n_features,
              SPR_E_INVALID, "spr_reconstruct_u32: bad feature layout");
  return reconstruct_groups<float>(d_Ur, n_rows, r, ldu, row0, n_points, n_features, d_rowmean, d_scale, d_rowscale, d_A,
                                   n_p, d_Xrec, ldo, static_cast<hipStream_t>(stream));
}
