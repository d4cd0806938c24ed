// Gram matrix for 256 < m <= 512 snapshots (BASELINE config 5 has m = 512).
//
// One workgroup cannot hold the 528 upper tiles of a 512 x 512 Gram matrix in registers, so
// the columns are split into A = [0, 256) and B = [256, m):
//     G = [ A^T A   A^T B ]      A^T A, B^T B : the symmetric kernel of stats_gram.hip on column
//         [ B^T A   B^T B ]                     slices (ldx = full row stride, centre mode 2),
//                                A^T B        : gram_cross_kernel below.
// The row means must be those of the WHOLE row, so they are produced first by a light streaming
// pass (rowstats_kernel: row means + per-feature Welford partials) and handed to the three Gram
// launches as external means.  MFMA work is exactly the 528 tiles (136 + 136 + 256); X is read
// 1 + 1 + 2 times instead of once, which at m > 256 is still far below the MFMA time.
#include "rowtile.hpp"

namespace {

__device__ inline void chan_merge_w(double &n, double &mu, double &m2, double nb, double mb, double sb) {
  if (nb > 0.0) {
    const double tot = n + nb, d = mb - mu;
    mu += d * nb / tot;
    m2 += sb + d * d * n * nb / tot;
    n = tot;
  }
}

// ---- row means + per-feature statistics of full rows (any m) --------------------------------
template <typename TX>
__global__ __launch_bounds__(256) void rowstats_kernel(const TX *__restrict__ X, int64_t ldx, int m, int vec_ok,
                                                       SegPlan plan, double *__restrict__ rowmean,
                                                       double *__restrict__ stat_part) {
  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double inv_m = 1.0 / (double)m;
  RowStats st;
  st.init();
  for (int64_t row = lo + (int64_t)wl * 4 + wave; row < hi; row += (int64_t)wpf * 4) {
    const TX *rp = X + row * ldx;
    double s = 0.0;
    if (vec_ok) {
      for (int c = 2 * lane; c < m; c += 128) {
        const f64x2 t = widen(*reinterpret_cast<const typename PieceOf<TX>::type *>(rp + c));
        s += t.x + t.y;
      }
    } else {
      for (int c = lane; c < m; c += 64) s += (double)rp[c];
    }
    s = group_sum_t<64>(s);
    const double mean = s * inv_m;
    if (lane == 0) rowmean[row] = mean;
    st.push(mean, true);
  }
  if (lane == 0) {
    double *q = stat_part + ((int64_t)blockIdx.x * 4 + wave) * 3;
    q[0] = st.cnt; q[1] = st.mean(); q[2] = st.m2();
  }
}

__global__ void rowstats_finalize_kernel(const double *__restrict__ stat_part, SegPlan plan, double *__restrict__ fstats) {
  const int f = blockIdx.x;
  int base = 0, wpf = 0, acc = 0;
  for (int ff = seg_first_feature(plan); ff <= seg_last_feature(plan); ++ff) {
    int64_t lo, hi;
    seg_range(plan, ff, lo, hi);
    const int w = seg_wgs(plan, hi - lo);
    if (ff == f) { base = acc; wpf = w; }
    acc += w;
  }
  if (threadIdx.x == 0) {   // fixed order: reproducible
    double n = 0.0, mu = 0.0, m2 = 0.0;
    const double *q = stat_part + (int64_t)base * 4 * 3;
    for (int p = 0; p < wpf * 4; ++p) chan_merge_w(n, mu, m2, q[3 * p], q[3 * p + 1], q[3 * p + 2]);
    fstats[3 * f] = n; fstats[3 * f + 1] = mu; fstats[3 * f + 2] = m2;
  }
}

// ---- per-feature statistics (count, mean, M2) of row means that already exist -----------------
// Same partial layout and fixed-order finalize as rowstats_kernel, but reads the n row means instead of X.
__global__ __launch_bounds__(256) void rowmean_stats_kernel(const double *__restrict__ rowmean, SegPlan plan,
                                                            double *__restrict__ stat_part) {
  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  RowStats st;
  st.init();
  // wave w of workgroup wl takes the rows of its 4-row groups; lane 0..63 strided inside 256-row spans
  for (int64_t row = lo + ((int64_t)wl * 4 + wave) * 64 + lane; row < hi; row += (int64_t)wpf * 256)
    st.push(rowmean[row], true);
  // lane partials -> one (count, mean, M2) per wave, fixed lane order
  double n = st.cnt, mu = st.mean(), m2 = st.m2();
  for (int o = 32; o > 0; o >>= 1) {
    const double on = __shfl_down(n, o, 64), om = __shfl_down(mu, o, 64), os = __shfl_down(m2, o, 64);
    chan_merge_w(n, mu, m2, on, om, os);
  }
  if (lane == 0) {
    double *q = stat_part + ((int64_t)blockIdx.x * 4 + wave) * 3;
    q[0] = n; q[1] = mu; q[2] = m2;
  }
}

// ---- cross block A^T B ------------------------------------------------------------------------
// Panels of 16 full rows (both column halves) in LDS, centred with the external row means.  Two
// workgroup flavours (og = 0, 1) share the 16 tile rows of A: wave w of flavour og owns tile row
// ti = 8 og + w and all NTJ tile columns of B -- one A fragment and NTJ B fragments per k step for
// NTJ MFMAs, everything at compile-time offsets except the wave's own A column block.
constexpr int CW = 8;        // waves
constexpr int CR = 16;       // panel rows
constexpr int CMA = 256;     // width of A

// OWN = true (centre mode 1): the row means are formed in this pass from the full rows the panel holds anyway (the
// staging computes the row sums in every mode) and written to rowmean[] -- both flavours write the same bits --, so
// the separate row-statistics pass over X is not needed; OWN = false: external means (mode 2) or none (mode 0).
template <int NTJ, int VEC, typename TX, bool OWN>
__global__ __launch_bounds__(CW * 64) void gram_cross_kernel(const TX *__restrict__ X, int64_t ldx, int m,
                                                            int center, SegPlan plan,
                                                            double *__restrict__ rowmean,
                                                            double *__restrict__ slab, int64_t gap) {
  constexpr int MTF = 16 + NTJ;                 // padded full width in tiles
  constexpr int MP = 16 * MTF + ((MTF % 2 == 0) ? 16 : 0);
  constexpr int KSTEPS = CR / 4;
  using RT = RowTile<MTF, CR, MP, CW, 32, TX>;
  __shared__ double lds[2][CR * MP];

  // The two flavours of a pair read the same rows: give them block indices that differ by 8, i.e. (workgroups being
  // dealt round-robin over the 8 XCDs) the same XCD and L2, so the second read of a panel is an L2 hit.  Placement is
  // a speed matter only; the last npairs % 8 pairs keep neighbouring block indices.
  const int npairs_full = (int)((gridDim.x >> 1) & ~7u);
  int og, pair;
  if ((int)blockIdx.x < 2 * npairs_full) { og = (blockIdx.x >> 3) & 1; pair = (int)((blockIdx.x >> 4) << 3) | (int)(blockIdx.x & 7); }
  else { const int t = (int)blockIdx.x - 2 * npairs_full; og = t & 1; pair = npairs_full + (t >> 1); }
  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, pair, f, wl, wpf, base, lo, hi)) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ti = og * 8 + wave;

  f64x4 acc[NTJ];
#pragma unroll
  for (int j = 0; j < NTJ; ++j) acc[j] = (f64x4){0.0, 0.0, 0.0, 0.0};

  RT tile;
  RowStats st;   // required by the staging interface; the feature statistics come from spr_rowmean_stats_f64
  st.init();
  const double *mean_in = OWN ? nullptr : rowmean;
  const int64_t nchunks = (hi - lo + CR - 1) / CR;
  int64_t c = wl;
  // m here is the PANEL width 256 + wB; panel columns >= 256 sit `gap` elements further along the row (0 when the two
  // slices are adjacent, as for m <= 512)
  tile.template load<VEC>(X, ldx, m, lo + c * CR, hi, wave, lane, mean_in, CMA, gap);
  tile.template center_store<OWN>(lds[0], m, center, lo + c * CR, hi, wave, lane, OWN ? rowmean : nullptr, &st);
  int64_t cn = c + wpf;
  int64_t nrow0 = (cn < nchunks) ? lo + cn * CR : hi;
  tile.template load<VEC>(X, ldx, m, nrow0, hi, wave, lane, mean_in, CMA, gap);
  int buf = 0;
  const int frag = (lane >> 4) * MP + (lane & 15);
  while (c < nchunks) {
    double *cur = lds[buf];
    double *nxt = lds[buf ^ 1];
    const int64_t c2 = cn + wpf;
    const int64_t n2row0 = (c2 < nchunks) ? lo + c2 * CR : hi;
    __syncthreads();
    const double *pa = cur + frag + 16 * ti;
    const double *pb = cur + frag + CMA;
#pragma unroll
    for (int k = 0; k < KSTEPS; ++k) {
      const double a = pa[k * 4 * MP];
      double b[NTJ];
#pragma unroll
      for (int j = 0; j < NTJ; ++j) b[j] = pb[k * 4 * MP + 16 * j];
#pragma unroll
      for (int j = 0; j < NTJ; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[j], acc[j], 0, 0, 0);
      if (k == 0) {
        tile.template center_store_pass<OWN>(0, nxt, m, center, nrow0, hi, wave, lane, OWN ? rowmean : nullptr, &st);
        tile.template load_pass<VEC>(0, X, ldx, m, n2row0, hi, wave, lane, mean_in, CMA, gap);
      }
    }
    buf ^= 1;
    c = cn;
    cn = c2;
    nrow0 = n2row0;
  }
  // slab[(block >> 1)][ti][tj][reg][lane]   (the two flavours of a block pair write disjoint tile rows)
  double *sp = slab + (((int64_t)pair * 16 + ti) * NTJ) * 256 + lane;
#pragma unroll
  for (int j = 0; j < NTJ; ++j) {
    double *tp = sp + (int64_t)j * 256;
    tp[0] = acc[j].x; tp[64] = acc[j].y; tp[128] = acc[j].z; tp[192] = acc[j].w;
  }
}

// grid (16 * NTJ tiles, n_features), 256 threads: fixed-order sum over the feature's block pairs
template <int NTJ>
__global__ __launch_bounds__(256) void gram_cross_finalize_kernel(const double *__restrict__ slab, int wB, SegPlan plan,
                                                                  double *__restrict__ gram, int ldg, int oa, int ob) {
  const int f = blockIdx.y;
  const int ti = blockIdx.x / NTJ, tj = blockIdx.x % NTJ;
  int base = 0, wpf = 0, acc = 0;
  for (int ff = seg_first_feature(plan); ff <= seg_last_feature(plan); ++ff) {
    int64_t lo, hi;
    seg_range(plan, ff, lo, hi);
    const int w = seg_wgs(plan, hi - lo);
    if (ff == f) { base = acc; wpf = w; }
    acc += w;
  }
  const int e = threadIdx.x;
  double s = 0.0;
  for (int p = 0; p < wpf; ++p) s += slab[((((int64_t)(base + p) * 16 + ti) * NTJ) + tj) * 256 + e];
  const int l = e & 63, reg = e >> 6;
  const int gi = oa + ti * 16 + (l >> 4) + 4 * reg, bj = tj * 16 + (l & 15), gj = ob + bj;
  if (bj < wB) {                                  // block (oa.., ob..) of the per-feature ldg x ldg matrix, and its mirror
    double *G = gram + (int64_t)f * ldg * ldg;
    G[(int64_t)gi * ldg + gj] = s;
    G[(int64_t)gj * ldg + gi] = s;
  }
}

int plan_wgs(SegPlan &plan, int64_t n_rows, int64_t row0, int64_t n_points, int32_t n_features, int chunk_rows,
             int per_cu) {
  const int cus = spr_cached_cus();
  plan.row0 = row0; plan.n_rows = n_rows; plan.n_points = n_points; plan.n_features = n_features;
  plan.total_wg = per_cu * (cus > 0 ? cus : 256);
  plan.chunk_rows = chunk_rows;
  return seg_total_wgs(plan);
}

// X points at column colA of the rows; the panel is [A = 256 columns at colA | B = wB columns at colB]
template <int NTJ, typename TX>
int launch_cross(const TX *X, int64_t n_rows, int wB, int64_t ldx, int64_t row0, int64_t n_points,
                 int32_t n_features, int center, double *rowmean, double *gram, void *ws, size_t ws_bytes,
                 hipStream_t st, int64_t gap, int ldg, int oa, int ob) {
  const int m = CMA + wB;
  SegPlan plan;
  const int pairs = plan_wgs(plan, n_rows, row0, n_points, n_features, CR, 1);   // one workgroup per CU: 2 flavours share them
  // halve the pair count so that pairs * 2 workgroups still fit one per CU
  plan.total_wg = plan.total_wg / 2 > 0 ? plan.total_wg / 2 : 1;
  const int npairs = seg_total_wgs(plan);
  (void)pairs;
  const size_t need = (size_t)npairs * 16 * NTJ * 256 * sizeof(double);
  SPR_REQUIRE(ws_bytes >= need, SPR_E_WORKSPACE, "spr_gram_cross_f64: workspace %zu < %zu", ws_bytes, need);
  const int vec_ok = (m % 2 == 0) && (ldx % 2 == 0) && (gap % 2 == 0) && ((reinterpret_cast<uintptr_t>(X) & (2 * sizeof(TX) - 1)) == 0);
  double *slab = static_cast<double *>(ws);
#define GXK(V, O) hipLaunchKernelGGL((gram_cross_kernel<NTJ, V, TX, O>), dim3(2 * npairs), dim3(CW * 64), 0, st, X, ldx, m, center, plan, rowmean, slab, gap)
  if (center == 1) { if (vec_ok) GXK(1, true); else GXK(0, true); }
  else { if (vec_ok) GXK(1, false); else GXK(0, false); }
#undef GXK
  SPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(gram_cross_finalize_kernel<NTJ>, dim3(16 * NTJ, n_features), dim3(256), 0, st, slab, wB, plan, gram, ldg, oa, ob);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

}  // namespace

extern "C" size_t spr_rowstats_workspace(int32_t n_features) {
  const int cus = spr_cached_cus();
  return sizeof(double) * 3 * 4 * ((size_t)8 * (cus > 0 ? cus : 256) + (size_t)n_features);
}

template <typename TX>
static int rowstats_entry(const char *who, const TX *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                          int64_t n_points, int32_t n_features, double *d_rowmean, double *d_fstats,
                          void *d_workspace, size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(d_X && d_rowmean && d_fstats && d_workspace, SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(n_rows > 0 && m > 0 && ldx >= m && row0 >= 0 && n_points > 0 && n_features > 0 &&
                  row0 + n_rows <= n_points * (int64_t)n_features,
              SPR_E_INVALID, "%s: bad shape", who);
  SPR_REQUIRE(workspace_bytes >= spr_rowstats_workspace(n_features), SPR_E_WORKSPACE, "%s: workspace too small", who);
  SegPlan plan;
  const int grid = plan_wgs(plan, n_rows, row0, n_points, n_features, 4, 8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int vec_ok = (m % 2 == 0) && (ldx % 2 == 0) && ((reinterpret_cast<uintptr_t>(d_X) & (2 * sizeof(TX) - 1)) == 0);
  hipLaunchKernelGGL(rowstats_kernel<TX>, dim3(grid), dim3(256), 0, st, d_X, ldx, (int)m, vec_ok, plan, d_rowmean,
                     static_cast<double *>(d_workspace));
  SPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(rowstats_finalize_kernel, dim3(n_features), dim3(64), 0, st,
                     static_cast<const double *>(d_workspace), plan, d_fstats);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

extern "C" int spr_rowstats_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                int64_t n_points, int32_t n_features, double *d_rowmean, double *d_fstats,
                                void *d_workspace, size_t workspace_bytes, void *stream) {
  return rowstats_entry("spr_rowstats_f64", d_X, n_rows, m, ldx, row0, n_points, n_features, d_rowmean, d_fstats,
                        d_workspace, workspace_bytes, stream);
}

extern "C" int spr_rowstats_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                int64_t n_points, int32_t n_features, double *d_rowmean, double *d_fstats,
                                void *d_workspace, size_t workspace_bytes, void *stream) {
  return rowstats_entry("spr_rowstats_x32", d_X, n_rows, m, ldx, row0, n_points, n_features, d_rowmean, d_fstats,
                        d_workspace, workspace_bytes, stream);
}

extern "C" int spr_rowmean_stats_f64(const double *d_rowmean, int64_t n_rows, int64_t row0, int64_t n_points,
                                     int32_t n_features, double *d_fstats, void *d_workspace, size_t workspace_bytes,
                                     void *stream) {
  SPR_REQUIRE(d_rowmean && d_fstats && d_workspace, SPR_E_INVALID, "spr_rowmean_stats_f64: NULL pointer");
  SPR_REQUIRE(n_rows > 0 && row0 >= 0 && n_points > 0 && n_features > 0 && row0 + n_rows <= n_points * (int64_t)n_features,
              SPR_E_INVALID, "spr_rowmean_stats_f64: bad shape");
  SPR_REQUIRE(workspace_bytes >= spr_rowstats_workspace(n_features), SPR_E_WORKSPACE,
              "spr_rowmean_stats_f64: workspace too small");
  SegPlan plan;
  const int grid = plan_wgs(plan, n_rows, row0, n_points, n_features, 256, 8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(rowmean_stats_kernel, dim3(grid), dim3(256), 0, st, d_rowmean, plan, static_cast<double *>(d_workspace));
  SPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(rowstats_finalize_kernel, dim3(n_features), dim3(64), 0, st,
                     static_cast<const double *>(d_workspace), plan, d_fstats);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

extern "C" size_t spr_gram_cross_workspace(int32_t m, int32_t n_features) {
  const int cus = spr_cached_cus();
  const int ntj = (m - CMA + 15) / 16;
  const size_t pairs = (size_t)(cus > 0 ? cus : 256) + (size_t)n_features;
  return pairs * 16 * (size_t)(ntj > 0 ? ((ntj + 3) / 4) * 4 : 4) * 256 * sizeof(double);
}

template <typename TX>
static int gram_cross_entry(const char *who, const TX *d_X, int64_t n_rows, int32_t col_a, int32_t col_b, int32_t w_b,
                            int32_t ldg, int64_t ldx, int64_t row0, int64_t n_points, int32_t n_features, int32_t center,
                            double *d_rowmean, double *d_gram, void *d_workspace, size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(d_X && d_rowmean && d_gram && d_workspace, SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(n_rows > 0 && col_a >= 0 && col_b >= col_a + CMA && w_b > 0 && w_b <= CMA && ldx >= col_b + w_b &&
                  ldg >= col_b + w_b && row0 >= 0 && n_points > 0 && n_features > 0 &&
                  row0 + n_rows <= n_points * (int64_t)n_features,
              SPR_E_INVALID, "%s: bad shape (A = 256 columns at col_a, B = 1..256 columns at col_b >= col_a + 256)", who);
  SPR_REQUIRE(center >= 0 && center <= 2, SPR_E_INVALID, "%s: centre mode must be 0, 1 (own means, written) or 2 (external means)", who);
  SPR_REQUIRE(center != 1 || (col_a == 0 && col_b == CMA && ldg == CMA + w_b), SPR_E_INVALID,
              "%s: centre mode 1 forms the means of the panel's rows: the panel must be the whole row", who);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ntj = (w_b + 15) / 16;
  const int64_t gap = (int64_t)col_b - col_a - CMA;
#define GX(N) return launch_cross<N, TX>(d_X + col_a, n_rows, w_b, ldx, row0, n_points, n_features, center, d_rowmean, d_gram, \
                                         d_workspace, workspace_bytes, st, gap, ldg, col_a, col_b)
  if (ntj <= 4) GX(4);
  if (ntj <= 8) GX(8);
  if (ntj <= 12) GX(12);
  GX(16);
#undef GX
}

extern "C" int spr_gram_cross_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                  int64_t n_points, int32_t n_features, int32_t center, double *d_rowmean,
                                  double *d_gram, void *d_workspace, size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(m > CMA && m <= 2 * CMA, SPR_E_INVALID, "spr_gram_cross_f64: m must be in (256, 512]");
  return gram_cross_entry("spr_gram_cross_f64", d_X, n_rows, 0, CMA, m - CMA, m, ldx, row0, n_points, n_features, center,
                          d_rowmean, d_gram, d_workspace, workspace_bytes, stream);
}

extern "C" int spr_gram_cross_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                  int64_t n_points, int32_t n_features, int32_t center, double *d_rowmean,
                                  double *d_gram, void *d_workspace, size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(m > CMA && m <= 2 * CMA, SPR_E_INVALID, "spr_gram_cross_x32: m must be in (256, 512]");
  return gram_cross_entry("spr_gram_cross_x32", d_X, n_rows, 0, CMA, m - CMA, m, ldx, row0, n_points, n_features, center,
                          d_rowmean, d_gram, d_workspace, workspace_bytes, stream);
}

// any two column slices: A = [col_a, col_a + 256), B = [col_b, col_b + w_b) of rows that are ldx wide; the block
// (col_a.., col_b..) and its mirror go into per-feature ldg x ldg matrices.  centre 0 / 2 (external means) only.
extern "C" int spr_gram_cross_pair_f64(const double *d_X, int64_t n_rows, int32_t col_a, int32_t col_b, int32_t w_b,
                                       int32_t ldg, int64_t ldx, int64_t row0, int64_t n_points, int32_t n_features,
                                       int32_t center, double *d_rowmean, double *d_gram, void *d_workspace,
                                       size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(center != 1, SPR_E_INVALID, "spr_gram_cross_pair_f64: centre mode 0 or 2");
  return gram_cross_entry("spr_gram_cross_pair_f64", d_X, n_rows, col_a, col_b, w_b, ldg, ldx, row0, n_points, n_features,
                          center, d_rowmean, d_gram, d_workspace, workspace_bytes, stream);
}

extern "C" int spr_gram_cross_pair_x32(const float *d_X, int64_t n_rows, int32_t col_a, int32_t col_b, int32_t w_b,
                                       int32_t ldg, int64_t ldx, int64_t row0, int64_t n_points, int32_t n_features,
                                       int32_t center, double *d_rowmean, double *d_gram, void *d_workspace,
                                       size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(center != 1, SPR_E_INVALID, "spr_gram_cross_pair_x32: centre mode 0 or 2");
  return gram_cross_entry("spr_gram_cross_pair_x32", d_X, n_rows, col_a, col_b, w_b, ldg, ldx, row0, n_points, n_features,
                          center, d_rowmean, d_gram, d_workspace, workspace_bytes, stream);
}

// ---- wide X (256 < m <= 512) without a pass for the full-row means ------------------------------------------------
// The Gram matrix of row-centred data is P G_s P, P = I - 1 1^T / m, for the Gram matrix G_s of rows shifted by ANY
// per-row constant c_i (P annihilates constants), and the cancellation in P G_s P is harmless while |mean_i - c_i| stays
// within a few standard deviations of the row.  So the three launches of the wide path all centre with c_i = the mean of
// the row's FIRST 256 columns -- which the first symmetric launch forms anyway (centre mode 1 on its slice) -- and the
// cross block no longer sums 512 columns per row in both of its workgroup flavours.  What is left to do afterwards:
//   mean_i = (wA c_i + sB_i) / m   with sB_i the raw sum of the row's other columns (spr_stats_gram_shifted_*), and
//   G_f <- P G_f P  for every feature's m x m matrix.
namespace {

__global__ __launch_bounds__(256) void shift_means_kernel(double *__restrict__ rowmean, const double *__restrict__ rowsum_b,
                                                          int64_t n_rows, double w_a, double inv_m) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * 256)
    rowmean[i] = (w_a * rowmean[i] + rowsum_b[i]) * inv_m;
}

// one workgroup per feature: row sums r_j of G (fixed order), t = sum of all entries, G_jk <- G_jk - (r_j + r_k)/m + t/m^2
__global__ __launch_bounds__(1024) void gram_pgp_kernel(double *__restrict__ gram, int m) {
  __shared__ double rs[SPR_MAX_M_WIDE];
  __shared__ double tot;
  double *G = gram + (int64_t)blockIdx.x * m * m;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < m; j += 16) {                        // G is symmetric: row sums = column sums
    double s = 0.0;
    for (int k = lane; k < m; k += 64) s += G[(int64_t)j * m + k];
    s = group_sum_t<64>(s);
    if (lane == 0) rs[j] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int j = 0; j < m; ++j) t += rs[j];
    tot = t;
  }
  __syncthreads();
  const double inv_m = 1.0 / (double)m, tt = tot * inv_m * inv_m;
  for (int64_t e = threadIdx.x; e < (int64_t)m * m; e += 1024) {
    const int j = (int)(e / m), k = (int)(e - (int64_t)j * m);
    G[e] = G[e] - (rs[j] + rs[k]) * inv_m + tt;
  }
}

}  // namespace

// d_rowmean: in the means of the first w_a columns (the shift the three launches used), out the means of the full rows;
// d_gram: in the F shifted Gram matrices (m x m each), out the centred ones.
extern "C" int spr_gram_shift_finish_f64(double *d_rowmean, const double *d_rowsum_b, int64_t n_rows, int32_t w_a,
                                         int32_t m, double *d_gram, int32_t n_features, void *stream) {
  SPR_REQUIRE(d_rowmean && d_rowsum_b && d_gram, SPR_E_INVALID, "spr_gram_shift_finish_f64: NULL pointer");
  SPR_REQUIRE(n_rows > 0 && w_a > 0 && m > w_a && m <= SPR_MAX_M_WIDE && n_features > 0, SPR_E_INVALID,
              "spr_gram_shift_finish_f64: bad shape w_a=%d m=%d", w_a, m);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t blocks = (n_rows + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(shift_means_kernel, dim3((int)blocks), dim3(256), 0, st, d_rowmean, d_rowsum_b, n_rows, (double)w_a,
                     1.0 / (double)m);
  SPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(gram_pgp_kernel, dim3(n_features), dim3(1024), 0, st, d_gram, (int)m);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}
