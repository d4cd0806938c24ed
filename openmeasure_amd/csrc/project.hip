// K4: basis projection  Ur = ((X - rowmean) . W) * (1/X_scl),  W = V_r Sigma_r^-1  (m x r).
//
// Second (and last) read of the snapshot shard.  Same persistent-workgroup / row-panel
// staging as the Gram kernel (rowtile.hpp); the centred panel goes to LDS, W lives in
// registers for the whole kernel: wave (cg, rg) keeps the MFMA B fragments of output
// columns [16 cg, 16 cg + 16) for every k (m/4 doubles per lane) and multiplies them with
// the 16-row blocks rg, rg+RG, ... of each panel.
//
// v_mfma_f64_16x16x4_f64 operands: A[i = l&15][k = l>>4] = panel[16 rb + i][k0 + k] -- a
// strided LDS read, conflict-free because the row stride MP is 2 (mod 4) doubles;
// B[k = l>>4][j = l&15] = W[k0 + k][16 cg + j].  Result: col = l&15, row = (l>>4) + 4 reg.
#include "rowtile.hpp"

namespace {

constexpr int NW = 8;

template <int MT> struct ProjRows { static constexpr int R = (MT >= 12) ? 32 : 64; };

template <int MT, int RTILES>
__global__ __launch_bounds__(NW * 64) void project_kernel(
    const double *__restrict__ X, int64_t ldx, int m, int vec_ok_i, int center_i, SegPlan plan,
    const double *__restrict__ inv_scale, const double *__restrict__ W, int r,
    double *__restrict__ Ur, int64_t ldu) {
  constexpr int R = ProjRows<MT>::R;
  constexpr int MPAD = 16 * MT, MP = MPAD + 2;
  constexpr int KSTEPS = MPAD / 4;
  constexpr int CG = RTILES, RG = NW / CG, RB = R / 16;
  using RT = RowTile<MT, R, MP, NW>;

  __shared__ double lds[2][R * MP];

  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = wave % CG, rg = wave / CG;
  const bool vec_ok = vec_ok_i != 0;
  const double isc = inv_scale[f];

  // B fragments of this wave's 16 output columns, all k
  double bfrag[KSTEPS];
  {
    const int col = cg * 16 + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int k = 4 * ks + (lane >> 4);
      bfrag[ks] = (k < m && col < r) ? W[(int64_t)k * r + col] : 0.0;
    }
  }

  RT tile;
  const int64_t nchunks = (hi - lo + R - 1) / R;
  int64_t c = wl;
  if (c < nchunks) tile.load(X, ldx, m, vec_ok, lo + c * R, hi, wave, lane);
  int buf = 0;
  const int afrag = (lane & 15) * MP + (lane >> 4);
  while (c < nchunks) {
    tile.template center_store<false>(lds[buf], m, center_i != 0, lo + c * R, hi, wave, lane, nullptr, nullptr);
    const int64_t cn = c + wpf;
    if (cn < nchunks) tile.load(X, ldx, m, vec_ok, lo + cn * R, hi, wave, lane);
    __syncthreads();
    for (int rb = rg; rb < RB; rb += RG) {
      const double *p = lds[buf] + rb * 16 * MP + afrag;
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(p[4 * ks], bfrag[ks], acc, 0, 0, 0);
      const int col = cg * 16 + (lane & 15);
      const int64_t row = lo + c * R + rb * 16 + (lane >> 4);
      if (col < r) {
        if (row < hi) Ur[row * ldu + col] = acc.x * isc;
        if (row + 4 < hi) Ur[(row + 4) * ldu + col] = acc.y * isc;
        if (row + 8 < hi) Ur[(row + 8) * ldu + col] = acc.z * isc;
        if (row + 12 < hi) Ur[(row + 12) * ldu + col] = acc.w * isc;
      }
    }
    buf ^= 1;
    c = cn;
  }
}

template <int MT, int RTILES>
int launch(const double *X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0, int64_t n_points,
           int32_t n_features, int center, const double *inv_scale, const double *W, int32_t r, double *Ur, int64_t ldu,
           hipStream_t st) {
  static int total_wg = 0;
  if (!total_wg) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, project_kernel<MT, RTILES>, NW * 64, 0) !=
            hipSuccess || per_cu < 1)
      per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    const int cus = spr_cached_cus();
    total_wg = per_cu * (cus > 0 ? cus : 256);
  }
  SegPlan plan;
  plan.row0 = row0; plan.n_rows = n_rows; plan.n_points = n_points; plan.n_features = n_features;
  plan.total_wg = total_wg; plan.chunk_rows = ProjRows<MT>::R;
  const int grid = seg_total_wgs(plan);
  const int vec_ok = (m % 2 == 0) && (ldx % 2 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  hipLaunchKernelGGL((project_kernel<MT, RTILES>), dim3(grid), dim3(NW * 64), 0, st, X, ldx, (int)m, vec_ok,
                     center, plan, inv_scale, W, (int)r, Ur, ldu);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

template <int MT>
int launch_rt(int rt, const double *X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0, int64_t n_points,
              int32_t n_features, int center, const double *inv_scale, const double *W, int32_t r, double *Ur, int64_t ldu,
              hipStream_t st) {
  switch (rt) {
    case 1: return launch<MT, 1>(X, n_rows, m, ldx, row0, n_points, n_features, center, inv_scale, W, r, Ur, ldu, st);
    case 2: return launch<MT, 2>(X, n_rows, m, ldx, row0, n_points, n_features, center, inv_scale, W, r, Ur, ldu, st);
    case 4: return launch<MT, 4>(X, n_rows, m, ldx, row0, n_points, n_features, center, inv_scale, W, r, Ur, ldu, st);
    case 8: return launch<MT, 8>(X, n_rows, m, ldx, row0, n_points, n_features, center, inv_scale, W, r, Ur, ldu, st);
  }
  spr_set_error("spr_project_f64: r tile count %d not built", rt);
  return SPR_E_UNSUPPORTED;
}

}  // namespace

extern "C" int spr_project_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                               int64_t n_points, int32_t n_features, int32_t center,
                               const double *d_inv_scale, const double *d_W, int32_t r, double *d_Ur, int64_t ldu, void *stream) {
  SPR_REQUIRE(d_X && d_inv_scale && d_W && d_Ur, SPR_E_INVALID, "spr_project_f64: NULL pointer");
  SPR_REQUIRE(n_rows > 0 && m > 0 && ldx >= m, SPR_E_INVALID, "spr_project_f64: bad shape");
  SPR_REQUIRE(r > 0 && r <= m && ldu >= r, SPR_E_INVALID, "spr_project_f64: bad r=%d (m=%d ldu=%lld)", r, m,
              (long long)ldu);
  SPR_REQUIRE(n_points > 0 && n_features > 0 && row0 >= 0 &&
                  row0 + n_rows <= n_points * (int64_t)n_features,
              SPR_E_INVALID, "spr_project_f64: bad feature layout");
  SPR_REQUIRE(m <= SPR_MAX_M && r <= SPR_MAX_R, SPR_E_UNSUPPORTED, "spr_project_f64: m=%d r=%d not built", m, r);
  const int need = (r + 15) / 16;
  const int rt = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define PJ(MTV) return launch_rt<MTV>(rt, d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale, d_W, r, d_Ur, ldu, st)
  switch (spr_round_mt(m)) {
    case 1: PJ(1);
    case 2: PJ(2);
    case 3: PJ(3);
    case 4: PJ(4);
    case 6: PJ(6);
    case 8: PJ(8);
    case 12: PJ(12);
    case 16: PJ(16);
  }
#undef PJ
  spr_set_error("spr_project_f64: m=%d not built", m);
  return SPR_E_UNSUPPORTED;
}
