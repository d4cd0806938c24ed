// K4: basis projection  Ur = ((X - rowmean) . W) * (1/X_scl),  W = V_r Sigma_r^-1  (m x r).
//
// Second (and last) read of the snapshot shard.  Same persistent-workgroup / row-panel
// staging as the Gram kernel (rowtile.hpp), but the panel goes to LDS RAW: VALU work does
// not overlap with v_mfma_f64 on gfx950 (tools/coexec_probe.hip: time = MFMA + VALU), so the
// centring is folded into the epilogue instead,
//     (x_i - mu_i 1) . W = x_i . W - mu_i * (1^T W),
// with mu_i read back from the row-mean array of the Gram pass (8 bytes per row) and the
// column sums of W formed once per wave.  The products x_i . W carry the row mean through the
// MFMA, so the result is accurate to eps * |mu|/|x - mu| * sqrt(m) relative -- the same order
// as the reference's own X0 = X - mean (whose mean is only known to eps * |mu|).
// W lives in registers for the whole kernel: wave (cg, rg) keeps the MFMA B fragments of output
// columns [16 cg, 16 cg + 16) for every k (m/4 doubles per lane) and multiplies them with
// the 16-row blocks rg, rg+RG, ... of each panel.
//
// v_mfma_f64_16x16x4_f64 operands: A[i = l&15][k = l>>4] = panel[16 rb + i][k0 + k] -- a
// strided ds_read_b64; with the row stride MP = MPAD + 2 (2 mod 4 doubles) the 32 lanes of a
// group fall on 32 distinct bank pairs.  B[k = l>>4][j = l&15] = W[k0 + k][16 cg + j].
// Result: col = l&15, row = (l>>4) + 4 reg.
#include <stdlib.h>

#include <type_traits>

#include "project_ws.hpp"
#include "rowtile.hpp"

#ifndef PROJ_PAD
#define PROJ_PAD 2   // row stride MPAD+2: ds_read_b64 A fragments conflict-free, rows 16-byte aligned (a sweep over 1..18 changed the kernel time by < 3 %)
#endif
#ifndef PROJ_ABLATE
#define PROJ_ABLATE 0   // diagnostic builds: 1 = no MFMAs, 2 = no centring/loads in the loop, 3 = no Ur stores
#endif
#if PROJ_ABLATE == 1
#define PROJ_MFMA(a, b, c) ({ asm volatile("" ::"v"(a), "v"(b)); (c); })
#else
#define PROJ_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#endif

namespace {

constexpr int NW = 8;

template <int MT> struct ProjRows { static constexpr int R = (MT >= 12) ? 32 : 64; };

// TX: storage type of the snapshot shard, TU: storage type of the basis -- f64 (the reference's U is float64 for any
// dtype of X, :106-107, :169) or, as a storage option, f32 rounded once from the f64 result; the arithmetic is f64 either way.
template <int MT, int RTILES, int VEC, typename TX, typename TU, bool ACCIN>
__global__ __launch_bounds__(NW * 64) void project_kernel(
    const TX *__restrict__ X, int64_t ldx, int m, int center_i, SegPlan plan,
    const double *__restrict__ inv_scale, const double *__restrict__ rowmean, const double *__restrict__ W, int r,
    TU *__restrict__ Ur, int64_t ldu, int accumulate, const double *__restrict__ acc_in, int64_t lda) {
  constexpr int R = ProjRows<MT>::R;
  constexpr int MPAD = 16 * MT, MP = MPAD + PROJ_PAD;
  constexpr int KSTEPS = MPAD / 4;
  constexpr int CG = RTILES, RG = NW / CG, RB = R / 16;
  using RT = RowTile<MT, R, MP, NW, 16, TX>;

  __shared__ double lds[2][R * MP];

  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cg = wave % CG, rg = wave / CG;
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
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(bfrag[ks]));  // pin: never re-load W inside the loop
  }
  // column sum of W for this lane's output column (centring term), 0 when the rows are used as they are
  double wbar = 0.0;
  if (center_i != 0) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) wbar += bfrag[ks];
    wbar += __shfl_xor(wbar, 16, 64);
    wbar += __shfl_xor(wbar, 32, 64);
  }

  // Same one-barrier software pipeline as the Gram kernel: the MFMAs of panel c run from one
  // LDS buffer while this wave centres panel c+1 into the other buffer (one row pass per slice
  // of the k loop) and re-issues the HBM loads of panel c+2 into the registers just freed.
  RT tile;
  const int64_t nchunks = (hi - lo + R - 1) / R;
  int64_t c = wl;
  tile.template load<VEC>(X, ldx, m, lo + c * R, hi, wave, lane);
  tile.raw_store(lds[0], m, lo + c * R, hi, wave, lane);
  int64_t cn = c + wpf;
  int64_t nrow0 = (cn < nchunks) ? lo + cn * R : hi;
  tile.template load<VEC>(X, ldx, m, nrow0, hi, wave, lane);
  int buf = 0;
  const int afrag = (lane & 15) * MP + (lane >> 4);
  const int col = cg * 16 + (lane & 15);
  constexpr int NRB = (RB + RG - 1) / RG;           // row blocks per wave and panel (upper bound)
  constexpr int SLOTS = NRB * KSTEPS;               // MFMAs per wave and panel
  constexpr int NPIECE = RT::IT * RT::VPL;          // 16-byte pieces per lane and panel
  constexpr bool ALL_LIVE = (RB % RG == 0);         // every (wave, b) pair has a row block
  while (c < nchunks) {
    double *cur = lds[buf];
    double *nxt = lds[buf ^ 1];
    const int64_t c2 = cn + wpf;
    const int64_t n2row0 = (c2 < nchunks) ? lo + c2 * R : hi;
    __syncthreads();
#pragma clang loop unroll(full)
    for (int b = 0; b < NRB; ++b) {
      const int rb = rg + b * RG;
      // One wave-uniform branch per row block, never one per MFMA: a conditional around each
      // MFMA made hipcc wait (lgkmcnt(0)) for every operand right before its use.
      auto body = [&](auto live_tag, auto full_tag) {
        constexpr bool LIVE = decltype(live_tag)::value;
        constexpr bool FULLP = decltype(full_tag)::value;   // whole panel inside the segment and r == 16*RTILES
        const double *p = cur + (LIVE ? rb : 0) * 16 * MP + afrag;
        f64x4 acc0 = {0.0, 0.0, 0.0, 0.0};
        const f64x4 acc1 = {0.0, 0.0, 0.0, 0.0};
        // row means of this block's four output rows per lane, requested before the MFMAs so their
        // latency is covered.  Unconditional, clamped loads: a branch here splits the block and hipcc
        // then waits vmcnt(0) -- i.e. for these very loads -- before the first staging store.
        const int64_t mrow = lo + c * R + (LIVE ? rb : 0) * 16 + (lane >> 4);
        double mu[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t rr = mrow + 4 * i;
          mu[i] = rowmean[rr < hi ? rr : hi - 1];
        }
#pragma clang loop unroll(full)
        for (int ks = 0; ks < KSTEPS; ++ks) {
          if (LIVE) {
            acc0 = PROJ_MFMA(p[4 * ks], bfrag[ks], acc0);   // one chain: a dependent f64 MFMA issues back to back
          }
          if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound how far ahead operands are fetched (VGPRs)
#pragma clang loop unroll(full)
          for (int pc = 0; pc < NPIECE; ++pc)        // piece pc = (pass, 16-byte piece): LDS store, then reload
            if (PROJ_ABLATE != 2 && (pc * SLOTS) / NPIECE == b * KSTEPS + ks) {
              tile.raw_store_piece(pc / RT::VPL, pc % RT::VPL, nxt, m, nrow0, hi, wave, lane);
              tile.template load_piece<VEC>(pc / RT::VPL, pc % RT::VPL, X, ldx, m, n2row0, hi, wave, lane);
            }
        }
        if (LIVE) {
          const int64_t row = mrow;
          const double s0 = (acc0.x + acc1.x - mu[0] * wbar) * isc, s1 = (acc0.y + acc1.y - mu[1] * wbar) * isc;
          const double s2 = (acc0.z + acc1.z - mu[2] * wbar) * isc, s3 = (acc0.w + acc1.w - mu[3] * wbar) * isc;
          if (PROJ_ABLATE == 3) asm volatile("" ::"v"(s0), "v"(s1), "v"(s2), "v"(s3));
          if (PROJ_ABLATE != 3) {
            if (FULLP && !accumulate) {                        // no predicates: one basic block per panel
              Ur[row * ldu + col] = (TU)s0;
              Ur[(row + 4) * ldu + col] = (TU)s1;
              Ur[(row + 8) * ldu + col] = (TU)s2;
              Ur[(row + 12) * ldu + col] = (TU)s3;
            } else if (col < r) {                              // accumulate: second column slice of a wide X
              // accumulate: the earlier slices' partial sum comes from Ur itself, or from a separate f64 block
              // (acc_in) when Ur is stored narrower than the partial sums may be rounded to
              auto prev = [&](int64_t rr) {
                return ACCIN ? acc_in[rr * lda + col] : (double)Ur[rr * ldu + col];
              };
              if (row < hi) Ur[row * ldu + col] = (TU)((accumulate ? prev(row) : 0.0) + s0);
              if (row + 4 < hi) Ur[(row + 4) * ldu + col] = (TU)((accumulate ? prev(row + 4) : 0.0) + s1);
              if (row + 8 < hi) Ur[(row + 8) * ldu + col] = (TU)((accumulate ? prev(row + 8) : 0.0) + s2);
              if (row + 12 < hi) Ur[(row + 12) * ldu + col] = (TU)((accumulate ? prev(row + 12) : 0.0) + s3);
            }
          }
        }
      };
      const bool fullp = (lo + (c + 1) * R <= hi) && (r == 16 * RTILES);
      if (ALL_LIVE || rb < RB) {
        if (fullp) body(std::true_type{}, std::true_type{});
        else body(std::true_type{}, std::false_type{});
      } else {
        body(std::false_type{}, std::false_type{});
      }
    }
    buf ^= 1;
    c = cn;
    cn = c2;
    nrow0 = n2row0;
  }
}

template <int MT, int RTILES, typename TX, typename TU>
int launch(const TX *X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0, int64_t n_points,
           int32_t n_features, int center, const double *inv_scale, const double *rowmean, const double *W, int32_t r,
           TU *Ur, int64_t ldu, int accumulate, const double *acc_in, int64_t lda, hipStream_t st) {
  static int total_wg = 0;
  if (!total_wg) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, project_kernel<MT, RTILES, 2, TX, TU, false>, NW * 64, 0) !=
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
  const int vec_ok = (m % 2 == 0) && (ldx % 2 == 0) && ((reinterpret_cast<uintptr_t>(X) & (2 * sizeof(TX) - 1)) == 0);
  const int lm = vec_ok ? ((m == 16 * MT) ? 2 : 1) : 0;
  // the separate f64 partial-sum input only exists for an f32 basis; every other instantiation keeps its register budget
  constexpr bool CAN_ACC = std::is_same<TU, float>::value;
#define PJ_LAUNCH(LM)                                                                                         \
  do {                                                                                                        \
    if (CAN_ACC && acc_in)                                                                                    \
      hipLaunchKernelGGL((project_kernel<MT, RTILES, LM, TX, TU, CAN_ACC>), dim3(grid), dim3(NW * 64), 0, st, X, ldx, (int)m, \
                         center, plan, inv_scale, rowmean, W, (int)r, Ur, ldu, accumulate, acc_in, lda);      \
    else                                                                                                      \
      hipLaunchKernelGGL((project_kernel<MT, RTILES, LM, TX, TU, false>), dim3(grid), dim3(NW * 64), 0, st, X, ldx, (int)m, \
                         center, plan, inv_scale, rowmean, W, (int)r, Ur, ldu, accumulate, acc_in, lda);      \
  } while (0)
  if (lm == 2) PJ_LAUNCH(2);
  else if (lm == 1) PJ_LAUNCH(1);
  else PJ_LAUNCH(0);
#undef PJ_LAUNCH
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

template <int MT, typename TX, typename TU>
int launch_rt(int rt, const TX *X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0, int64_t n_points,
              int32_t n_features, int center, const double *inv_scale, const double *rowmean, const double *W, int32_t r,
              TU *Ur, int64_t ldu, int accumulate, const double *acc_in, int64_t lda, hipStream_t st) {
  switch (rt) {
    case 1: return launch<MT, 1, TX, TU>(X, n_rows, m, ldx, row0, n_points, n_features, center, inv_scale, rowmean, W, r, Ur, ldu, accumulate, acc_in, lda, st);
    case 2: return launch<MT, 2, TX, TU>(X, n_rows, m, ldx, row0, n_points, n_features, center, inv_scale, rowmean, W, r, Ur, ldu, accumulate, acc_in, lda, st);
    case 4: return launch<MT, 4, TX, TU>(X, n_rows, m, ldx, row0, n_points, n_features, center, inv_scale, rowmean, W, r, Ur, ldu, accumulate, acc_in, lda, st);
    case 8: return launch<MT, 8, TX, TU>(X, n_rows, m, ldx, row0, n_points, n_features, center, inv_scale, rowmean, W, r, Ur, ldu, accumulate, acc_in, lda, st);
  }
  spr_set_error("spr_project_f64: r tile count %d not built", rt);
  return SPR_E_UNSUPPORTED;
}

template <typename TX, typename TU>
int project_entry(const char *who, const TX *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0, int64_t n_points,
                  int32_t n_features, int32_t center, const double *d_inv_scale, const double *d_rowmean,
                  const double *d_W, int32_t r, TU *d_Ur, int64_t ldu, int32_t accumulate, void *stream,
                  const double *d_acc_in = nullptr, int64_t lda = 0, double *d_rownorm2 = nullptr) {
  SPR_REQUIRE(!d_acc_in || (accumulate && lda >= r), SPR_E_INVALID, "%s: acc_in needs accumulate = 1 and lda >= r", who);
  SPR_REQUIRE(d_X && d_inv_scale && d_W && d_Ur && (d_rowmean || !center), SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(n_rows > 0 && m > 0 && ldx >= m, SPR_E_INVALID, "%s: bad shape", who);
  SPR_REQUIRE(r > 0 && ldu >= r, SPR_E_INVALID, "%s: bad r=%d (m=%d ldu=%lld)", who, r, m, (long long)ldu);
  if (!center && !d_rowmean) {
    // the kernel reads a row mean unconditionally (clamped, in range) and multiplies it by a zero column sum:
    // any buffer of n_rows doubles will do
    if (ldx * sizeof(TX) >= sizeof(double)) d_rowmean = reinterpret_cast<const double *>(d_X);
    else if (ldu * sizeof(TU) >= sizeof(double)) d_rowmean = reinterpret_cast<const double *>(d_Ur);
    SPR_REQUIRE(d_rowmean, SPR_E_INVALID, "%s: center=0 on single-column f32 data needs a row-mean buffer", who);
  }
  SPR_REQUIRE(n_points > 0 && n_features > 0 && row0 >= 0 &&
                  row0 + n_rows <= n_points * (int64_t)n_features,
              SPR_E_INVALID, "%s: bad feature layout", who);
  SPR_REQUIRE(m <= SPR_MAX_M && r <= SPR_MAX_R, SPR_E_UNSUPPORTED, "%s: m=%d r=%d not built", who, m, r);
  const int need = (r + 15) / 16;
  const int rt = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // W-stationary form first (project_ws.hip): W in LDS, X straight into the MFMA operand registers, no barrier in the
  // loop -- 15 % faster where both pipes are loaded (m = 256, r = 64).  SPR_PROJECT_WS=0 forces the general kernel
  // (A/B measurements, and the parity tests run both).
  if constexpr (std::is_same<TX, TU>::value || std::is_same<TU, double>::value) {
    static const bool ws_on = [] { const char *e = getenv("SPR_PROJECT_WS"); return !(e && e[0] == '0'); }();
    if (ws_on && !d_acc_in && n_rows >= 4096) {
      const int rc = spr_project_ws<TX, TU>(d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,
                                            d_rowmean, d_W, r, d_Ur, ldu, accumulate, d_rownorm2, st);
      if (rc != SPR_E_UNSUPPORTED) return rc;
    }
  }
  // the general kernel spreads a row's columns over several waves: no row norms from it (spr_project_stream_norms_*
  // takes every shape)
  SPR_REQUIRE(!d_rownorm2, SPR_E_UNSUPPORTED, "%s: row norms only come from the W-stationary form (m = 64/128/192/256 packed, "
              "r <= 64); use spr_project_stream_norms_* for m=%d r=%d", who, m, r);
#define PJ(MTV) return launch_rt<MTV, TX, TU>(rt, d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale, d_rowmean, d_W, r, d_Ur, ldu, accumulate, d_acc_in, lda, st)
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
  spr_set_error("%s: m=%d not built", who, m);
  return SPR_E_UNSUPPORTED;
}

}  // namespace

extern "C" int spr_project_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                               int64_t n_points, int32_t n_features, int32_t center,
                               const double *d_inv_scale, const double *d_rowmean, const double *d_W, int32_t r,
                               double *d_Ur, int64_t ldu, int32_t accumulate, void *stream) {
  return project_entry("spr_project_f64", d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,
                       d_rowmean, d_W, r, d_Ur, ldu, accumulate, stream);
}

extern "C" int spr_project_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                               int64_t n_points, int32_t n_features, int32_t center,
                               const double *d_inv_scale, const double *d_rowmean, const double *d_W, int32_t r,
                               float *d_Ur, int64_t ldu, int32_t accumulate, void *stream) {
  return project_entry("spr_project_x32", d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,
                       d_rowmean, d_W, r, d_Ur, ldu, accumulate, stream);
}

// f32 shard, f64 result: the partial sums of a wide X (m > 256, two column slices) cancel by up to sigma_1/sigma_r
// between the slices, so they are accumulated in an f64 scratch block and rounded to f32 once by the caller
extern "C" int spr_project_x32_f64out(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                      int64_t n_points, int32_t n_features, int32_t center,
                                      const double *d_inv_scale, const double *d_rowmean, const double *d_W,
                                      int32_t r, double *d_Ur, int64_t ldu, int32_t accumulate, void *stream) {
  return project_entry("spr_project_x32_f64out", d_X, n_rows, m, ldx, row0, n_points, n_features, center,
                       d_inv_scale, d_rowmean, d_W, r, d_Ur, ldu, accumulate, stream);
}

// last column slice of a wide f32 shard: adds the f64 partial sums of the earlier slices (d_acc_in, row stride lda) to
// this slice's product and stores the total, rounded to f32 ONCE, in d_Ur -- no separate conversion pass
extern "C" int spr_project_x32_acc(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                   int64_t n_points, int32_t n_features, int32_t center,
                                   const double *d_inv_scale, const double *d_rowmean, const double *d_W, int32_t r,
                                   const double *d_acc_in, int64_t lda, float *d_Ur, int64_t ldu, void *stream) {
  SPR_REQUIRE(d_acc_in != nullptr, SPR_E_INVALID, "spr_project_x32_acc: acc_in is NULL");
  return project_entry("spr_project_x32_acc", d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,
                       d_rowmean, d_W, r, d_Ur, ldu, 1, stream, d_acc_in, lda);
}

// ---- K4 + first sweep of K6: the projection that also leaves the squared norms of the rows it stores --------------
// d_rownorm2[n_rows] = |Ur[i, :]|^2 of the STORED values (rounded to the basis type first), for spr_qr_init_norms_*.
// Only the W-stationary form produces them: SPR_E_UNSUPPORTED (and nothing launched) for any other shape -- ask
// spr_project_norms_supported first, or call spr_project_stream_norms_*, which takes every shape.
extern "C" int32_t spr_project_norms_supported(int32_t m, int32_t r, int64_t n_rows, int64_t ldx, const void *d_X,
                                               int32_t x_is_f32) {
  const char *e = getenv("SPR_PROJECT_WS");
  if (e && e[0] == '0') return 0;
  const size_t es = x_is_f32 ? sizeof(float) : sizeof(double);
  return (m == 64 || m == 128 || m == 192 || m == 256) && r >= 1 && r <= 64 && n_rows >= 4096 && (es * ldx) % 16 == 0 &&
         (reinterpret_cast<uintptr_t>(d_X) & 15) == 0;
}

extern "C" int spr_project_norms_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                     int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                                     const double *d_rowmean, const double *d_W, int32_t r, double *d_Ur, int64_t ldu,
                                     double *d_rownorm2, void *stream) {
  SPR_REQUIRE(d_rownorm2, SPR_E_INVALID, "spr_project_norms_f64: NULL norm vector");
  return project_entry("spr_project_norms_f64", d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,
                       d_rowmean, d_W, r, d_Ur, ldu, 0, stream, nullptr, 0, d_rownorm2);
}

extern "C" int spr_project_norms_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                     int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                                     const double *d_rowmean, const double *d_W, int32_t r, float *d_Ur, int64_t ldu,
                                     double *d_rownorm2, void *stream) {
  SPR_REQUIRE(d_rownorm2, SPR_E_INVALID, "spr_project_norms_x32: NULL norm vector");
  return project_entry("spr_project_norms_x32", d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,
                       d_rowmean, d_W, r, d_Ur, ldu, 0, stream, nullptr, 0, d_rownorm2);
}

extern "C" int spr_project_norms_x32_f64out(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                            int64_t n_points, int32_t n_features, int32_t center,
                                            const double *d_inv_scale, const double *d_rowmean, const double *d_W,
                                            int32_t r, double *d_Ur, int64_t ldu, double *d_rownorm2, void *stream) {
  SPR_REQUIRE(d_rownorm2, SPR_E_INVALID, "spr_project_norms_x32_f64out: NULL norm vector");
  return project_entry("spr_project_norms_x32_f64out", d_X, n_rows, m, ldx, row0, n_points, n_features, center,
                       d_inv_scale, d_rowmean, d_W, r, d_Ur, ldu, 0, stream, nullptr, 0, d_rownorm2);
}
