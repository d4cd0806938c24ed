// K4, W-stationary form: basis projection  Ur = ((X - rowmean) . W) * (1/X_scl)  with W resident in LDS.
//
// The general projection kernel (project.hip) keeps W in registers and stages panels of X through LDS behind one
// barrier per panel; at m = 256, r = 64 it needs 50 TFLOP/s and 3.9 TB/s at once and its waves sit in s_waitcnt /
// s_barrier a third of the time (round-1 PMC: SQ_WAIT_ANY 34 %, 40 % of the LDS cycles bank conflicts, MFMA pipe
// busy 71 %).  Whenever W (m x r doubles, padded) fits in LDS -- 128 KB at m = 256, r = 64 -- the roles are swapped:
//
//   * W is loaded into LDS once per workgroup (row stride 16 RT + 4 doubles: the four k-rows a B fragment touches
//     fall on disjoint bank halves, so every ds_read_b64 is conflict free) and is read-only afterwards;
//   * X goes HBM -> registers directly in the MFMA A layout, never through LDS: lane (i = l & 15, kk = l >> 4) of a
//     wave owns row i of the wave's 16-row block and loads the 16-byte pieces [16 j + 4 kk, +4) of it, j = 0..K/16;
//     the contraction index is permuted accordingly (MFMA step 4 j + t multiplies X[:, 16 j + 4 kk + t] with
//     W[16 j + 4 kk + t, :] -- the order of a sum is free), so one wave instruction moves 16 rows x 64 bytes;
//   * every wave is independent: its own rows, its own accumulators for all RT column tiles, no barrier in the
//     loop, no LDS stores, no selects.  The registers of piece j are re-loaded with the NEXT block's piece j right
//     after their last use, so a whole row block (32 KB per wave) is always in flight behind the MFMAs.
//
// v_mfma_f64_16x16x4_f64 operands as in project.hip: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], result
// col = l&15, row = (l>>4) + 4 reg.  The centring is folded into the epilogue, (x - mu 1) W = x W - mu (1^T W).
#include <type_traits>

#include "rowtile.hpp"

namespace {

#ifndef WS_ABLATE
#define WS_ABLATE 0
#endif
constexpr int WS_WAVES = 8;
constexpr int WS_ROWS = 16 * WS_WAVES;   // rows a workgroup consumes per step: one 16-row block per wave

// LDS image of W: row k holds, per PAIR of column tiles p, the 32 doubles [li][ct & 1]; a lane reads the two tiles of a
// pair with one ds_read_b128.  Row stride 16 RT doubles (a multiple of 8): the 16-lane groups a ds_read_b128 is
// served in then cover all 64 banks exactly once (lanes of k-row kk + 1 fill the banks the lanes of kk leave free).
template <int RT> struct WsLds { static constexpr int NC = 16 * RT, LDW = NC; static_assert(RT % 2 == 0, "column tiles come in pairs"); };

template <typename TX> struct WsPiece;                     // four consecutive elements of a row, as loaded
template <> struct WsPiece<double> { f64x2 a, b; };
template <> struct WsPiece<float> { float x, y, z, w; };

template <int VEC, typename TX>
__device__ inline WsPiece<TX> ws_load(const TX *__restrict__ rp, int col0, int m) {
  WsPiece<TX> p;
  if constexpr (std::is_same<TX, double>::value) {
    if (VEC) {
      p.a = *reinterpret_cast<const f64x2 *>(rp + col0);
      p.b = *reinterpret_cast<const f64x2 *>(rp + col0 + 2);
    } else {   // any m / alignment: columns past m re-read column 0 (their W rows are zero in LDS)
      p.a.x = rp[col0 < m ? col0 : 0];         p.a.y = rp[col0 + 1 < m ? col0 + 1 : 0];
      p.b.x = rp[col0 + 2 < m ? col0 + 2 : 0]; p.b.y = rp[col0 + 3 < m ? col0 + 3 : 0];
    }
  } else {
    if (VEC) {
      const float4 v = *reinterpret_cast<const float4 *>(rp + col0);
      p.x = v.x; p.y = v.y; p.z = v.z; p.w = v.w;
    } else {
      p.x = rp[col0 < m ? col0 : 0];         p.y = rp[col0 + 1 < m ? col0 + 1 : 0];
      p.z = rp[col0 + 2 < m ? col0 + 2 : 0]; p.w = rp[col0 + 3 < m ? col0 + 3 : 0];
    }
  }
  return p;
}

template <typename TX>
__device__ inline double ws_elem(const WsPiece<TX> &p, int t) {
  if constexpr (std::is_same<TX, double>::value) return t == 0 ? p.a.x : t == 1 ? p.a.y : t == 2 ? p.b.x : p.b.y;
  else return (double)(t == 0 ? p.x : t == 1 ? p.y : t == 2 ? p.z : p.w);
}

// NRM: the squared norms of the rows of Ur AS STORED (rounded to TU first) also go to nrm2[] -- what the first sweep of
// optimal_placement would otherwise read all of Ur again for (qr_pivot.hip, spr_qr_init_norms_*).
template <int MT, int RT, int VEC, typename TX, typename TU, bool NRM>
__global__ __launch_bounds__(WS_WAVES * 64) void project_ws_kernel(
    const TX *__restrict__ X, int64_t ldx, int m, int center_i, SegPlan plan, const double *__restrict__ inv_scale,
    const double *__restrict__ rowmean, const double *__restrict__ W, int r, TU *__restrict__ Ur, int64_t ldu,
    double *__restrict__ nrm2) {
  constexpr int K = 16 * MT, NJ = MT;                 // NJ pieces of 16 columns per row
  constexpr int NC = WsLds<RT>::NC, LDW = WsLds<RT>::LDW;
  __shared__ double Wl[K * LDW];

  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kk = lane >> 4;
  const double isc = inv_scale[f];

  for (int e = threadIdx.x; e < K * NC; e += WS_WAVES * 64) {
    const int k = e / NC, c = e - k * NC;
    // 1/X_scl of the workgroup's feature is folded into the image: the epilogue is one fused multiply-add per output
    Wl[k * LDW + (c >> 5) * 32 + (c & 15) * 2 + ((c >> 4) & 1)] = (k < m && c < r) ? W[(int64_t)k * r + c] * isc : 0.0;
  }
  __syncthreads();
  // column sums of W for this lane's output columns (centring term)
  double wbar[RT];
#pragma unroll
  for (int ct = 0; ct < RT; ++ct) {
    double sacc = 0.0;
    if (center_i != 0) {
      for (int k = kk; k < K; k += 4) sacc += Wl[k * LDW + (ct >> 1) * 32 + li * 2 + (ct & 1)];
      sacc += __shfl_xor(sacc, 16, 64);
      sacc += __shfl_xor(sacc, 32, 64);
    }
    wbar[ct] = sacc;
  }

  const int64_t nchunks = (hi - lo + WS_ROWS - 1) / WS_ROWS;
  int64_t c = wl;
  if (c >= nchunks) return;
  // B fragments of step (j, t), column tiles 2p and 2p+1: the two doubles at Wl[(16 j + 4 kk + t) * LDW + 32 p + 2 li]
  const double *wb = Wl + (4 * kk) * LDW + 2 * li;
  const double *wb_hi = wb + 128 * LDW;                    // pieces 8..15 (only dereferenced when NJ > 8)

  auto row_ptr = [&](int64_t cc) {
    int64_t row = lo + cc * WS_ROWS + 16 * wave + li;
    row = row < hi ? row : hi - 1;
    return X + row * ldx;
  };
  WsPiece<TX> areg[NJ];
  {
    const TX *rp = row_ptr(c);
#pragma unroll
    for (int j = 0; j < NJ; ++j) areg[j] = ws_load<VEC, TX>(rp, 16 * j + 4 * kk, m);
  }
  while (c < nchunks) {
    const int64_t cn = c + wpf;
    const TX *rpn = row_ptr(cn < nchunks ? cn : c);          // past the end: harmless re-read of the current block
    const int64_t blk0 = lo + c * WS_ROWS + 16 * wave;
    double mu[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t rr = blk0 + kk + 4 * i;
      const double v = rowmean[rr < hi ? rr : hi - 1];   // center = 0: any readable buffer (see project_entry), value unused
      mu[i] = center_i != 0 ? v : 0.0;
    }
    f64x4 acc[RT];
#pragma unroll
    for (int ct = 0; ct < RT; ++ct) acc[ct] = (f64x4){0.0, 0.0, 0.0, 0.0};
    // software pipeline over the 4 NJ steps: the B fragments of step s+1 are requested from LDS BEFORE the RT MFMAs
    // of step s are issued (sched_group_barrier pins that order; a fence per step keeps hipcc from hoisting more reads
    // and their registers).  Two base pointers (pieces 0-7 and 8-15) keep every LDS offset inside the 16-bit immediate.
    f64x2 bcur[RT / 2], bnxt[RT / 2];
#pragma unroll
    for (int p = 0; p < RT / 2; ++p) bcur[p] = *reinterpret_cast<const f64x2 *>(wb + 32 * p);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int sn = 4 * j + t + 1;                          // next step (wraps to 0: a harmless extra read)
        const int jn = (sn >> 2) % NJ, tn = sn & 3;
        const double *wj = (jn < 8 ? wb : wb_hi) + (16 * (jn & 7) + tn) * LDW;
#pragma unroll
        for (int p = 0; p < RT / 2; ++p) bnxt[p] = *reinterpret_cast<const f64x2 *>(wj + 32 * p);
        const double a = ws_elem<TX>(areg[j], t);
#pragma unroll
        for (int p = 0; p < RT / 2; ++p) {
#if WS_ABLATE == 1                                           // diagnostic (wrong results): everything but the MFMAs
          asm volatile("" ::"v"(a), "v"(bcur[p].x), "v"(bcur[p].y));
#else
          acc[2 * p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bcur[p].x, acc[2 * p], 0, 0, 0);
          acc[2 * p + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bcur[p].y, acc[2 * p + 1], 0, 0, 0);
#endif
        }
        if (t == 3) areg[j] = ws_load<VEC, TX>(rpn, 16 * j + 4 * kk, m);   // next block's piece j into the registers just freed
        __builtin_amdgcn_sched_group_barrier(0x100, RT / 2, 0);              // DS reads of the next step first,
        __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);                  // then this step's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < RT / 2; ++p) bcur[p] = bnxt[p];
      }
    }
    {
      const bool full = (blk0 + 16 <= hi) && (r == NC);     // wave-uniform: whole block inside the segment, no padded column
      TU *up = Ur + (blk0 + kk) * ldu + li;                  // element (row 0 of this lane, column tile 0)
      double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ct = 0; ct < RT; ++ct) {
        const double s[4] = {acc[ct].x - mu[0] * wbar[ct], acc[ct].y - mu[1] * wbar[ct],
                             acc[ct].z - mu[2] * wbar[ct], acc[ct].w - mu[3] * wbar[ct]};
        if (NRM) {                                           // padded columns hold exact zeros (zero rows of the image)
#pragma unroll
          for (int i = 0; i < 4; ++i) { const double v = (double)(TU)s[i]; q[i] = fma(v, v, q[i]); }
        }
        if (full) {
#pragma unroll
          for (int i = 0; i < 4; ++i) up[4 * i * ldu + 16 * ct] = (TU)s[i];
        } else if (16 * ct + li < r) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (blk0 + kk + 4 * i < hi) up[4 * i * ldu + 16 * ct] = (TU)s[i];
        }
      }
      if (NRM) {                                             // the 16 lanes of a row hold its 16 RT columns: one DPP butterfly
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double t = group_sum_t<16>(q[i]);
          if (li == i && blk0 + kk + 4 * i < hi) nrm2[blk0 + kk + 4 * i] = t;
        }
      }
    }
    c = cn;
  }
}

template <int MT, int RT, typename TX, typename TU>
int ws_launch(const TX *X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0, int64_t n_points, int32_t n_features,
              int center, const double *inv_scale, const double *rowmean, const double *W, int32_t r, TU *Ur,
              int64_t ldu, int accumulate, double *nrm2, hipStream_t st) {
  const int cus = spr_cached_cus();
  SegPlan plan;
  plan.row0 = row0; plan.n_rows = n_rows; plan.n_points = n_points; plan.n_features = n_features;
  // the LDS image of W allows one workgroup per CU at m = 256; the narrow image of m = 64 (16-32 KB; HBM-bound shape) leaves
  // room for more rows in flight: SPR_WS_WG_PER_CU workgroups per CU there (default 2)
  static const int narrow_per_cu = [] { const char *e = getenv("SPR_WS_WG_PER_CU"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 4 ? v : 2; }();
  plan.total_wg = (cus > 0 ? cus : 256) * (MT <= 4 ? narrow_per_cu : 1);
  plan.chunk_rows = WS_ROWS;
  const int grid = seg_total_wgs(plan);
  // only the packed, 16-byte-aligned layout is built (one 64-byte piece per row and wave instruction); anything else
  // stays on the general kernel
  const bool vec = (m == 16 * MT) && ((sizeof(TX) * ldx) % 16 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  if (!vec) return SPR_E_UNSUPPORTED;
  if (nrm2)
    hipLaunchKernelGGL((project_ws_kernel<MT, RT, 1, TX, TU, true>), dim3(grid), dim3(WS_WAVES * 64), 0, st, X, ldx, (int)m,
                       center, plan, inv_scale, rowmean, W, (int)r, Ur, ldu, nrm2);
  else
    hipLaunchKernelGGL((project_ws_kernel<MT, RT, 1, TX, TU, false>), dim3(grid), dim3(WS_WAVES * 64), 0, st, X, ldx, (int)m,
                       center, plan, inv_scale, rowmean, W, (int)r, Ur, ldu, nrm2);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

}  // namespace

// Internal (not part of the C ABI): called by project.hip's entry points.  Returns SPR_E_UNSUPPORTED when the shape is
// outside the W-stationary kernel's range (the caller then uses the general kernel).
#include "project_ws.hpp"

template <typename TX, typename TU>
int spr_project_ws(const TX *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0, int64_t n_points,
                   int32_t n_features, int32_t center, const double *d_inv_scale, const double *d_rowmean,
                   const double *d_W, int32_t r, TU *d_Ur, int64_t ldu, int32_t accumulate, double *d_rownorm2,
                   hipStream_t st) {
  if (accumulate) return SPR_E_UNSUPPORTED;          // second column slice of a wide X: general kernel
  const int mt = spr_round_mt(m);
  const int need = (r + 15) / 16;
  const int rt = need <= 2 ? 2 : need <= 4 ? 4 : 0;
#define WS(MTV, RTV)                                                                                             \
  if (mt == MTV && rt == RTV)                                                                                    \
    return ws_launch<MTV, RTV, TX, TU>(d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,   \
                                        d_rowmean, d_W, r, d_Ur, ldu, accumulate, d_rownorm2, st)
#ifdef PROJ_WS_LAB
  WS(16, 4);
#else
  WS(4, 2); WS(4, 4); WS(8, 2); WS(8, 4); WS(12, 2); WS(12, 4); WS(16, 2); WS(16, 4);
#endif
#undef WS
  return SPR_E_UNSUPPORTED;
}

template int spr_project_ws<double, double>(const double *, int64_t, int32_t, int64_t, int64_t, int64_t, int32_t, int32_t,
                                            const double *, const double *, const double *, int32_t, double *, int64_t,
                                            int32_t, double *, hipStream_t);
#ifndef PROJ_WS_LAB
template int spr_project_ws<float, float>(const float *, int64_t, int32_t, int64_t, int64_t, int64_t, int32_t, int32_t,
                                          const double *, const double *, const double *, int32_t, float *, int64_t,
                                          int32_t, double *, hipStream_t);
// f32 shard, f64 basis: the default for a float32 X (the reference's U is float64 whatever the dtype of X)
template int spr_project_ws<float, double>(const float *, int64_t, int32_t, int64_t, int64_t, int64_t, int32_t, int32_t,
                                           const double *, const double *, const double *, int32_t, double *, int64_t,
                                           int32_t, double *, hipStream_t);
#endif
