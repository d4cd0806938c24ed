// Row-panel staging shared by the Gram and the projection kernels.
//
// A workgroup of NWAVES waves consumes the snapshot matrix in panels of R rows.  Each
// row (m doubles, contiguous in HBM) is read by a group of LPR consecutive lanes as
// 16-byte pieces, so a wave instruction moves 64*16 B of contiguous memory when the
// rows are packed (ldx == m).  The row mean is formed while the row is still in
// registers (butterfly over the LPR lanes), the centred row is written to an LDS image
// of R x MP doubles (MP = padded row stride chosen by the consumer for its bank
// pattern), and the next panel's loads are issued before the consumer starts on the
// current one.  Columns >= m and rows past the segment end are written as zeros, so
// the consumers never need a bounds test.
//
// Everything here is branch-free on purpose: loads always go to a clamped, valid address
// and validity is applied with selects.  Predicated loads split the basic block, and
// hipcc's s_waitcnt insertion then falls back to vmcnt(0) right behind the loads it has
// just issued, which serialises the HBM latency into the MFMA loop.
#pragma once
#include "common.hpp"

// Storage type of the streamed matrix: f64, or f32 widened to f64 when a piece is consumed (all arithmetic stays
// f64; the raw piece is what waits in registers, so an f32 shard also halves the prefetch registers).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename TX> struct PieceOf;
template <> struct PieceOf<double> { using type = f64x2; };
template <> struct PieceOf<float> { using type = f32x2; };
__device__ inline f64x2 widen(f64x2 p) { return p; }
__device__ inline f64x2 widen(f32x2 p) { return (f64x2){(double)p.x, (double)p.y}; }

// Wave-uniform "no selects needed" fast paths.  Measured: the extra branch costs more than the selects it
// saves in the projection staging (basic-block split -> conservative waits), so it is off there.
#ifndef ROWTILE_RAW_FAST
#define ROWTILE_RAW_FAST 0
#endif
#ifndef ROWTILE_CENTER_FAST
#define ROWTILE_CENTER_FAST 1
#endif

struct RowStats {  // running statistics of the row means seen by one lane group
  // Shifted sums: d = x - ref with ref = the first row mean seen, so no division sits in the
  // streaming loop; converted to the (count, mean, M2) form that Chan's merge wants at the end.
  double cnt, ref, s1, s2;
  __device__ inline void init() { cnt = 0.0; ref = 0.0; s1 = 0.0; s2 = 0.0; }
  __device__ inline void push(double x, bool valid) {
    ref = (cnt == 0.0) ? x : ref;
    const double d = valid ? x - ref : 0.0;
    cnt += valid ? 1.0 : 0.0;
    s1 += d;
    s2 += d * d;
  }
  __device__ inline double mean() const { return cnt > 0.0 ? ref + s1 / cnt : 0.0; }
  __device__ inline double m2() const {
    const double v = cnt > 0.0 ? s2 - s1 * s1 / cnt : 0.0;
    return v > 0.0 ? v : 0.0;
  }
};

// LPRMAX caps the lanes that share a row.  Fewer lanes per row means more of the row sum is
// formed by plain per-lane adds and less by cross-lane steps: at 16 lanes the butterfly is
// four DPP steps (no ds_bpermute) and one wave instruction handles four rows at once.  This
// matters because VALU work does not hide behind v_mfma_f64_16x16x4_f64 on gfx950 (measured:
// kernel time = MFMA time + VALU time), so the centring pass is kept as short as possible.
template <int MT, int R, int MP, int NWAVES, int LPRMAX = 16, typename TX = double>
struct RowTile {
  using Piece = typename PieceOf<TX>::type;
  static constexpr int MPAD = 16 * MT;
  static constexpr int NV = MPAD / 2;                       // 16-byte pieces per padded row
  static constexpr int LPR0 = spr_pow2_divisor_le64(NV);
  static constexpr int LPR = LPR0 < LPRMAX ? LPR0 : LPRMAX; // lanes per row
  static constexpr int VPL = NV / LPR;                      // pieces per lane
  static constexpr int RPW = 64 / LPR;                      // rows per wave instruction
  static constexpr int ROWS_PER_IT = NWAVES * RPW;
  static constexpr int IT = R / ROWS_PER_IT;
  static_assert(R % ROWS_PER_IT == 0, "panel rows must be a multiple of rows per pass");
  static constexpr bool WIDE_STORE = (MP % 2 == 0);   // odd stride: rows are only 8-byte aligned in LDS

  Piece pre[IT][VPL];
  double pmean[IT];   // centre mode 2: the caller-supplied row mean of each staged row (loaded with the row)

  // Pass `it` (a constant after unrolling) of the panel whose first local row is crow0.
  // VEC 0: any layout (8-byte loads); 1: rows 16-byte aligned, m even; 2: additionally m == 16*MT
  // (no column clamp, constant offsets).  Rows >= seg_hi re-read the last valid row
  // and columns >= m re-read column 0; center_store_pass discards both.
  // split / gap: panel columns >= split come from `gap` elements further along the row -- a panel made of two column
  // slices of a wider matrix that are not adjacent (the off-diagonal Gram blocks of m > 512, gram_wide.hip).
  template <int VEC>
  __device__ inline void load_pass(int it, const TX *__restrict__ X, int64_t ldx, int m, int64_t crow0,
                                   int64_t seg_hi, int wave, int lane, const double *__restrict__ mean_in = nullptr,
                                   int split = 1 << 30, int64_t gap = 0) {
    const int grp = lane / LPR, lig = lane % LPR;
    int64_t lrow = crow0 + it * ROWS_PER_IT + wave * RPW + grp;
    lrow = lrow < seg_hi ? lrow : seg_hi - 1;
    const TX *rp = X + lrow * ldx;
    if (mean_in) pmean[it] = mean_in[lrow];   // callers pass either nullptr at compile time or a pointer that is always valid
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int col = 2 * (lig + v * LPR);
      const int64_t g0 = col >= split ? gap : 0, g1 = col + 1 >= split ? gap : 0;
      Piece t;
      if (VEC == 2) {
        t = *reinterpret_cast<const Piece *>(rp + col + g0);          // m == MPAD: constant offsets
      } else if (VEC == 1) {
        t = *reinterpret_cast<const Piece *>(rp + (col < m ? col + g0 : 0));
      } else {
        t.x = rp[col < m ? col + g0 : 0];
        t.y = rp[col + 1 < m ? col + 1 + g1 : 0];
      }
      pre[it][v] = t;
    }
  }

  // one 16-byte piece (it, v) of a pass: lets a consumer spread the HBM requests of the next
  // panel over its whole MFMA phase instead of issuing them in one burst
  template <int VEC>
  __device__ inline void load_piece(int it, int v, const TX *__restrict__ X, int64_t ldx, int m, int64_t crow0,
                                    int64_t seg_hi, int wave, int lane) {
    const int grp = lane / LPR, lig = lane % LPR;
    int64_t lrow = crow0 + it * ROWS_PER_IT + wave * RPW + grp;
    lrow = lrow < seg_hi ? lrow : seg_hi - 1;
    const TX *rp = X + lrow * ldx;
    const int col = 2 * (lig + v * LPR);
    Piece t;
    if (VEC == 2) {
      t = *reinterpret_cast<const Piece *>(rp + col);
    } else if (VEC == 1) {
      t = *reinterpret_cast<const Piece *>(rp + (col < m ? col : 0));
    } else {
      t.x = rp[col < m ? col : 0];
      t.y = rp[col + 1 < m ? col + 1 : 0];
    }
    pre[it][v] = t;
  }

  __device__ inline void raw_store_piece(int it, int v, double *__restrict__ lds, int m, int64_t crow0,
                                         int64_t seg_hi, int wave, int lane) {
    const int grp = lane / LPR, lig = lane % LPR;
    const int rloc = it * ROWS_PER_IT + wave * RPW + grp;
    const bool rv = crow0 + rloc < seg_hi;
    const int col = 2 * (lig + v * LPR);
    f64x2 c = widen(pre[it][v]);
    // wave-uniform fast path: whole pass inside the segment and no padded columns -> no selects
    const bool fast = ROWTILE_RAW_FAST && (crow0 + (it + 1) * ROWS_PER_IT <= seg_hi) && (m == MPAD);
    if (!fast) {
      c.x = (rv && col < m) ? c.x : 0.0;
      c.y = (rv && col + 1 < m) ? c.y : 0.0;
    }
    if constexpr (WIDE_STORE) {
      *reinterpret_cast<f64x2 *>(lds + rloc * MP + col) = c;
    } else {
      lds[rloc * MP + col] = c.x;
      lds[rloc * MP + col + 1] = c.y;
    }
  }

  template <int VEC>
  __device__ inline void load(const TX *__restrict__ X, int64_t ldx, int m, int64_t crow0, int64_t seg_hi,
                              int wave, int lane, const double *__restrict__ mean_in = nullptr, int split = 1 << 30,
                              int64_t gap = 0) {
#pragma unroll
    for (int it = 0; it < IT; ++it) load_pass<VEC>(it, X, ldx, m, crow0, seg_hi, wave, lane, mean_in, split, gap);
  }

  // one pass of mean -> centre -> LDS; optionally stores the row means and feeds the running
  // statistics.  Split per pass so that a caller can slot the passes between the MFMA steps of
  // the previous panel.
  // rowsum (centre mode 2 only, may be NULL): receives the RAW sum of the row's m staged values -- the caller shifted the
  // rows by constants that are not their means and needs the sums to form the means afterwards (gram_wide.hip).
  template <bool WRITE_MEAN>
  __device__ inline void center_store_pass(int it, double *__restrict__ lds, int m, int center, int64_t crow0,
                                           int64_t seg_hi, int wave, int lane, double *__restrict__ rowmean,
                                           RowStats *st, double *__restrict__ rowsum = nullptr) {
    const int grp = lane / LPR, lig = lane % LPR;
    const double inv_m = 1.0 / (double)m;
    const int rloc = it * ROWS_PER_IT + wave * RPW + grp;
    const int64_t lrow = crow0 + rloc;
    const bool rv = lrow < seg_hi;
    // wave-uniform fast path: whole pass inside the segment and no padded columns -> no selects
    const bool fast = ROWTILE_CENTER_FAST && (crow0 + (it + 1) * ROWS_PER_IT <= seg_hi) && (m == MPAD);
    if (fast) {
      double mean;
      if constexpr (WRITE_MEAN) {
        double s = 0.0;
#pragma unroll
        for (int v = 0; v < VPL; ++v) { const f64x2 w = widen(pre[it][v]); s += w.x + w.y; }
        s = group_sum_t<LPR>(s);
        mean = center == 2 ? pmean[it] : (center ? s * inv_m : 0.0);
        if (rowsum && center == 2 && lig == 0) rowsum[lrow] = s;
      } else {
        mean = center ? pmean[it] : 0.0;                   // callers that do not write means run modes 0 and 2 only: no row sum
      }
      if (WRITE_MEAN) {
        if (lig == 0 && center != 2) rowmean[lrow] = mean;
        st->push(mean, center != 2);
      }
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int col = 2 * (lig + v * LPR);
        const f64x2 w = widen(pre[it][v]);
        f64x2 c = {w.x - mean, w.y - mean};
        if constexpr (WIDE_STORE) {
          *reinterpret_cast<f64x2 *>(lds + rloc * MP + col) = c;
        } else {
          lds[rloc * MP + col] = c.x;
          lds[rloc * MP + col + 1] = c.y;
        }
      }
      return;
    }
    double s = 0.0;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int col = 2 * (lig + v * LPR);
      const f64x2 w = widen(pre[it][v]);
      s += (col < m ? w.x : 0.0) + (col + 1 < m ? w.y : 0.0);
    }
    s = group_sum_t<LPR>(s);
    const double mean = center == 2 ? pmean[it] : (center ? s * inv_m : 0.0);
    if (WRITE_MEAN) {
      if (rv && lig == 0 && center != 2) rowmean[lrow] = mean;
      if (rowsum && center == 2 && rv && lig == 0) rowsum[lrow] = s;
      st->push(mean, rv && center != 2);
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int col = 2 * (lig + v * LPR);
      const f64x2 w = widen(pre[it][v]);
      f64x2 c;
      c.x = (rv && col < m) ? w.x - mean : 0.0;
      c.y = (rv && col + 1 < m) ? w.y - mean : 0.0;
      if constexpr (WIDE_STORE) {
        *reinterpret_cast<f64x2 *>(lds + rloc * MP + col) = c;
      } else {
        lds[rloc * MP + col] = c.x;
        lds[rloc * MP + col + 1] = c.y;
      }
    }
  }

  // ---- own-means fast lane of the m = 16 MT Gram kernels (centre mode 1, packed 16-byte-aligned rows, m == MPAD) -----
  // Every f64 VALU instruction of the staging comes out of the f64 MFMA pipe's time (tools/coexec_probe.hip), so this
  // pair does the same work as load_pass / center_store_pass with fewer of them: the row pointer is a per-lane offset
  // added to a wave-uniform base (no 64-bit multiply per pass), the centre mode is a compile-time fact (no selects, no
  // mean load), the per-feature statistics are taken afterwards from the n row means (no running sums in the loop).
  // lane_off = (wave * RPW + grp) * ldx elements, computed once; ubase = X + crow0 * ldx (wave-uniform);
  // rows_left = seg_hi - crow0 (wave-uniform, may be <= 0 for the past-the-end panel).
  // EXT: the row means come from mean_in (centre mode 2: a column slice of a wider matrix, whose means are those of the
  // full rows) -- loaded with the row, no row sum at all; else they are formed here and written (centre mode 1).
  template <bool EXT>
  __device__ inline void load_pass_own(int it, const TX *__restrict__ ubase, int64_t ldx, int64_t lane_off,
                                       int64_t rows_left, int wave, int lane, const double *__restrict__ mean_in,
                                       int64_t crow0, int split = 1 << 30, int64_t gap = 0) {
    const int grp = lane / LPR, lig = lane % LPR;
    const int64_t lrow = (int64_t)it * ROWS_PER_IT + wave * RPW + grp;          // row inside the panel
    // rows past the segment end re-read the segment's last row (never stored): one compare + select on the offset
    const int64_t last = rows_left > 0 ? rows_left - 1 : 0;
    const bool in = lrow < rows_left;
    const int64_t off = in ? lane_off + (int64_t)it * ROWS_PER_IT * ldx : last * ldx;
    const TX *rp = ubase + off;
    if (EXT) pmean[it] = mean_in[crow0 + (in ? lrow : last)];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int col = 2 * (lig + v * LPR);
      pre[it][v] = *reinterpret_cast<const Piece *>(rp + col + (col >= split ? gap : 0));   // gap: see load_pass
    }
  }

  // SUMS (with EXT): the external constants are not the rows' means -- the raw row sums go to rowsum[] as well
  template <bool FULL, bool EXT, bool SUMS = false>   // FULL: the whole pass lies inside the segment (wave-uniform): no selects
  __device__ inline void center_store_own(int it, double *__restrict__ lds, int64_t crow0, int64_t rows_left, int wave,
                                          int lane, double *__restrict__ rowmean, double *__restrict__ rowsum = nullptr) {
    const int grp = lane / LPR, lig = lane % LPR;
    constexpr double inv_m = 1.0 / (double)MPAD;
    const int rloc = it * ROWS_PER_IT + wave * RPW + grp;
    const bool rv = FULL || rloc < rows_left;
    f64x2 w[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) w[v] = widen(pre[it][v]);
    double mean;
    if (EXT) {
      mean = pmean[it];
      if (SUMS) {
        double s = w[0].x + w[0].y;
#pragma unroll
        for (int v = 1; v < VPL; ++v) s += w[v].x + w[v].y;
        s = group_sum_t<LPR>(s);
        if (rv && lig == 0) rowsum[crow0 + rloc] = s;
      }
    } else {
      double s = w[0].x + w[0].y;
#pragma unroll
      for (int v = 1; v < VPL; ++v) s += w[v].x + w[v].y;
      s = group_sum_t<LPR>(s);
      mean = s * inv_m;
      if (rv && lig == 0) rowmean[crow0 + rloc] = mean;
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int col = 2 * (lig + v * LPR);
      f64x2 c = {w[v].x - mean, w[v].y - mean};
      if (!FULL) {                               // rows past the segment end (re-reads of its last row) contribute zeros
        c.x = rv ? c.x : 0.0;
        c.y = rv ? c.y : 0.0;
      }
      *reinterpret_cast<f64x2 *>(lds + rloc * MP + col) = c;
    }
  }

  // one pass of registers -> LDS without centring (the consumer removes the row mean
  // algebraically); invalid rows / padded columns are written as zeros
  __device__ inline void raw_store_pass(int it, double *__restrict__ lds, int m, int64_t crow0, int64_t seg_hi,
                                        int wave, int lane) {
    const int grp = lane / LPR, lig = lane % LPR;
    const int rloc = it * ROWS_PER_IT + wave * RPW + grp;
    const bool rv = crow0 + rloc < seg_hi;
    const bool fast = (crow0 + (it + 1) * ROWS_PER_IT <= seg_hi) && (m == MPAD);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int col = 2 * (lig + v * LPR);
      f64x2 c = widen(pre[it][v]);
      if (!fast) {
        c.x = (rv && col < m) ? c.x : 0.0;
        c.y = (rv && col + 1 < m) ? c.y : 0.0;
      }
      if constexpr (WIDE_STORE) {
        *reinterpret_cast<f64x2 *>(lds + rloc * MP + col) = c;
      } else {
        lds[rloc * MP + col] = c.x;
        lds[rloc * MP + col + 1] = c.y;
      }
    }
  }

  __device__ inline void raw_store(double *__restrict__ lds, int m, int64_t crow0, int64_t seg_hi, int wave,
                                   int lane) {
#pragma unroll
    for (int it = 0; it < IT; ++it) raw_store_pass(it, lds, m, crow0, seg_hi, wave, lane);
  }

  template <bool WRITE_MEAN>
  __device__ inline void center_store(double *__restrict__ lds, int m, int center, int64_t crow0, int64_t seg_hi,
                                      int wave, int lane, double *__restrict__ rowmean, RowStats *st,
                                      double *__restrict__ rowsum = nullptr) {
#pragma unroll
    for (int it = 0; it < IT; ++it)
      center_store_pass<WRITE_MEAN>(it, lds, m, center, crow0, seg_hi, wave, lane, rowmean, st, rowsum);
  }
};
