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
#pragma once
#include "common.hpp"

struct RowStats {  // running statistics of the row means seen by one lane group
  // Shifted sums: d = x - ref with ref = the first row mean seen, so no division sits in the
  // streaming loop; converted to the (count, mean, M2) form that Chan's merge wants at the end.
  double cnt, ref, s1, s2;
  __device__ inline void init() { cnt = 0.0; ref = 0.0; s1 = 0.0; s2 = 0.0; }
  __device__ inline void push(double x) {
    ref = (cnt == 0.0) ? x : ref;
    const double d = x - ref;
    cnt += 1.0;
    s1 += d;
    s2 += d * d;
  }
  __device__ inline double mean() const { return cnt > 0.0 ? ref + s1 / cnt : 0.0; }
  __device__ inline double m2() const {
    const double v = cnt > 0.0 ? s2 - s1 * s1 / cnt : 0.0;
    return v > 0.0 ? v : 0.0;
  }
};

template <int MT, int R, int MP, int NWAVES>
struct RowTile {
  static constexpr int MPAD = 16 * MT;
  static constexpr int NV = MPAD / 2;                       // 16-byte pieces per padded row
  static constexpr int LPR = spr_pow2_divisor_le64(NV);     // lanes per row
  static constexpr int VPL = NV / LPR;                      // pieces per lane
  static constexpr int RPW = 64 / LPR;                      // rows per wave instruction
  static constexpr int ROWS_PER_IT = NWAVES * RPW;
  static constexpr int IT = R / ROWS_PER_IT;
  static_assert(R % ROWS_PER_IT == 0, "panel rows must be a multiple of rows per pass");
  static_assert(MP % 2 == 0, "LDS row stride must keep rows 16-byte aligned");

  f64x2 pre[IT][VPL];

  // issue the loads of the panel whose first local row is crow0 (rows >= seg_hi read as 0)
  __device__ inline void load(const double *__restrict__ X, int64_t ldx, int m, bool vec_ok,
                              int64_t crow0, int64_t seg_hi, int wave, int lane) {
    const int grp = lane / LPR, lig = lane % LPR;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int64_t lrow = crow0 + it * ROWS_PER_IT + wave * RPW + grp;
      const bool rv = lrow < seg_hi;
      const double *rp = X + lrow * ldx;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int col = 2 * (lig + v * LPR);
        f64x2 t = {0.0, 0.0};
        if (vec_ok) {
          if (rv && col < m) t = *reinterpret_cast<const f64x2 *>(rp + col);
        } else {
          if (rv && col < m) t.x = rp[col];
          if (rv && col + 1 < m) t.y = rp[col + 1];
        }
        pre[it][v] = t;
      }
    }
  }

  // mean -> centre -> LDS; optionally store the row means and feed the Welford state
  template <bool WRITE_MEAN>
  __device__ inline void center_store(double *__restrict__ lds, int m, bool center, int64_t crow0, int64_t seg_hi,
                                      int wave, int lane, double *__restrict__ rowmean,
                                      RowStats *st) {
    const int grp = lane / LPR, lig = lane % LPR;
    const double inv_m = 1.0 / (double)m;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int rloc = it * ROWS_PER_IT + wave * RPW + grp;
      const int64_t lrow = crow0 + rloc;
      const bool rv = lrow < seg_hi;
      double s = 0.0;
#pragma unroll
      for (int v = 0; v < VPL; ++v) s += pre[it][v].x + pre[it][v].y;
      s = group_sum_t<LPR>(s);
      const double mean = center ? s * inv_m : 0.0;
      if (WRITE_MEAN && rv) {
        if (lig == 0) rowmean[lrow] = mean;
        st->push(mean);
      }
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int col = 2 * (lig + v * LPR);
        f64x2 c;
        c.x = (rv && col < m) ? pre[it][v].x - mean : 0.0;
        c.y = (rv && col + 1 < m) ? pre[it][v].y - mean : 0.0;
        *reinterpret_cast<f64x2 *>(lds + rloc * MP + col) = c;
      }
    }
  }
};
