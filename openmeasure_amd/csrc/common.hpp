// Shared helpers for the gfx950 SPR kernels (libspr_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "spr_hip.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// ---- host-side error plumbing (thread-local text behind spr_last_error) -------------
void spr_set_error(const char *fmt, ...);

#define SPR_REQUIRE(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      spr_set_error(__VA_ARGS__);         \
      return (code);                      \
    }                                     \
  } while (0)

#define SPR_HIP_TRY(expr)                                                          \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess) {                                                       \
      spr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),       \
                    __FILE__, __LINE__);                                           \
      return SPR_E_HIP;                                                            \
    }                                                                              \
  } while (0)

#define SPR_LAUNCH_CHECK()                                                         \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      spr_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__),   \
                    __FILE__, __LINE__);                                           \
      return SPR_E_HIP;                                                            \
    }                                                                              \
  } while (0)

int spr_cached_cus();  // compute units of the current device (0 on failure)

// ---- feature segments of a row shard ------------------------------------------------
// A rank holds global rows [row0, row0+n_rows) of the feature-major matrix; feature f
// owns global rows [f*n_points, (f+1)*n_points).  Persistent workgroups are dealt to the
// features in proportion to their local row counts; host and device evaluate the same
// integer formulas so no table has to be shipped.
struct SegPlan {
  int64_t row0, n_rows, n_points;
  int32_t n_features;
  int32_t total_wg;    // workgroups to spread over the shard
  int32_t chunk_rows;  // rows one workgroup consumes per step
};

__host__ __device__ inline void seg_range(const SegPlan &p, int f, int64_t &lo, int64_t &hi) {
  int64_t glo = (int64_t)f * p.n_points, ghi = glo + p.n_points;
  if (glo < p.row0) glo = p.row0;
  if (ghi > p.row0 + p.n_rows) ghi = p.row0 + p.n_rows;
  if (ghi < glo) ghi = glo;
  lo = glo - p.row0;
  hi = ghi - p.row0;
}

__host__ __device__ inline int seg_wgs(const SegPlan &p, int64_t rows) {
  if (rows <= 0) return 0;
  int64_t w = rows * (int64_t)p.total_wg / p.n_rows;
  int64_t chunks = (rows + p.chunk_rows - 1) / p.chunk_rows;
  if (w > chunks) w = chunks;
  if (w < 1) w = 1;
  return (int)w;
}

__host__ __device__ inline int seg_first_feature(const SegPlan &p) { return (int)(p.row0 / p.n_points); }
__host__ __device__ inline int seg_last_feature(const SegPlan &p) {  // inclusive
  int64_t f = (p.row0 + p.n_rows - 1) / p.n_points;
  if (f > p.n_features - 1) f = p.n_features - 1;
  return (int)f;
}

inline int seg_total_wgs(const SegPlan &p) {
  int tot = 0;
  for (int f = seg_first_feature(p); f <= seg_last_feature(p); ++f) {
    int64_t lo, hi;
    seg_range(p, f, lo, hi);
    tot += seg_wgs(p, hi - lo);
  }
  return tot;
}

// blockIdx -> (feature, workgroup index inside the feature, workgroups of the feature,
// first block of the feature).  Returns false for surplus blocks.
__device__ inline bool seg_locate(const SegPlan &p, int b, int &f, int &wl, int &wpf, int &base,
                                  int64_t &lo, int64_t &hi) {
  int acc = 0;
  int f1 = seg_last_feature(p);
  for (int ff = seg_first_feature(p); ff <= f1; ++ff) {
    seg_range(p, ff, lo, hi);
    int w = seg_wgs(p, hi - lo);
    if (b < acc + w) {
      f = ff; wl = b - acc; wpf = w; base = acc;
      return true;
    }
    acc += w;
  }
  return false;
}

// ---- small device utilities ------------------------------------------------------------
// Butterfly sum over aligned groups of `width` lanes; every lane of a group ends with the
// same bits (each step adds the same two partial sums in both partners).  The four steps
// inside a 16-lane row are DPP moves (quad_perm xor 1, xor 2, row_half_mirror, row_mirror):
// VALU speed, no LDS crossbar; only the 16<->32 and 32<->64 exchanges use ds_bpermute.
template <int CTRL>
__device__ inline double dpp_f64(double v) {
  union { double d; int i[2]; } a, b;
  a.d = v;
  b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xF, 0xF, true);
  b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xF, 0xF, true);
  return b.d;
}

template <int WIDTH>
__device__ inline double group_sum_t(double v) {
  if (WIDTH >= 2) v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  if (WIDTH >= 4) v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  if (WIDTH >= 8) v += dpp_f64<0x141>(v);   // row_half_mirror
  if (WIDTH >= 16) v += dpp_f64<0x140>(v);  // row_mirror
  if (WIDTH >= 32) v += __shfl_xor(v, 16, 64);
  if (WIDTH >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ inline double group_sum(double v, int width) {  // run-time width (cold paths)
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int spr_pow2_divisor_le64(int n) {  // largest power of two <= 64 dividing n
  int p = 1;
  while (p < 64 && n % (2 * p) == 0) p *= 2;
  return p;
}

// padded column count (multiple of 16) the MFMA kernels are instantiated for
inline int spr_round_mt(int m) {
  static const int sup[] = {1, 2, 3, 4, 6, 8, 12, 16};
  int need = (m + 15) / 16;
  for (int s : sup)
    if (s >= need) return s;
  return -1;
}
