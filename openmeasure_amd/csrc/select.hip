// K2b: per-feature order statistics of the raw blocks by radix selection -- the device half of
// scale_type='median' (np.median(x) over the n_points*m values of a feature block, sparse_sensing.py:140-141).
//
// A value x maps to an order-preserving 64-bit key (sign bit flipped for x >= 0, all bits inverted for x < 0).
// One call histograms one digit of the keys (most significant digit first, <= 13 bits) for every feature and
// for two targets at once -- the lower and the upper middle element, which np.median averages for an even count.
// Only keys that agree with the target's prefix above the digit are counted.  The host (or every rank, after an
// all-reduce of the histograms) walks the cumulative counts, extends the two prefixes by one digit and calls
// again: 5 passes (13+13+13+13+12 bits), each one coalesced read of X, HBM-bound.  Histograms live in LDS
// (ds_add_u32) and are flushed with one 64-bit global atomic per non-empty bin: integer sums, so the result does
// not depend on the order of workgroups or ranks.
#include "common.hpp"

namespace {

constexpr int SEL_THREADS = 512;
constexpr int SEL_MAX_BITS = 13;

__device__ __forceinline__ uint64_t order_key(double x) {
  const uint64_t u = (uint64_t)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

template <bool TWO, typename TX>
__global__ __launch_bounds__(SEL_THREADS) void digit_hist_kernel(const TX *__restrict__ X, int64_t ldx, int m,
                                                                 SegPlan plan, const uint64_t *__restrict__ prefix,
                                                                 int shift, int bits,
                                                                 unsigned long long *__restrict__ hist) {
  extern __shared__ uint32_t lh[];   // [TWO ? 2 : 1][1 << bits]
  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;
  const int nb = 1 << bits;
  for (int b = threadIdx.x; b < (TWO ? 2 : 1) * nb; b += SEL_THREADS) lh[b] = 0u;
  __syncthreads();
  const int top = shift + bits;                                  // bits above the digit must match the prefix
  const bool all = top >= 64;
  const uint64_t pa = all ? 0 : (prefix[2 * f] >> top), pb = all ? 0 : (prefix[2 * f + 1] >> top);
  const uint32_t mask = (uint32_t)nb - 1u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int WAVES = SEL_THREADS / 64;
  for (int64_t row = lo + (int64_t)wl * WAVES + wave; row < hi; row += (int64_t)wpf * WAVES) {
    const TX *rp = X + row * ldx;
    for (int c = lane; c < m; c += 64) {
      const uint64_t k = order_key((double)rp[c]);   // f32 -> f64 is monotone: same order statistics
      const uint64_t up = all ? 0 : (k >> top);
      const uint32_t d = (uint32_t)(k >> shift) & mask;
      if (up == pa) atomicAdd(&lh[d], 1u);
      if (TWO && up == pb) atomicAdd(&lh[nb + d], 1u);
    }
  }
  __syncthreads();
  unsigned long long *out = hist + (size_t)f * 2 * nb;
  for (int b = threadIdx.x; b < nb; b += SEL_THREADS) {
    const uint32_t a = lh[b];
    if (a) atomicAdd(&out[b], (unsigned long long)a);
    const uint32_t c = TWO ? lh[nb + b] : a;                    // identical prefixes: both targets see the same counts
    if (c) atomicAdd(&out[nb + b], (unsigned long long)c);
  }
}

}  // namespace

template <typename TX>
static int digit_hist_entry(const char *who, const TX *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                            int64_t n_points, int32_t n_features, const uint64_t *d_prefix, int32_t shift, int32_t bits,
                            int32_t two_targets, uint64_t *d_hist, void *stream) {
  SPR_REQUIRE(d_X && d_prefix && d_hist, SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(n_rows > 0 && m > 0 && ldx >= m && row0 >= 0 && n_points > 0 && n_features > 0 &&
                  row0 + n_rows <= n_points * (int64_t)n_features,
              SPR_E_INVALID, "%s: bad shape", who);
  SPR_REQUIRE(bits >= 1 && bits <= SEL_MAX_BITS && shift >= 0 && shift + bits <= 64, SPR_E_INVALID,
              "%s: digit shift=%d bits=%d outside [0,64), width 1..%d", who, shift, bits, SEL_MAX_BITS);
  const int cus = spr_cached_cus();
  SegPlan plan;
  plan.row0 = row0; plan.n_rows = n_rows; plan.n_points = n_points; plan.n_features = n_features;
  plan.total_wg = 4 * (cus > 0 ? cus : 256); plan.chunk_rows = SEL_THREADS / 64;
  const int grid = seg_total_wgs(plan);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t lds = sizeof(uint32_t) * ((size_t)1 << bits) * (two_targets ? 2 : 1);
  if (two_targets)
    hipLaunchKernelGGL((digit_hist_kernel<true, TX>), dim3(grid), dim3(SEL_THREADS), lds, st, d_X, ldx, (int)m, plan,
                       d_prefix, (int)shift, (int)bits, reinterpret_cast<unsigned long long *>(d_hist));
  else
    hipLaunchKernelGGL((digit_hist_kernel<false, TX>), dim3(grid), dim3(SEL_THREADS), lds, st, d_X, ldx, (int)m, plan,
                       d_prefix, (int)shift, (int)bits, reinterpret_cast<unsigned long long *>(d_hist));
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

extern "C" int spr_feature_digit_hist_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                          int64_t n_points, int32_t n_features, const uint64_t *d_prefix,
                                          int32_t shift, int32_t bits, int32_t two_targets, uint64_t *d_hist,
                                          void *stream) {
  return digit_hist_entry("spr_feature_digit_hist_f64", d_X, n_rows, m, ldx, row0, n_points, n_features, d_prefix,
                          shift, bits, two_targets, d_hist, stream);
}

extern "C" int spr_feature_digit_hist_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                          int64_t n_points, int32_t n_features, const uint64_t *d_prefix,
                                          int32_t shift, int32_t bits, int32_t two_targets, uint64_t *d_hist,
                                          void *stream) {
  return digit_hist_entry("spr_feature_digit_hist_x32", d_X, n_rows, m, ldx, row0, n_points, n_features, d_prefix,
                          shift, bits, two_targets, d_hist, stream);
}
