// K2 / K11 as stand-alone calls: the explicit scaled matrix X0 = (X - X_cnt)/X_scl that
// ROM.scale_data returns (sparse_sensing.py:169) and ROM.unscale_data (:235).  The fitted
// path never materialises X0 (the Gram and projection kernels fold the scaling into their
// loads); these exist so the two public methods keep working.  Plain grid-stride
// element-wise kernels, 16-byte accesses when the layout allows.
#include "common.hpp"

namespace {

template <typename TX>
__global__ __launch_bounds__(256) void scale_rows_kernel(
    const TX *__restrict__ X, int64_t n_rows, int m, int64_t ldx, int64_t row0, int64_t n_points,
    int n_features, const double *__restrict__ rowmean, const double *__restrict__ inv_scale,
    double *__restrict__ X0, int64_t ldo) {
  // one wave per row at a time: the feature lookup is a wave-uniform division
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  for (int64_t row = wave_id; row < n_rows; row += n_waves) {
    int64_t f = (row0 + row) / n_points;
    if (f > n_features - 1) f = n_features - 1;
    const double mu = rowmean[row], is = inv_scale[f];
    for (int c = lane; c < m; c += 64) X0[row * ldo + c] = ((double)X[row * ldx + c] - mu) * is;
  }
}

__global__ __launch_bounds__(256) void unscale_kernel(
    const double *__restrict__ x0, int64_t n_rows, int64_t row0, int64_t n_points, int n_features,
    const double *__restrict__ rowmean, const double *__restrict__ scale, const double *__restrict__ rowscale,
    double *__restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t f = (row0 + i) / n_points;
    if (f > n_features - 1) f = n_features - 1;
    x[i] = (rowscale ? rowscale[i] : scale[f]) * x0[i] + rowmean[i];
  }
}

int grid_for(int64_t work_items, int per_block) {
  int64_t b = (work_items + per_block - 1) / per_block;
  const int cus = spr_cached_cus();
  const int64_t cap = 16LL * (cus > 0 ? cus : 256);
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

template <typename TX>
static int scale_rows_entry(const char *who, const TX *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                            int64_t n_points, int32_t n_features, const double *d_rowmean, const double *d_inv_scale,
                            double *d_X0, int64_t ldo, void *stream) {
  SPR_REQUIRE(d_X && d_rowmean && d_inv_scale && d_X0, SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(n_rows > 0 && m > 0 && ldx >= m && ldo >= m && row0 >= 0 && n_points > 0 && n_features > 0,
              SPR_E_INVALID, "%s: bad shape", who);
  hipLaunchKernelGGL(scale_rows_kernel<TX>, dim3(grid_for(n_rows, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     d_X, n_rows, (int)m, ldx, row0, n_points, (int)n_features, d_rowmean, d_inv_scale, d_X0, ldo);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

extern "C" int spr_scale_rows_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                  int64_t n_points, int32_t n_features, const double *d_rowmean,
                                  const double *d_inv_scale, double *d_X0, int64_t ldo, void *stream) {
  return scale_rows_entry("spr_scale_rows_f64", d_X, n_rows, m, ldx, row0, n_points, n_features, d_rowmean, d_inv_scale,
                          d_X0, ldo, stream);
}

extern "C" int spr_scale_rows_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                  int64_t n_points, int32_t n_features, const double *d_rowmean,
                                  const double *d_inv_scale, double *d_X0, int64_t ldo, void *stream) {
  return scale_rows_entry("spr_scale_rows_x32", d_X, n_rows, m, ldx, row0, n_points, n_features, d_rowmean, d_inv_scale,
                          d_X0, ldo, stream);
}

extern "C" int spr_unscale_f64(const double *d_x0, int64_t n_rows, int64_t row0, int64_t n_points,
                               int32_t n_features, const double *d_rowmean, const double *d_scale,
                               const double *d_rowscale, double *d_x, void *stream) {
  SPR_REQUIRE(d_x0 && d_rowmean && d_scale && d_x, SPR_E_INVALID, "spr_unscale_f64: NULL pointer");
  SPR_REQUIRE(n_rows > 0 && row0 >= 0 && n_points > 0 && n_features > 0, SPR_E_INVALID,
              "spr_unscale_f64: bad shape");
  hipLaunchKernelGGL(unscale_kernel, dim3(grid_for(n_rows, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     d_x0, n_rows, row0, n_points, (int)n_features, d_rowmean, d_scale, d_rowscale, d_x);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

// ---- per-feature minimum / maximum of the raw block (scale_type 'range' :128, 'max' :135) -----
// A separate streaming pass, only run for those two scalings, so the fused Gram pass stays lean.
namespace {

template <typename TX>
__global__ __launch_bounds__(256) void minmax_kernel(const TX *__restrict__ X, int64_t ldx, int m, SegPlan plan,
                                                     double *__restrict__ part) {
  __shared__ double smin[4], smax[4];
  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double mn = INFINITY, mx = -INFINITY;
  for (int64_t row = lo + (int64_t)wl * 4 + wave; row < hi; row += (int64_t)wpf * 4) {
    const TX *rp = X + row * ldx;
    for (int c = lane; c < m; c += 64) {
      const double v = (double)rp[c];
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if (lane == 0) { smin[wave] = mn; smax[wave] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) { mn = smin[w] < mn ? smin[w] : mn; mx = smax[w] > mx ? smax[w] : mx; }
    part[2 * (int64_t)blockIdx.x] = mn;
    part[2 * (int64_t)blockIdx.x + 1] = mx;
  }
}

__global__ void minmax_finalize_kernel(const double *__restrict__ part, SegPlan plan, double *__restrict__ out) {
  const int f = blockIdx.x;
  int base = 0, wpf = 0, acc = 0;
  for (int ff = seg_first_feature(plan); ff <= seg_last_feature(plan); ++ff) {
    int64_t lo, hi;
    seg_range(plan, ff, lo, hi);
    const int w = seg_wgs(plan, hi - lo);
    if (ff == f) { base = acc; wpf = w; }
    acc += w;
  }
  double mn = INFINITY, mx = -INFINITY;
  for (int p = threadIdx.x; p < wpf; p += 64) {
    const double a = part[2 * (int64_t)(base + p)], b = part[2 * (int64_t)(base + p) + 1];
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if (threadIdx.x == 0) { out[2 * f] = mn; out[2 * f + 1] = mx; }   // +inf / -inf for features with no local rows
}

}  // namespace

extern "C" size_t spr_feature_minmax_workspace(int32_t n_features) {
  const int cus = spr_cached_cus();
  return sizeof(double) * 2 * ((size_t)8 * (cus > 0 ? cus : 256) + (size_t)n_features);
}

template <typename TX>
static int minmax_entry(const char *who, const TX *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                        int64_t n_points, int32_t n_features, double *d_minmax, void *d_workspace,
                        size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(d_X && d_minmax && d_workspace, SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(n_rows > 0 && m > 0 && ldx >= m && row0 >= 0 && n_points > 0 && n_features > 0 &&
                  row0 + n_rows <= n_points * (int64_t)n_features,
              SPR_E_INVALID, "%s: bad shape", who);
  SPR_REQUIRE(workspace_bytes >= spr_feature_minmax_workspace(n_features), SPR_E_WORKSPACE, "%s: workspace too small", who);
  const int cus = spr_cached_cus();
  SegPlan plan;
  plan.row0 = row0; plan.n_rows = n_rows; plan.n_points = n_points; plan.n_features = n_features;
  plan.total_wg = 8 * (cus > 0 ? cus : 256); plan.chunk_rows = 4;
  const int grid = seg_total_wgs(plan);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(minmax_kernel<TX>, dim3(grid), dim3(256), 0, st, d_X, ldx, (int)m, plan,
                     static_cast<double *>(d_workspace));
  SPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(minmax_finalize_kernel, dim3(n_features), dim3(64), 0, st,
                     static_cast<const double *>(d_workspace), plan, d_minmax);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

extern "C" int spr_feature_minmax_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                      int64_t n_points, int32_t n_features, double *d_minmax, void *d_workspace,
                                      size_t workspace_bytes, void *stream) {
  return minmax_entry("spr_feature_minmax_f64", d_X, n_rows, m, ldx, row0, n_points, n_features, d_minmax, d_workspace,
                      workspace_bytes, stream);
}

extern "C" int spr_feature_minmax_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                      int64_t n_points, int32_t n_features, double *d_minmax, void *d_workspace,
                                      size_t workspace_bytes, void *stream) {
  return minmax_entry("spr_feature_minmax_x32", d_X, n_rows, m, ldx, row0, n_points, n_features, d_minmax, d_workspace,
                      workspace_bytes, stream);
}

// ---- axis_cnt=None support (scalar centring per feature, sparse_sensing.py:112 with axis=None) ----
// The Gram pass centres every row by its own mean.  For X0 = (X - mu_f)/scl the Gram matrix is
//   sum_i (c_i + d_i 1)(c_i + d_i 1)^T = G_c + v 1^T + 1 v^T + (sum d_i^2) 1 1^T,   d_i = mean_i - mu_f,
// with v = sum_i d_i c_i = w_f - mu_f z_f, w_f = sum_i mean_i c_i, z_f = sum_i c_i (c_i = x_i - mean_i).
// spr_colsums_f64 produces z_f and w_f (one extra read of X, only for this option);
// spr_fill_feature_f64 expands a per-feature scalar to a per-row vector (the new X_cnt).
namespace {

template <typename TX>
__global__ __launch_bounds__(256) void colsums_kernel(const TX *__restrict__ X, int64_t ldx, int m, int c0, SegPlan plan,
                                                      const double *__restrict__ rowmean, double *__restrict__ part) {
  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;
  double z[2] = {0.0, 0.0}, w[2] = {0.0, 0.0};
  for (int64_t row = lo + wl; row < hi; row += wpf) {
    const double mu = rowmean[row];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = c0 + threadIdx.x + 256 * q;                 // this launch covers columns [c0, c0 + 512)
      if (c < m) {
        const double d = (double)X[row * ldx + c] - mu;
        z[q] += d;
        w[q] += mu * d;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = c0 + threadIdx.x + 256 * q;
    if (c < m) {
      part[((int64_t)blockIdx.x * 2 + 0) * m + c] = z[q];
      part[((int64_t)blockIdx.x * 2 + 1) * m + c] = w[q];
    }
  }
}

__global__ void colsums_finalize_kernel(const double *__restrict__ part, int m, SegPlan plan, double *__restrict__ out) {
  const int f = blockIdx.x;
  int base = 0, wpf = 0, acc = 0;
  for (int ff = seg_first_feature(plan); ff <= seg_last_feature(plan); ++ff) {
    int64_t lo, hi;
    seg_range(plan, ff, lo, hi);
    const int wgs = seg_wgs(plan, hi - lo);
    if (ff == f) { base = acc; wpf = wgs; }
    acc += wgs;
  }
  for (int e = threadIdx.x; e < 2 * m; e += blockDim.x) {
    const int k = e / m, c = e - k * m;
    double s = 0.0;
    for (int p = 0; p < wpf; ++p) s += part[((int64_t)(base + p) * 2 + k) * m + c];
    out[((int64_t)f * 2 + k) * m + c] = s;
  }
}

__global__ void fill_feature_kernel(double *__restrict__ out, int64_t n_rows, int64_t row0, int64_t n_points,
                                    int n_features, const double *__restrict__ values) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t f = (row0 + i) / n_points;
    if (f > n_features - 1) f = n_features - 1;
    out[i] = values[f];
  }
}

}  // namespace

extern "C" size_t spr_colsums_workspace(int32_t m, int32_t n_features) {
  const int cus = spr_cached_cus();
  return sizeof(double) * 2 * (size_t)(m > 0 ? m : 1) * ((size_t)4 * (cus > 0 ? cus : 256) + (size_t)n_features);
}

template <typename TX>
static int colsums_entry(const char *who, const TX *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                         int64_t n_points, int32_t n_features, const double *d_rowmean, double *d_out,
                         void *d_workspace, size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(d_X && d_rowmean && d_out && d_workspace, SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(n_rows > 0 && m > 0 && ldx >= m && row0 >= 0 && n_points > 0 && n_features > 0 &&
                  row0 + n_rows <= n_points * (int64_t)n_features,
              SPR_E_INVALID, "%s: bad shape", who);
  SPR_REQUIRE(workspace_bytes >= spr_colsums_workspace(m, n_features), SPR_E_WORKSPACE, "%s: workspace too small", who);
  const int cus = spr_cached_cus();
  SegPlan plan;
  plan.row0 = row0; plan.n_rows = n_rows; plan.n_points = n_points; plan.n_features = n_features;
  plan.total_wg = 4 * (cus > 0 ? cus : 256); plan.chunk_rows = 1;
  const int grid = seg_total_wgs(plan);
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int c0 = 0; c0 < m; c0 += 512) {                        // 512 columns per launch, any m
    hipLaunchKernelGGL(colsums_kernel<TX>, dim3(grid), dim3(256), 0, st, d_X, ldx, (int)m, c0, plan, d_rowmean,
                       static_cast<double *>(d_workspace));
    SPR_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(colsums_finalize_kernel, dim3(n_features), dim3(256), 0, st,
                     static_cast<const double *>(d_workspace), (int)m, plan, d_out);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

extern "C" int spr_colsums_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                               int64_t n_points, int32_t n_features, const double *d_rowmean, double *d_out,
                               void *d_workspace, size_t workspace_bytes, void *stream) {
  return colsums_entry("spr_colsums_f64", d_X, n_rows, m, ldx, row0, n_points, n_features, d_rowmean, d_out,
                       d_workspace, workspace_bytes, stream);
}

extern "C" int spr_colsums_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                               int64_t n_points, int32_t n_features, const double *d_rowmean, double *d_out,
                               void *d_workspace, size_t workspace_bytes, void *stream) {
  return colsums_entry("spr_colsums_x32", d_X, n_rows, m, ldx, row0, n_points, n_features, d_rowmean, d_out,
                       d_workspace, workspace_bytes, stream);
}

extern "C" int spr_fill_feature_f64(double *d_out, int64_t n_rows, int64_t row0, int64_t n_points, int32_t n_features,
                                    const double *d_values, void *stream) {
  SPR_REQUIRE(d_out && d_values, SPR_E_INVALID, "spr_fill_feature_f64: NULL pointer");
  SPR_REQUIRE(n_rows > 0 && row0 >= 0 && n_points > 0 && n_features > 0, SPR_E_INVALID,
              "spr_fill_feature_f64: bad shape");
  hipLaunchKernelGGL(fill_feature_kernel, dim3(grid_for(n_rows, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     d_out, n_rows, row0, n_points, (int)n_features, d_values);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}
