// K3a': per-feature Gram blocks -> ONE scaled m x m Gram matrix, on the device.
//
// fit() needs G = sum_f G_f / X_scl_f^2 (the Gram matrix of the centred AND scaled snapshot matrix, reference
// :115, :169, :272) for its m x m eigen-problem.  The fused pass (stats_gram.hip) leaves F blocks G_f (all-reduced
// over the ranks) and per-rank feature statistics; merging them on the host meant downloading F m^2 doubles
// (4.7 MB at config 3) and three small uploads between the two passes over X.  This kernel does the Chan merge
// (rank order), the block variance (:115: trace(G_f) and M2), the per-feature scale of the chosen scale_type
// (:114-161, the ones that derive from block mean/variance -- same codes as spectrum.hip) and the scaled sum, so
// one packed download of m^2 + 5 F doubles remains and the scales stay in HBM for the projection.
#include "common.hpp"

namespace {

enum { SC_STD = 0, SC_NONE = 1, SC_PARETO = 2, SC_VAST = 3, SC_LEVEL = 4, SC_VARIANCE = 5, SC_POISSON = 6, SC_L2 = 7 };
constexpr int CB_THREADS = 256;
constexpr int CB_MAXF = 1024;

__global__ __launch_bounds__(CB_THREADS) void gram_combine_kernel(
    const double *__restrict__ gram, const double *__restrict__ fstats_all, int n_ranks, int n_features, int m,
    int scale_code, double *__restrict__ G_out, double *__restrict__ feat_out /*[F][5]: cnt, mu, var, scl, fluctuation variance*/,
    double *__restrict__ scale, double *__restrict__ inv_scale) {
  __shared__ double scl2[CB_MAXF];
  const int tid = threadIdx.x;
  // every workgroup derives the F scales itself (F m + 3 F ranks loads: nothing next to its share of F m^2).
  // traces: thread t adds the diagonal entries i = t % 64, t % 64 + 64, ... of feature t / 64 (+4, +8, ...), then a
  // 64-lane butterfly -- a fixed order, so every workgroup (and every rank) gets the same bits
  __shared__ double trs[CB_MAXF];
  for (int f = tid >> 6; f < n_features; f += CB_THREADS / 64) {
    const double *G = gram + (int64_t)f * m * m;
    double t = 0.0;
    for (int i = tid & 63; i < m; i += 64) t += G[(int64_t)i * m + i];
    t = group_sum_t<64>(t);
    if ((tid & 63) == 0) trs[f] = t;
  }
  __syncthreads();
  for (int f = tid; f < n_features; f += CB_THREADS) {
    double n = 0.0, mu = 0.0, m2 = 0.0;
    for (int w = 0; w < n_ranks; ++w) {   // Chan merge in rank order
      const double *q = fstats_all + ((int64_t)w * n_features + f) * 3;
      const double nb = q[0], mb = q[1], sb = q[2];
      if (nb > 0.0) {
        const double tot = n + nb, d = mb - mu;
        mu += d * nb / tot;
        m2 += sb + d * d * n * nb / tot;
        n = tot;
      }
    }
    // population variance of the raw block (:115).  A feature WITHOUT rows (only possible in a partial row group,
    // RowShard(partial=True): one rank's block of a larger job run alone) takes no part: scale 1, Gram block zero
    const bool present = n > 0.0;
    const double var = present ? (trs[f] + m * m2) / (n * m) : 0.0;
    const double sd = sqrt(var);
    double scl;
    switch (scale_code) {
      case SC_NONE: scl = 1.0; break;
      case SC_PARETO: scl = sqrt(sd); break;
      case SC_VAST: scl = var / mu; break;
      case SC_LEVEL: scl = mu; break;
      case SC_VARIANCE: scl = var; break;
      case SC_POISSON: scl = sqrt(mu); break;
      case SC_L2: scl = sqrt(n * m * (var + mu * mu)); break;
      default: scl = sd; break;
    }
    if (!present) scl = 1.0;
    scl2[f] = scl * scl;
    if (blockIdx.x == 0) {
      feat_out[5 * f] = n; feat_out[5 * f + 1] = mu; feat_out[5 * f + 2] = var;
      // slot 4: variance of the ROW-CENTRED values, trace(G_f) / (n m) -- var minus the spread of the row means; what the
      // host compares the row means with when it decides how the projection removes them (ROM._needs_precenter)
      feat_out[5 * f + 3] = scl; feat_out[5 * f + 4] = present ? trs[f] / (n * m) : 0.0;
      scale[f] = scl;
      inv_scale[f] = 1.0 / scl;
    }
  }
  __syncthreads();
  const int64_t mm = (int64_t)m * m;
  for (int64_t e = (int64_t)blockIdx.x * CB_THREADS + tid; e < mm; e += (int64_t)gridDim.x * CB_THREADS) {
    double acc = 0.0;
    for (int f = 0; f < n_features; ++f) acc += gram[(int64_t)f * mm + e] / scl2[f];   // G_f / scl_f^2, feature order
    G_out[e] = acc;
  }
}

}  // namespace

extern "C" int spr_gram_combine_f64(const double *d_gram, const double *d_fstats_all, int32_t n_ranks,
                                    int32_t n_features, int32_t m, int32_t scale_code, double *d_G, double *d_feat,
                                    double *d_scale, double *d_inv_scale, void *stream) {
  SPR_REQUIRE(d_gram && d_fstats_all && d_G && d_feat && d_scale && d_inv_scale, SPR_E_INVALID,
              "spr_gram_combine_f64: NULL pointer");
  SPR_REQUIRE(n_ranks > 0 && n_features > 0 && m > 0, SPR_E_INVALID, "spr_gram_combine_f64: bad shape");
  SPR_REQUIRE(n_features <= CB_MAXF, SPR_E_UNSUPPORTED, "spr_gram_combine_f64: more than %d features", CB_MAXF);
  SPR_REQUIRE(scale_code >= 0 && scale_code <= 7, SPR_E_INVALID, "spr_gram_combine_f64: scale code %d", scale_code);
  int grid = (int)(((int64_t)m * m + CB_THREADS - 1) / CB_THREADS);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(gram_combine_kernel, dim3(grid), dim3(CB_THREADS), 0, static_cast<hipStream_t>(stream), d_gram,
                     d_fstats_all, (int)n_ranks, (int)n_features, (int)m, (int)scale_code, d_G, d_feat, d_scale,
                     d_inv_scale);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}
