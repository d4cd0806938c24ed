// K4, streamed-W form: basis projection  Ur = ((X - rowmean) . W) * (1/X_scl)  for ANY snapshot count m, r <= 128.
//
// project_ws.hip keeps W (m x r doubles) resident in LDS, which ends at 128 KB (m = 256, r = 64); project.hip keeps W in
// registers, which ends at m = 256 (a wider X went through it as two column slices whose f64 partial sums were written,
// read back and added -- 1.8x the algorithmic traffic and 0.57 of the MFMA peak at BASELINE config 5, m = 512, r = 128).
// Here the whole contraction over m happens in ONE launch with the accumulators alive across it:
//
//   * the output tile of a workgroup is 256 rows x 16 RT columns: wave w owns NB = 2 row blocks of 16 rows and ALL
//     RT column tiles, 2 x RT accumulators (128 VGPRs at RT = 8) that live for the whole k loop -- Ur is written once,
//     in its storage type, rounded once;
//   * W is streamed through LDS in chunks of KC k-rows (64 KB at KC = 64, RT = 8), double-buffered: while the MFMAs of
//     chunk g run from one buffer, every thread copies 16-byte pieces of chunk g+1 from a pre-permuted global image
//     (L2-resident: 512 KB at m = 512, r = 128) into the other -- one barrier per chunk.  The image has the byte order of
//     the LDS layout (per k-row the two column tiles of a pair interleaved, so a lane fetches both B values with one
//     conflict-free ds_read_b128, as in project_ws.hip), the copy is linear;
//   * X goes HBM -> registers directly in the MFMA A layout (project_ws.hip): lane (i = l & 15, kk = l >> 4) owns row i of
//     its block and loads the four consecutive elements [16 j + 4 kk, +4) of it; the contraction index is permuted to
//     match.  The registers of a piece are re-loaded with the same piece of the NEXT chunk right after its last use, so
//     a chunk's worth of X per wave is always in flight behind the MFMAs;
//   * the B fragments of a k-step are shared by the wave's two row blocks: 4 ds_read_b128 per 16 MFMAs.
//
// m only enters as the number of chunks, a run-time loop: no upper bound on the snapshot count.  r > 128 is handled by the
// caller in column groups of <= 128.  Centring: mode 1 = epilogue, (x - mu 1) W = x W - mu (1^T W) (no VALU work in the
// loop; accurate to eps |mu|/|x - mu| relative, like the reference's own X - mean); mode 2 = the row mean is subtracted
// from the A operand in registers before the MFMA (one v_add_f64 per RT MFMAs), for data whose mean dwarfs its
// fluctuation.
#include <type_traits>

#include "common.hpp"

namespace {

constexpr int PS_WAVES = 8;
constexpr int PS_THREADS = PS_WAVES * 64;
// 16-row blocks per wave: NB = 2 with up to 8 column tiles (r <= 128: the B fragments of a k-step serve both blocks), NB = 1 with
// 16 column tiles (r <= 256, round 4: ALL m columns of the refinement pass of fit() in one launch -- X read once per pass);
// 2 x 8 or 1 x 16 accumulator tiles, 128 VGPRs either way
constexpr int ps_rows(int nb) { return 16 * PS_WAVES * nb; }   // rows a workgroup finishes per pass over W

template <typename TX> struct PsK;                       // k-rows of W per chunk: a chunk of A pieces must fit the registers
template <> struct PsK<double> { static constexpr int KC = 32; };
template <> struct PsK<float> { static constexpr int KC = 32; };   // 64 spills (wst + 32 A registers + 128 accumulators > 256)

template <typename TX> struct PsPiece;                   // four consecutive elements of a row, as loaded
template <> struct PsPiece<double> { f64x2 a, b; };
template <> struct PsPiece<float> { float x, y, z, w; };

// VEC 1: 16-byte-aligned pieces, m a multiple of 4 (a piece is inside the row or wholly past it); VEC 0: any layout.
// Columns >= m re-read column 0: their rows of the W image are zero.
template <int VEC, bool FULLK, typename TX>
__device__ inline PsPiece<TX> ps_load(const TX *__restrict__ rp, int col0, int m) {
  PsPiece<TX> p;
  if constexpr (std::is_same<TX, double>::value) {
    if (VEC) {
      const int c = (FULLK || col0 < m) ? col0 : 0;
      p.a = *reinterpret_cast<const f64x2 *>(rp + c);
      p.b = *reinterpret_cast<const f64x2 *>(rp + c + 2);
    } else {
      p.a.x = rp[col0 < m ? col0 : 0];         p.a.y = rp[col0 + 1 < m ? col0 + 1 : 0];
      p.b.x = rp[col0 + 2 < m ? col0 + 2 : 0]; p.b.y = rp[col0 + 3 < m ? col0 + 3 : 0];
    }
  } else {
    if (VEC) {
      const int c = (FULLK || col0 < m) ? col0 : 0;
      const float4 v = *reinterpret_cast<const float4 *>(rp + c);
      p.x = v.x; p.y = v.y; p.z = v.z; p.w = v.w;
    } else {
      p.x = rp[col0 < m ? col0 : 0];         p.y = rp[col0 + 1 < m ? col0 + 1 : 0];
      p.z = rp[col0 + 2 < m ? col0 + 2 : 0]; p.w = rp[col0 + 3 < m ? col0 + 3 : 0];
    }
  }
  return p;
}

template <typename TX>
__device__ inline double ps_elem(const PsPiece<TX> &p, int t) {
  if constexpr (std::is_same<TX, double>::value) return t == 0 ? p.a.x : t == 1 ? p.a.y : t == 2 ? p.b.x : p.b.y;
  else return (double)(t == 0 ? p.x : t == 1 ? p.y : t == 2 ? p.z : p.w);
}

// position of column c inside a k-row of the image: column tiles come in pairs, [pair][li][tile & 1]
__host__ __device__ inline int ps_pos(int c) { return (c >> 5) * 32 + (c & 15) * 2 + ((c >> 4) & 1); }

// W (m x r row-major) -> image [nch * KC][16 RT] (zero padded) + column sums wbar[16 RT]
__global__ void ps_image_kernel(const double *__restrict__ W, int m, int r, int nrows, int nc, double *__restrict__ img,
                                double *__restrict__ wbar) {
  const int64_t total = (int64_t)nrows * nc;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(e / nc), c = (int)(e - (int64_t)k * nc);
    img[(int64_t)k * nc + ps_pos(c)] = (k < m && c < r) ? W[(int64_t)k * r + c] : 0.0;
  }
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < nc; c += blockDim.x) {
      double s = 0.0;
      if (c < r)
        for (int k = 0; k < m; ++k) s += W[(int64_t)k * r + c];
      wbar[c] = s;
    }
}

// NRM: the squared norms of the rows of Ur AS STORED (rounded to TU first) also go to nrm2[] (spr_qr_init_norms_*)
template <int RT, int NB, int VEC, bool FULLK, bool PRE, typename TX, typename TU, bool NRM>
__global__ __launch_bounds__(PS_THREADS) void project_stream_kernel(
    const TX *__restrict__ X, int64_t ldx, int m, int center_i, SegPlan plan, const double *__restrict__ inv_scale,
    const double *__restrict__ rowmean, const double *__restrict__ img, const double *__restrict__ wbar_g, int nch,
    int r, TU *__restrict__ Ur, int64_t ldu, double *__restrict__ nrm2) {
  constexpr int KC = PsK<TX>::KC, NJ = KC / 16, PS_ROWS = ps_rows(NB);
  constexpr int LDW = 16 * RT;                               // doubles per k-row of the image (a multiple of 32: see project_ws.hip)
  constexpr int CHUNK = KC * LDW;                            // doubles per chunk
  constexpr int NST = CHUNK / (2 * PS_THREADS);              // 16-byte pieces per thread and chunk
  static_assert(RT % 2 == 0 && CHUNK % (2 * PS_THREADS) == 0, "chunk must divide into one piece per thread and pass");
  __shared__ double Wl[2 * CHUNK];

  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kk = lane >> 4;
  const double isc = inv_scale[f];
  const bool centre_epi = !PRE && center_i != 0;

  const int64_t nsteps = (hi - lo + PS_ROWS - 1) / PS_ROWS;
  int64_t c = wl;
  if (c >= nsteps) return;                                   // workgroup-uniform: no barrier is left behind

  const f64x2 *img2 = reinterpret_cast<const f64x2 *>(img);
  f64x2 *Wl2 = reinterpret_cast<f64x2 *>(Wl);
  // chunk 0 -> buffer 0
#pragma unroll
  for (int i = 0; i < NST; ++i) Wl2[i * PS_THREADS + threadIdx.x] = img2[i * PS_THREADS + threadIdx.x];

  // B fragments of step (j, t), column tiles 2p and 2p+1: the two doubles at W[(16 j + 4 kk + t) * LDW + 32 p + 2 li]
  const double *wb0 = Wl + (4 * kk) * LDW + 2 * li;
  const double *wb1 = wb0 + CHUNK;

  auto row_ptr = [&](int64_t cc, int b) {
    int64_t row = lo + cc * PS_ROWS + 16 * (NB * wave + b) + li;
    row = row < hi ? row : hi - 1;
    return X + row * ldx;
  };
  auto pre_mean = [&](int64_t cc, int b) {                   // PRE: the mean of this lane's A row
    int64_t row = lo + cc * PS_ROWS + 16 * (NB * wave + b) + li;
    row = row < hi ? row : hi - 1;
    return (center_i != 0) ? rowmean[row] : 0.0;
  };

  PsPiece<TX> areg[NB][NJ];
  const TX *rp[NB];
  double mua[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    rp[b] = row_ptr(c, b);
    mua[b] = PRE ? pre_mean(c, b) : 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) areg[b][j] = ps_load<VEC, FULLK, TX>(rp[b], 16 * j + 4 * kk, m);
  }

  int g = 0;                                                 // chunks consumed so far (buffer parity)
  int chn = (nch > 1) ? 1 : 0;                               // chunk being staged during the current phase
  while (c < nsteps) {
    const int64_t cn = c + wpf;
    const TX *rpn[NB];
    double muan[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      rpn[b] = row_ptr(cn < nsteps ? cn : c, b);             // past the end: harmless re-read of the current rows
      muan[b] = PRE ? pre_mean(cn < nsteps ? cn : c, b) : 0.0;
    }
    f64x4 acc[NB][RT];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int ct = 0; ct < RT; ++ct) acc[b][ct] = (f64x4){0.0, 0.0, 0.0, 0.0};

    for (int ch = 0; ch < nch; ++ch) {
      __syncthreads();                                       // chunk g is in its buffer; every wave has left the other one
      const double *wb = (g & 1) ? wb1 : wb0;
      f64x2 *wdst = Wl2 + ((g & 1) ? 0 : CHUNK / 2);
      // next chunk of the image -> registers now, -> the other buffer at the end of the phase
      f64x2 wst[NST];
      {
        const f64x2 *src = img2 + (int64_t)chn * (CHUNK / 2);
#pragma unroll
        for (int i = 0; i < NST; ++i) wst[i] = src[i * PS_THREADS + threadIdx.x];
      }
      // where the A pieces of the next phase come from: the next chunk of the same rows, or chunk 0 of the next rows
      const bool lastc = (ch + 1 == nch);
      const int ncol = lastc ? 0 : (ch + 1) * KC;
      const TX *nb[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) nb[b] = lastc ? rpn[b] : rp[b];

      // One pair of B fragments per ds_read_b128, single-buffered: the read of pair p for step s+1 is issued right behind
      // the 2 NB MFMAs that consumed pair p in step s, and has the other 6 NB MFMAs (>= 768 cycles) to land.
      f64x2 bf[RT / 2];
#pragma unroll
      for (int p = 0; p < RT / 2; ++p) bf[p] = *reinterpret_cast<const f64x2 *>(wb + 32 * p);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        // the four A values of piece j per block are taken out of the load registers at the START of their use and the
        // registers re-loaded at once with the same piece of the next phase: every load has a whole phase to land, and the
        // wait hipcc places at the top of the phase (all loads but the youngest few) finds them done
        double av[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
          for (int t = 0; t < 4; ++t) av[b][t] = ps_elem<TX>(areg[b][j], t) - (PRE ? mua[b] : 0.0);
          areg[b][j] = ps_load<VEC, FULLK, TX>(nb[b], ncol + 16 * j + 4 * kk, m);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int sn = 4 * j + t + 1;                      // next step (wraps to 0: a harmless extra read)
          const int jn = (sn >> 2) % NJ, tn = sn & 3;
          const double *wj = wb + (16 * jn + tn) * LDW;
#pragma unroll
          for (int p = 0; p < RT / 2; ++p) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              acc[b][2 * p] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[b][t], bf[p].x, acc[b][2 * p], 0, 0, 0);
              acc[b][2 * p + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[b][t], bf[p].y, acc[b][2 * p + 1], 0, 0, 0);
            }
            bf[p] = *reinterpret_cast<const f64x2 *>(wj + 32 * p);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NB, 0);      // this pair's MFMAs,
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // then its re-load for the next step
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int i = 0; i < NST; ++i) wdst[i * PS_THREADS + threadIdx.x] = wst[i];
      ++g;
      chn = (chn + 1 == nch) ? 0 : chn + 1;
    }

    // epilogue: result rows kk + 4 i of each block, columns 16 ct + li
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int64_t blk0 = lo + c * PS_ROWS + 16 * (NB * wave + b);
      double mu[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t rr = blk0 + kk + 4 * i;
        mu[i] = centre_epi ? rowmean[rr < hi ? rr : hi - 1] : 0.0;
      }
      const bool full = (blk0 + 16 <= hi) && (r == 16 * RT);   // wave-uniform: whole block inside, no padded column
      TU *up = Ur + (blk0 + kk) * ldu + li;
      double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ct = 0; ct < RT; ++ct) {
        const double wbar = centre_epi ? wbar_g[16 * ct + li] : 0.0;
        const double s[4] = {(acc[b][ct].x - mu[0] * wbar) * isc, (acc[b][ct].y - mu[1] * wbar) * isc,
                             (acc[b][ct].z - mu[2] * wbar) * isc, (acc[b][ct].w - mu[3] * wbar) * isc};
        if (NRM) {                                           // padded columns hold exact zeros (zero columns of the image)
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
#pragma unroll
    for (int b = 0; b < NB; ++b) { rp[b] = rpn[b]; mua[b] = muan[b]; }
    c = cn;
  }
}

template <int RT, int NB, typename TX, typename TU>
int ps_launch(const TX *X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0, int64_t n_points, int32_t n_features,
              int center, const double *inv_scale, const double *rowmean, const double *W, int32_t r, TU *Ur, int64_t ldu,
              double *ws, double *nrm2, hipStream_t st) {
  constexpr int KC = PsK<TX>::KC, NC = 16 * RT;
  const int nch = (m + KC - 1) / KC;
  double *img = ws, *wbar = ws + (size_t)nch * KC * NC;
  hipLaunchKernelGGL(ps_image_kernel, dim3(64), dim3(256), 0, st, W, (int)m, (int)r, nch * KC, NC, img, wbar);
  SPR_LAUNCH_CHECK();
  const int cus = spr_cached_cus();
  SegPlan plan;
  plan.row0 = row0; plan.n_rows = n_rows; plan.n_points = n_points; plan.n_features = n_features;
  plan.total_wg = cus > 0 ? cus : 256;                     // registers allow two waves per SIMD: one workgroup per CU
  plan.chunk_rows = ps_rows(NB);
  const int grid = seg_total_wgs(plan);
  const bool vec = (m % 4 == 0) && ((sizeof(TX) * ldx) % 16 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  const bool fullk = vec && (m % KC == 0);
  const bool pre = center == 2;
#define PSKN(V, FK, PR, NR)                                                                                                  \
  hipLaunchKernelGGL((project_stream_kernel<RT, NB, V, FK, PR, TX, TU, NR>), dim3(grid), dim3(PS_THREADS), 0, st, X, ldx, (int)m, \
                     center, plan, inv_scale, rowmean, img, wbar, nch, (int)r, Ur, ldu, nrm2)
#define PSK(V, FK, PR) do { if (nrm2) PSKN(V, FK, PR, true); else PSKN(V, FK, PR, false); } while (0)
  if constexpr (RT > 8) {
    // the 16-tile form is only built for aligned rows and without the row-norm epilogue (its one user, the refinement
    // pass, needs neither); anything else goes in two column groups through the 8-tile form
    if (!vec || nrm2) return SPR_E_UNSUPPORTED;
    if (fullk) { if (pre) PSKN(1, true, true, false); else PSKN(1, true, false, false); }
    else { if (pre) PSKN(1, false, true, false); else PSKN(1, false, false, false); }
  } else {
    if (fullk) { if (pre) PSK(1, true, true); else PSK(1, true, false); }
    else if (vec) { if (pre) PSK(1, false, true); else PSK(1, false, false); }
    else { if (pre) PSK(0, false, true); else PSK(0, false, false); }
  }
#undef PSK
#undef PSKN
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

template <typename TX>
size_t ps_workspace(int32_t m, int32_t r) {
  if (m <= 0 || r <= 0 || r > SPR_MAX_R_STREAM) return 0;
  constexpr int KC = PsK<TX>::KC;
  const int rt = (r <= 64) ? 4 : (r <= SPR_MAX_R) ? 8 : 16;
  const size_t nch = (size_t)(m + KC - 1) / KC;
  return sizeof(double) * (nch * KC * 16 * rt + 16 * rt);
}

template <typename TX, typename TU>
int ps_entry(const char *who, const TX *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0, int64_t n_points,
             int32_t n_features, int32_t center, const double *d_inv_scale, const double *d_rowmean, const double *d_W,
             int32_t r, TU *d_Ur, int64_t ldu, void *d_workspace, size_t workspace_bytes, void *stream,
             double *d_rownorm2 = nullptr) {
  SPR_REQUIRE(d_X && d_inv_scale && d_W && d_Ur && d_workspace && (d_rowmean || !center), SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(n_rows > 0 && m > 0 && ldx >= m, SPR_E_INVALID, "%s: bad shape", who);
  SPR_REQUIRE(r > 0 && ldu >= r, SPR_E_INVALID, "%s: bad r=%d (m=%d ldu=%lld)", who, r, m, (long long)ldu);
  SPR_REQUIRE(center >= 0 && center <= 2, SPR_E_INVALID, "%s: centre mode must be 0, 1 (epilogue) or 2 (registers)", who);
  SPR_REQUIRE(n_points > 0 && n_features > 0 && row0 >= 0 && row0 + n_rows <= n_points * (int64_t)n_features,
              SPR_E_INVALID, "%s: bad feature layout", who);
  constexpr bool wide_ok = std::is_same<TU, double>::value;      // 129..256 columns per launch: float64 output only
  SPR_REQUIRE(r <= (wide_ok ? SPR_MAX_R_STREAM : SPR_MAX_R), SPR_E_UNSUPPORTED,
              "%s: r=%d > %d per call (project wider bases in column groups)", who, r, wide_ok ? SPR_MAX_R_STREAM : SPR_MAX_R);
  SPR_REQUIRE(workspace_bytes >= ps_workspace<TX>(m, r), SPR_E_WORKSPACE, "%s: workspace %zu < %zu", who, workspace_bytes,
              ps_workspace<TX>(m, r));
  SPR_REQUIRE((reinterpret_cast<uintptr_t>(d_workspace) & 15) == 0, SPR_E_INVALID, "%s: workspace must be 16-byte aligned", who);
  hipStream_t st = static_cast<hipStream_t>(stream);
  double *ws = static_cast<double *>(d_workspace);
  if (r <= 64)
    return ps_launch<4, 2, TX, TU>(d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale, d_rowmean, d_W, r,
                                   d_Ur, ldu, ws, d_rownorm2, st);
  if (r <= SPR_MAX_R)
    return ps_launch<8, 2, TX, TU>(d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale, d_rowmean, d_W, r,
                                   d_Ur, ldu, ws, d_rownorm2, st);
  if constexpr (wide_ok) {
    const int rc = ps_launch<16, 1, TX, TU>(d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale, d_rowmean,
                                            d_W, r, d_Ur, ldu, ws, d_rownorm2, st);
    if (rc == SPR_E_UNSUPPORTED) spr_set_error("%s: r=%d in one launch needs 16-byte-aligned rows, m %% 4 == 0 and no row norms", who, r);
    return rc;
  }
  return SPR_E_UNSUPPORTED;
}

}  // namespace

extern "C" size_t spr_project_stream_workspace(int32_t m, int32_t r, int32_t x_is_f32) {
  return x_is_f32 ? ps_workspace<float>(m, r) : ps_workspace<double>(m, r);
}

extern "C" int spr_project_stream_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                      int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                                      const double *d_rowmean, const double *d_W, int32_t r, double *d_Ur, int64_t ldu,
                                      void *d_workspace, size_t workspace_bytes, void *stream) {
  return ps_entry("spr_project_stream_f64", d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,
                  d_rowmean, d_W, r, d_Ur, ldu, d_workspace, workspace_bytes, stream);
}

extern "C" int spr_project_stream_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                      int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                                      const double *d_rowmean, const double *d_W, int32_t r, float *d_Ur, int64_t ldu,
                                      void *d_workspace, size_t workspace_bytes, void *stream) {
  return ps_entry("spr_project_stream_x32", d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,
                  d_rowmean, d_W, r, d_Ur, ldu, d_workspace, workspace_bytes, stream);
}

extern "C" int spr_project_stream_x32_f64out(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                             int64_t n_points, int32_t n_features, int32_t center,
                                             const double *d_inv_scale, const double *d_rowmean, const double *d_W,
                                             int32_t r, double *d_Ur, int64_t ldu, void *d_workspace,
                                             size_t workspace_bytes, void *stream) {
  return ps_entry("spr_project_stream_x32_f64out", d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,
                  d_rowmean, d_W, r, d_Ur, ldu, d_workspace, workspace_bytes, stream);
}

// ---- the same launches, also leaving the squared norms of the rows they store (d_rownorm2[n_rows], of the values rounded
// to the basis type): the first sweep of optimal_placement then reads 8 bytes per row instead of the whole basis
// (spr_qr_init_norms_*).  Every shape the plain entry points take.
extern "C" int spr_project_stream_norms_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                            int64_t n_points, int32_t n_features, int32_t center,
                                            const double *d_inv_scale, const double *d_rowmean, const double *d_W,
                                            int32_t r, double *d_Ur, int64_t ldu, double *d_rownorm2, void *d_workspace,
                                            size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(d_rownorm2, SPR_E_INVALID, "spr_project_stream_norms_f64: NULL norm vector");
  return ps_entry("spr_project_stream_norms_f64", d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,
                  d_rowmean, d_W, r, d_Ur, ldu, d_workspace, workspace_bytes, stream, d_rownorm2);
}

extern "C" int spr_project_stream_norms_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                            int64_t n_points, int32_t n_features, int32_t center,
                                            const double *d_inv_scale, const double *d_rowmean, const double *d_W,
                                            int32_t r, float *d_Ur, int64_t ldu, double *d_rownorm2, void *d_workspace,
                                            size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(d_rownorm2, SPR_E_INVALID, "spr_project_stream_norms_x32: NULL norm vector");
  return ps_entry("spr_project_stream_norms_x32", d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_inv_scale,
                  d_rowmean, d_W, r, d_Ur, ldu, d_workspace, workspace_bytes, stream, d_rownorm2);
}

extern "C" int spr_project_stream_norms_x32_f64out(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                                   int64_t n_points, int32_t n_features, int32_t center,
                                                   const double *d_inv_scale, const double *d_rowmean,
                                                   const double *d_W, int32_t r, double *d_Ur, int64_t ldu,
                                                   double *d_rownorm2, void *d_workspace, size_t workspace_bytes,
                                                   void *stream) {
  SPR_REQUIRE(d_rownorm2, SPR_E_INVALID, "spr_project_stream_norms_x32_f64out: NULL norm vector");
  return ps_entry("spr_project_stream_norms_x32_f64out", d_X, n_rows, m, ldx, row0, n_points, n_features, center,
                  d_inv_scale, d_rowmean, d_W, r, d_Ur, ldu, d_workspace, workspace_bytes, stream, d_rownorm2);
}
