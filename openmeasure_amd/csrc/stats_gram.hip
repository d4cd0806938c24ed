// K1 + K3a: fused row mean / per-feature statistics / per-feature Gram matrix.
//
// One pass over the row shard.  Persistent 512-thread workgroups are dealt to the feature
// segments of the shard; each keeps the upper-triangular 16x16 tiles of the m x m Gram
// matrix of its feature in MFMA accumulators (v_mfma_f64_16x16x4_f64) for its whole
// life, consuming R-row panels that are centred on the way into LDS (rowtile.hpp).
// At the end every wave stores its tiles to a slab; a second tiny kernel sums the slabs
// in a fixed order (bitwise reproducible, no float atomics), mirrors the lower triangle
// and Chan-merges the row-mean statistics.
//
// MFMA operand pattern.  G[i][j] += sum_k P[k][i] P[k][j] over panel rows k.  For
// v_mfma_f64_16x16x4_f64 lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15],
// so both operands of tile (ti,tj) are "16 consecutive doubles of panel row k0+(l>>4)":
// conflict-free ds_read_b64 when consecutive rows sit 32 banks apart (MP == 16 mod 32).
// The result lane map is col = l&15, row = (l>>4) + 4*reg.
#include <stdlib.h>

#include <type_traits>

#include "rowtile.hpp"

#ifndef GRAM_UNIT_PAIRING
#define GRAM_UNIT_PAIRING 1
#endif
#ifndef GRAM_ABLATE
#define GRAM_ABLATE 0   // diagnostic builds: 1 = no MFMAs, 2 = no centring/loads in the loop, 3 = loads + raw LDS stores, no centring arithmetic
#endif
#if GRAM_ABLATE == 1
#define GRAM_MFMA(a, b, c) ({ asm volatile("" ::"v"(a), "v"(b)); (c); })
#else
#define GRAM_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#endif

namespace {

template <int MT>
struct GramShape {
  static constexpr int MPAD = 16 * MT;
  static constexpr int MP = MPAD + ((MT % 2 == 0) ? 16 : 0);
  static constexpr int T = MT * (MT + 1) / 2;
  static constexpr int NU = (MT + 1) / 2;  // wave "units": tile rows u and MT-1-u together
};

// waves per workgroup, k slices and panel height per padded width (NW = NU * KS)
template <int MT> struct GramCfg;
template <> struct GramCfg<1>  { static constexpr int NW = 8,  R = 64, KS = 8; static constexpr bool DBUF = true; static constexpr int LPRMAX = 16; };
template <> struct GramCfg<2>  { static constexpr int NW = 8,  R = 32, KS = 8; static constexpr bool DBUF = true; static constexpr int LPRMAX = 16; };
template <> struct GramCfg<3>  { static constexpr int NW = 8,  R = 64, KS = 4; static constexpr bool DBUF = true; static constexpr int LPRMAX = 16; };
// m in 49..64: 8 lanes per row (3 butterfly steps, 8 rows per wave instruction) on 64-row panels -- the centring pass is
// what the short rows pay for (4 adds + 12 butterfly instructions + 4 subs per 4 rows with 16 lanes): 0.53 -> 0.46 ms
// per 2.05 GB (4.0 -> 4.5 TB/s)
#ifndef GRAM4_R
#define GRAM4_R 64
#define GRAM4_LPR 8
#endif
template <> struct GramCfg<4>  { static constexpr int NW = 8,  R = GRAM4_R, KS = 4; static constexpr bool DBUF = true; static constexpr int LPRMAX = GRAM4_LPR; };
template <> struct GramCfg<6>  { static constexpr int NW = 12, R = 48, KS = 4; static constexpr bool DBUF = true; static constexpr int LPRMAX = 16; };
template <> struct GramCfg<8>  { static constexpr int NW = 8,  R = 32, KS = 2; static constexpr bool DBUF = true; static constexpr int LPRMAX = 16; };
template <> struct GramCfg<12> { static constexpr int NW = 12, R = 24, KS = 2; static constexpr bool DBUF = false; static constexpr int LPRMAX = 32; };
#ifndef GRAM_DBUF16
#define GRAM_DBUF16 false
#endif
template <> struct GramCfg<16> { static constexpr int NW = 8,  R = 32, KS = 1; static constexpr bool DBUF = GRAM_DBUF16; static constexpr int LPRMAX = 16; };

// Slabs a workgroup writes: its KS k-slices hold partial sums of the SAME tiles; for the narrow shapes (KS >= 4, where the
// slabs are a visible share of the HBM traffic of an HBM-bound pass: 84 MB written and read again next to the 2 GB of
// config 2) they are added up inside the workgroup, through the dead panel buffers, before one slab goes out.
template <int MT> struct GramOut { static constexpr int KSO = (GramCfg<MT>::KS >= 4) ? 1 : GramCfg<MT>::KS; };

// linear index over the upper triangle (row-major) <-> tile row / column
constexpr int tri_index(int mt, int ti, int tj) { return ti * mt - ti * (ti - 1) / 2 + (tj - ti); }

template <int MT>
__device__ inline void tile_coords(int idx, int &ti, int &tj) {
  ti = 0;
  while (ti < MT - 1 && idx >= MT - ti) { idx -= MT - ti; ++ti; }
  tj = ti + idx;
  if (tj > MT - 1) tj = MT - 1;
}

// The whole life of one wave: unit U owns tile rows RA = U (tiles (RA, RA..MT-1)) and, when
// different, RB = MT-1-U (tiles (RB, RB..MT-1)) -- MT+1 tiles for every unit, so the eight
// waves of the m = 256 case are perfectly balanced (17 tiles each).  Per k step the wave
// reads the MT-RA operand fragments of column blocks RA..MT-1 once (every fragment is both
// an A and a B operand) and issues its MFMAs from registers; the fragments of step k+1 are
// requested before the MFMAs of step k so LDS latency hides behind the 64-cycle MFMAs.
template <int MT, int U, int VEC, typename TX>
__device__ inline void gram_wave(const TX *__restrict__ X, int64_t ldx, int m, int center,
                                 int64_t lo, int64_t hi, int wl, int wpf, int ks, int wave, int lane,
                                 double *__restrict__ lds0, double *__restrict__ lds1,
                                 double *__restrict__ rowmean, double *__restrict__ stat_part,
                                 double *__restrict__ slab, double *__restrict__ rowsum) {
  using S = GramShape<MT>;
  using C = GramCfg<MT>;
  constexpr int R = C::R, KS = C::KS, NW = C::NW, MP = S::MP, T = S::T;
  constexpr int KROWS = R / KS, KSTEPS = KROWS / 4;
  constexpr int RA = U, RB = MT - 1 - U;
  constexpr int NA = MT - RA;                 // tiles in row RA == operand fragments per k step
  constexpr int NB = (RB != RA) ? MT - RB : 0;
  static_assert(KROWS % 4 == 0, "k slice must be a multiple of the MFMA depth");
  using RT = RowTile<MT, R, MP, NW, C::LPRMAX, TX>;

  f64x4 accA[NA];
  f64x4 accB[NB > 0 ? NB : 1];
#pragma unroll
  for (int j = 0; j < NA; ++j) accA[j] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < NB; ++j) accB[j] = (f64x4){0.0, 0.0, 0.0, 0.0};

  RowStats st;
  st.init();
  RT tile;

  // Software pipeline, one barrier per panel: while the MFMAs of panel c run from one LDS
  // buffer, the same wave centres panel c+1 (already in registers) into the other buffer,
  // one row pass between k steps, and then requests panel c+2 from HBM.
  const int64_t nchunks = (hi - lo + R - 1) / R;
  int64_t c = wl;
  // centre mode 2 (column slice of a wider X): the row means come from rowmean[].  The load is issued in
  // every mode (the array is always valid memory; the value is only selected in mode 2) so that the loop
  // body keeps one shape and stays a single basic block.
  const double *mean_in = rowmean;
  tile.template load<VEC>(X, ldx, m, lo + c * R, hi, wave, lane, mean_in);
  tile.template center_store<true>(lds0, m, center, lo + c * R, hi, wave, lane, rowmean, &st, rowsum);
  int64_t cn = c + wpf;
  int64_t nrow0 = (cn < nchunks) ? lo + cn * R : hi;      // past-the-end panel: every row invalid
  tile.template load<VEC>(X, ldx, m, nrow0, hi, wave, lane, mean_in);
  int buf = 0;
  const int frag = (lane >> 4) * MP + (lane & 15) + ks * KROWS * MP + RA * 16;
  while (c < nchunks) {
    double *cur = buf ? lds1 : lds0;
    double *nxt = buf ? lds0 : lds1;
    const int64_t c2 = cn + wpf;
    const int64_t n2row0 = (c2 < nchunks) ? lo + c2 * R : hi;
    __syncthreads();
    const double *p = cur + frag;
    if constexpr (C::DBUF) {
      double op[2][NA];
#pragma unroll
      for (int j = 0; j < NA; ++j) op[0][j] = p[16 * j];
#pragma unroll
      for (int k = 0; k < KSTEPS; ++k) {
        if (k + 1 < KSTEPS) {
#pragma unroll
          for (int j = 0; j < NA; ++j) op[(k + 1) & 1][j] = p[(k + 1) * 4 * MP + 16 * j];
        }
#pragma unroll
        for (int j = 0; j < NA; ++j)
          accA[j] = GRAM_MFMA(op[k & 1][0], op[k & 1][j], accA[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j)
          accB[j] = GRAM_MFMA(op[k & 1][RB - RA], op[k & 1][RB - RA + j], accB[j]);
#pragma unroll
        for (int it = 0; it < RT::IT; ++it)
          if (GRAM_ABLATE != 2 && (it * KSTEPS) / RT::IT == k) {
            if (GRAM_ABLATE == 3) tile.raw_store_pass(it, nxt, m, nrow0, hi, wave, lane);
            else tile.template center_store_pass<true>(it, nxt, m, center, nrow0, hi, wave, lane, rowmean, &st, rowsum);
            tile.template load_pass<VEC>(it, X, ldx, m, n2row0, hi, wave, lane, mean_in);   // panel c+2 into the freed registers
          }
      }
    } else {  // register-tight shapes: one operand set, three waves per SIMD cover the LDS latency
#pragma unroll
      for (int k = 0; k < KSTEPS; ++k) {
        double op[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) op[j] = p[k * 4 * MP + 16 * j];
#pragma unroll
        for (int j = 0; j < NA; ++j)
          accA[j] = GRAM_MFMA(op[0], op[j], accA[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j)
          accB[j] = GRAM_MFMA(op[RB - RA], op[RB - RA + j], accB[j]);
#pragma unroll
        for (int it = 0; it < RT::IT; ++it)
          if (GRAM_ABLATE != 2 && (it * KSTEPS) / RT::IT == k) {
            if (GRAM_ABLATE == 3) tile.raw_store_pass(it, nxt, m, nrow0, hi, wave, lane);
            else tile.template center_store_pass<true>(it, nxt, m, center, nrow0, hi, wave, lane, rowmean, &st, rowsum);
            tile.template load_pass<VEC>(it, X, ldx, m, n2row0, hi, wave, lane, mean_in);   // panel c+2 into the freed registers
          }
      }
    }
    buf ^= 1;
    c = cn;
    cn = c2;
    nrow0 = n2row0;
  }

  constexpr int KSO = GramOut<MT>::KSO;
  if constexpr (KSO != KS) {
    // k-slice s adds its tiles onto slice s - 1, s = KS - 1 .. 1, through LDS (T x 2 KB: fits the two panel buffers, which
    // are contiguous); fixed order, every wave of the workgroup passes the same 2 (KS - 1) barriers
    static_assert((size_t)T * 256 <= (size_t)2 * R * MP, "tile staging must fit the panel buffers");
    double *red = lds0 + lane;
    for (int s = KS - 1; s >= 1; --s) {
      __syncthreads();                                     // panels (first round) / the previous round's reads are done
      if (ks == s) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          double *tp = red + tri_index(MT, RA, RA + j) * 256;
          tp[0] = accA[j].x; tp[64] = accA[j].y; tp[128] = accA[j].z; tp[192] = accA[j].w;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          double *tp = red + tri_index(MT, RB, RB + j) * 256;
          tp[0] = accB[j].x; tp[64] = accB[j].y; tp[128] = accB[j].z; tp[192] = accB[j].w;
        }
      }
      __syncthreads();
      if (ks == s - 1) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const double *tp = red + tri_index(MT, RA, RA + j) * 256;
          accA[j].x += tp[0]; accA[j].y += tp[64]; accA[j].z += tp[128]; accA[j].w += tp[192];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const double *tp = red + tri_index(MT, RB, RB + j) * 256;
          accB[j].x += tp[0]; accB[j].y += tp[64]; accB[j].z += tp[128]; accB[j].w += tp[192];
        }
      }
    }
  }
  // tiles -> slab[(block*KSO + slice)][tile][reg][lane]
  if (KSO == KS || ks == 0) {
    double *sp = slab + ((int64_t)blockIdx.x * KSO + (KSO == KS ? ks : 0)) * T * 256 + lane;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      double *tp = sp + (int64_t)tri_index(MT, RA, RA + j) * 256;
      tp[0] = accA[j].x; tp[64] = accA[j].y; tp[128] = accA[j].z; tp[192] = accA[j].w;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      double *tp = sp + (int64_t)tri_index(MT, RB, RB + j) * 256;
      tp[0] = accB[j].x; tp[64] = accB[j].y; tp[128] = accB[j].z; tp[192] = accB[j].w;
    }
  }
  // Welford partials: one slot per lane group
  if ((lane % RT::LPR) == 0) {
    double *q = stat_part + ((int64_t)blockIdx.x * RT::ROWS_PER_IT + wave * RT::RPW + lane / RT::LPR) * 3;
    q[0] = st.cnt; q[1] = st.mean(); q[2] = st.m2();
  }
}

__device__ inline void chan_merge(double &n, double &mu, double &m2, double nb, double mb, double sb) {
  if (nb > 0.0) {
    const double tot = n + nb, d = mb - mu;
    mu += d * nb / tot;
    m2 += sb + d * d * n * nb / tot;
    n = tot;
  }
}

// Own-means lane of gram_wave (centre mode 1, m == 16 MT, packed 16-byte-aligned rows: the shape of the BASELINE
// workloads): the same tiles, operand reads and one-barrier pipeline, but the staging goes through load_pass_own /
// center_store_own of rowtile.hpp (pointer offsets instead of a 64-bit row multiply, no centre-mode selects, no mean load,
// no running statistics: gram_rowmean_stats_kernel forms those from the n row means afterwards).
template <int MT, int U, typename TX, bool EXT, bool SUMS>
__device__ inline void gram_wave_own(const TX *__restrict__ X, int64_t ldx, int64_t lo, int64_t hi, int wl, int wpf, int ks,
                                     int wave, int lane, double *__restrict__ lds0, double *__restrict__ lds1,
                                     double *__restrict__ rowmean, double *__restrict__ slab, double *__restrict__ rowsum) {
  using S = GramShape<MT>;
  using C = GramCfg<MT>;
  constexpr int R = C::R, KS = C::KS, NW = C::NW, MP = S::MP, T = S::T;
  constexpr int KROWS = R / KS, KSTEPS = KROWS / 4;
  constexpr int RA = U, RB = MT - 1 - U;
  constexpr int NA = MT - RA;
  constexpr int NB = (RB != RA) ? MT - RB : 0;
  using RT = RowTile<MT, R, MP, NW, C::LPRMAX, TX>;

  f64x4 accA[NA];
  f64x4 accB[NB > 0 ? NB : 1];
#pragma unroll
  for (int j = 0; j < NA; ++j) accA[j] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < NB; ++j) accB[j] = (f64x4){0.0, 0.0, 0.0, 0.0};

  RT tile;
  const int64_t nchunks = (hi - lo + R - 1) / R;
  const int64_t lane_off = (int64_t)(wave * RT::RPW + lane / RT::LPR) * ldx;
  auto base_of = [&](int64_t row0) { return X + row0 * ldx; };       // wave-uniform: scalar arithmetic
  int64_t c = wl;
  int64_t crow0 = lo + c * R;
#pragma unroll
  for (int it = 0; it < RT::IT; ++it) tile.template load_pass_own<EXT>(it, base_of(crow0), ldx, lane_off, hi - crow0, wave, lane, rowmean, crow0);
  if (crow0 + R <= hi) {
#pragma unroll
    for (int it = 0; it < RT::IT; ++it) tile.template center_store_own<true, EXT, SUMS>(it, lds0, crow0, hi - crow0, wave, lane, rowmean, rowsum);
  } else {
#pragma unroll
    for (int it = 0; it < RT::IT; ++it) tile.template center_store_own<false, EXT, SUMS>(it, lds0, crow0, hi - crow0, wave, lane, rowmean, rowsum);
  }
  int64_t cn = c + wpf;
  int64_t nrow0 = (cn < nchunks) ? lo + cn * R : hi;
#pragma unroll
  for (int it = 0; it < RT::IT; ++it) tile.template load_pass_own<EXT>(it, base_of(nrow0 < hi ? nrow0 : lo), ldx, lane_off, hi - nrow0, wave, lane, rowmean, nrow0 < hi ? nrow0 : lo);
  int buf = 0;
  const int frag = (lane >> 4) * MP + (lane & 15) + ks * KROWS * MP + RA * 16;
  while (c < nchunks) {
    double *cur = buf ? lds1 : lds0;
    double *nxt = buf ? lds0 : lds1;
    const int64_t c2 = cn + wpf;
    const int64_t n2row0 = (c2 < nchunks) ? lo + c2 * R : hi;
    const TX *n2base = base_of(n2row0 < hi ? n2row0 : lo);
    const bool nfull = nrow0 + R <= hi;                    // wave-uniform
    if (GRAM_ABLATE != 4) __syncthreads();               // GRAM_ABLATE=4 (diagnostic, wrong results): what the panel barrier costs
    const double *p = cur + frag;
    // (Round 4 measured a staggered slot for the SIMD partners -- units >= 4 staging half a panel later, same barrier: 9.53
    // vs 9.56 ms per 18.4 GB and 48.24 vs 48.21 ms per 92 GB, i.e. nothing: profiles/r04_gram_stage_stagger_ab.txt.)
    auto stage = [&](auto full_tag, int k) {
      constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
      for (int it = 0; it < RT::IT; ++it)
        if ((it * KSTEPS) / RT::IT == k) {
          tile.template center_store_own<FULL, EXT, SUMS>(it, nxt, nrow0, hi - nrow0, wave, lane, rowmean, rowsum);
          tile.template load_pass_own<EXT>(it, n2base, ldx, lane_off, hi - n2row0, wave, lane, rowmean, n2row0 < hi ? n2row0 : lo);   // panel c+2
        }
    };
    if constexpr (C::DBUF) {
      double op[2][NA];
#pragma unroll
      for (int j = 0; j < NA; ++j) op[0][j] = p[16 * j];
#pragma unroll
      for (int k = 0; k < KSTEPS; ++k) {
        if (k + 1 < KSTEPS) {
#pragma unroll
          for (int j = 0; j < NA; ++j) op[(k + 1) & 1][j] = p[(k + 1) * 4 * MP + 16 * j];
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) accA[j] = GRAM_MFMA(op[k & 1][0], op[k & 1][j], accA[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) accB[j] = GRAM_MFMA(op[k & 1][RB - RA], op[k & 1][RB - RA + j], accB[j]);
        if (nfull) stage(std::true_type{}, k); else stage(std::false_type{}, k);
      }
    } else {
#pragma unroll
      for (int k = 0; k < KSTEPS; ++k) {
        double op[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) op[j] = p[k * 4 * MP + 16 * j];
#pragma unroll
        for (int j = 0; j < NA; ++j) accA[j] = GRAM_MFMA(op[0], op[j], accA[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) accB[j] = GRAM_MFMA(op[RB - RA], op[RB - RA + j], accB[j]);
        if (nfull) stage(std::true_type{}, k); else stage(std::false_type{}, k);
      }
    }
    buf ^= 1;
    c = cn;
    cn = c2;
    nrow0 = n2row0;
  }
  double *sp = slab + ((int64_t)blockIdx.x * KS + ks) * T * 256 + lane;
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    double *tp = sp + (int64_t)tri_index(MT, RA, RA + j) * 256;
    tp[0] = accA[j].x; tp[64] = accA[j].y; tp[128] = accA[j].z; tp[192] = accA[j].w;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double *tp = sp + (int64_t)tri_index(MT, RB, RB + j) * 256;
    tp[0] = accB[j].x; tp[64] = accB[j].y; tp[128] = accB[j].z; tp[192] = accB[j].w;
  }
}

template <int MT, typename TX, bool EXT, bool SUMS>
__global__ __launch_bounds__(GramCfg<MT>::NW * 64) void stats_gram_own_kernel(
    const TX *__restrict__ X, int64_t ldx, SegPlan plan, double *__restrict__ rowmean, double *__restrict__ slab,
    double *__restrict__ rowsum) {
  using S = GramShape<MT>;
  using C = GramCfg<MT>;
  constexpr int NU = S::NU;
  __shared__ double lds[2][C::R * S::MP];
  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ks = wave / NU;
  // waves w and w + 4 share a SIMD: unit u reads 16 - u operand fragments per k step, so the pairs (u, 7 - u) give every SIMD
  // the same LDS-read count (25 of 100) instead of 28 / 26 / 24 / 22
  const int unit = (GRAM_UNIT_PAIRING && NU == 8 && C::KS == 1) ? ((wave < 4) ? wave : 11 - wave) : wave % NU;
#define GRAM_UNIT(UV)                                                                                          \
  case UV:                                                                                                     \
    if constexpr (UV < NU)                                                                                     \
      gram_wave_own<MT, UV, TX, EXT, SUMS>(X, ldx, lo, hi, wl, wpf, ks, wave, lane, lds[0], lds[1], rowmean, slab, rowsum); \
    break;
  switch (unit) {
    GRAM_UNIT(0) GRAM_UNIT(1) GRAM_UNIT(2) GRAM_UNIT(3) GRAM_UNIT(4) GRAM_UNIT(5) GRAM_UNIT(6) GRAM_UNIT(7)
    default: break;
  }
#undef GRAM_UNIT
}

// Per-feature statistics (count, mean, M2) of the row means the own-means kernel has written, in the slot layout
// gram_finalize_kernel merges: same grid and SegPlan as the Gram launch, block b reduces the means of its feature's rows
// b, b + wpf, ... (1024-row chunks) to ONE triple in its first slot (fixed order: reproducible) and zeroes its other slots.
constexpr int RS_THREADS = 1024;
__global__ __launch_bounds__(RS_THREADS) void gram_rowmean_stats_kernel(const double *__restrict__ rowmean, SegPlan plan,
                                                                        int slots_per_wg, double *__restrict__ stat_part) {
  __shared__ double sn[RS_THREADS / 64], smu[RS_THREADS / 64], sm2[RS_THREADS / 64];
  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a latency-bound read of 8 bytes per row: four independent rows in flight per thread, pushed in a fixed order
  RowStats st;
  st.init();
  const int64_t step = (int64_t)wpf * RS_THREADS;
  int64_t row = lo + (int64_t)wl * RS_THREADS + threadIdx.x;
  for (; row + 3 * step < hi; row += 4 * step) {
    const double a = rowmean[row], b = rowmean[row + step], c = rowmean[row + 2 * step], d = rowmean[row + 3 * step];
    st.push(a, true); st.push(b, true); st.push(c, true); st.push(d, true);
  }
  for (; row < hi; row += step) st.push(rowmean[row], true);
  double n = st.cnt, mu = st.mean(), m2 = st.m2();
  for (int o = 32; o > 0; o >>= 1) {
    const double on = __shfl_down(n, o, 64), om = __shfl_down(mu, o, 64), os = __shfl_down(m2, o, 64);
    chan_merge(n, mu, m2, on, om, os);
  }
  if (lane == 0) { sn[wave] = n; smu[wave] = mu; sm2[wave] = m2; }
  __syncthreads();
  double *q = stat_part + (int64_t)blockIdx.x * slots_per_wg * 3;
  if (threadIdx.x == 0) {
    n = 0.0; mu = 0.0; m2 = 0.0;
    for (int w = 0; w < RS_THREADS / 64; ++w) chan_merge(n, mu, m2, sn[w], smu[w], sm2[w]);
    q[0] = n; q[1] = mu; q[2] = m2;
  }
  for (int e = 3 + threadIdx.x; e < slots_per_wg * 3; e += RS_THREADS) q[e] = 0.0;
}

template <int MT, int VEC, typename TX>
__global__ __launch_bounds__(GramCfg<MT>::NW * 64) void stats_gram_kernel(
    const TX *__restrict__ X, int64_t ldx, int m, int center_i, SegPlan plan,
    double *__restrict__ rowmean, double *__restrict__ stat_part, double *__restrict__ slab, double *__restrict__ rowsum) {
  using S = GramShape<MT>;
  using C = GramCfg<MT>;
  constexpr int NU = S::NU;
  static_assert(NU * C::KS == C::NW, "waves = units x k slices");
  __shared__ double lds[2][C::R * S::MP];

  int f, wl, wpf, base;
  int64_t lo, hi;
  if (!seg_locate(plan, blockIdx.x, f, wl, wpf, base, lo, hi)) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int unit = wave % NU, ks = wave / NU;
#define GRAM_UNIT(UV)                                                                                        \
  case UV:                                                                                                   \
    if constexpr (UV < NU)                                                                                   \
      gram_wave<MT, UV, VEC, TX>(X, ldx, m, center_i, lo, hi, wl, wpf, ks, wave, lane, lds[0],    \
                        lds[1], rowmean, stat_part, slab, rowsum);                                           \
    break;
  switch (unit) {
    GRAM_UNIT(0) GRAM_UNIT(1) GRAM_UNIT(2) GRAM_UNIT(3) GRAM_UNIT(4) GRAM_UNIT(5) GRAM_UNIT(6) GRAM_UNIT(7)
    default: break;
  }
#undef GRAM_UNIT
}

// grid (T, n_features), 1024 threads: fixed-order sum of the slabs of feature f for one tile.
// Thread (g, e) adds the partials p = g, g+4, ... of tile element e (four independent load
// streams per thread); the four group sums are then added in group order through LDS, so
// the result does not depend on scheduling.
template <int MT>
__global__ __launch_bounds__(1024) void gram_finalize_kernel(
    const double *__restrict__ slab, const double *__restrict__ stat_part, int m, SegPlan plan,
    int slots_per_wg, double *__restrict__ gram, double *__restrict__ fstats, int ldg, int origin) {
  constexpr int T = GramShape<MT>::T;
  constexpr int KS = GramOut<MT>::KSO;                 // slabs per workgroup
  __shared__ double red[4][256];
  const int f = blockIdx.y;
  const int tile = blockIdx.x;
  // blocks of feature f
  int base = 0, wpf = 0;
  {
    int acc = 0;
    const int f0 = seg_first_feature(plan), f1 = seg_last_feature(plan);
    for (int ff = f0; ff <= f1; ++ff) {
      int64_t lo, hi;
      seg_range(plan, ff, lo, hi);
      const int w = seg_wgs(plan, hi - lo);
      if (ff == f) { base = acc; wpf = w; }
      acc += w;
    }
  }
  int ti, tj;
  tile_coords<MT>(tile, ti, tj);
  const int e = threadIdx.x & 255, g = threadIdx.x >> 8;
  const int np = wpf * KS;
  const double *sp = slab + ((int64_t)base * KS * T + tile) * 256 + e;
  const int64_t stride = (int64_t)T * 256;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int p = g;
  for (; p + 12 < np; p += 16) {
    s0 += sp[p * stride];
    s1 += sp[(p + 4) * stride];
    s2 += sp[(p + 8) * stride];
    s3 += sp[(p + 12) * stride];
  }
  for (; p < np; p += 4) s0 += sp[p * stride];
  red[g][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0) {
    const double sum = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    const int l = e & 63, reg = e >> 6;
    const int gi = ti * 16 + (l >> 4) + 4 * reg, gj = tj * 16 + (l & 15);
    if (gi < m && gj < m) {
      double *G = gram + (int64_t)f * ldg * ldg + (int64_t)origin * ldg + origin;   // block (origin, origin) of an ldg x ldg matrix
      G[(int64_t)gi * ldg + gj] = sum;
      if (ti != tj) G[(int64_t)gj * ldg + gi] = sum;
    }
  }
  if (tile == 0) {  // Chan merge of the Welford partials: strided per thread, then a fixed binary tree
    __syncthreads();
    double n = 0.0, mu = 0.0, m2 = 0.0;
    const double *q = stat_part + (int64_t)base * slots_per_wg * 3;
    const int tot_slots = wpf * slots_per_wg;
    for (int pp = threadIdx.x; pp < tot_slots; pp += 1024) chan_merge(n, mu, m2, q[3 * pp], q[3 * pp + 1], q[3 * pp + 2]);
    double *sn = &red[0][0], *smu = &red[1][0], *sm2 = &red[2][0];  // 1024 doubles each would not fit: fold to 256 first
    for (int o = 32; o > 0; o >>= 1) {                               // wave tree (lane i <- lane i+o)
      const double on = __shfl_down(n, o, 64), om = __shfl_down(mu, o, 64), os = __shfl_down(m2, o, 64);
      chan_merge(n, mu, m2, on, om, os);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sn[w] = n; smu[w] = mu; sm2[w] = m2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      n = 0.0; mu = 0.0; m2 = 0.0;
      for (int ww = 0; ww < 16; ++ww) chan_merge(n, mu, m2, sn[ww], smu[ww], sm2[ww]);
      fstats[3 * f] = n; fstats[3 * f + 1] = mu; fstats[3 * f + 2] = m2;
    }
  }
}

template <int MT>
int occupancy_wgs() {
  static int cached = 0;
  if (cached) return cached;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stats_gram_kernel<MT, 2, double>, GramCfg<MT>::NW * 64, 0) != hipSuccess ||
      per_cu < 1)
    per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  const int cus = spr_cached_cus();
  cached = per_cu * (cus > 0 ? cus : 256);
  return cached;
}

template <int MT>
SegPlan make_plan(int64_t n_rows, int64_t row0, int64_t n_points, int32_t n_features) {
  SegPlan p;
  p.row0 = row0; p.n_rows = n_rows; p.n_points = n_points; p.n_features = n_features;
  p.total_wg = occupancy_wgs<MT>();
  p.chunk_rows = GramCfg<MT>::R;
  return p;
}

template <int MT>
size_t workspace_bytes(int32_t n_features) {
  using RT = RowTile<MT, GramCfg<MT>::R, GramShape<MT>::MP, GramCfg<MT>::NW, GramCfg<MT>::LPRMAX>;
  const int64_t max_grid = (int64_t)occupancy_wgs<MT>() + n_features;
  return (size_t)max_grid * GramCfg<MT>::KS * GramShape<MT>::T * 256 * sizeof(double) +
         (size_t)max_grid * RT::ROWS_PER_IT * 3 * sizeof(double);
}

template <int MT, typename TX>
int launch(const TX *X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0, int64_t n_points,
           int32_t n_features, int center, double *rowmean, void *ws, size_t ws_bytes, hipStream_t st,
           double *rowsum = nullptr) {
  SPR_REQUIRE(ws_bytes >= workspace_bytes<MT>(n_features), SPR_E_WORKSPACE,
              "spr_stats_gram_f64: workspace %zu < %zu", ws_bytes, workspace_bytes<MT>(n_features));
  SPR_REQUIRE(!rowsum || center == 2, SPR_E_INVALID, "spr_stats_gram: row sums are an output of centre mode 2 only");
  SegPlan plan = make_plan<MT>(n_rows, row0, n_points, n_features);
  const int grid = seg_total_wgs(plan);
  const int64_t max_grid = (int64_t)occupancy_wgs<MT>() + n_features;
  SPR_REQUIRE(grid >= 1 && grid <= max_grid, SPR_E_INVALID, "spr_stats_gram_f64: bad grid %d", grid);
  double *slab = static_cast<double *>(ws);
  double *stat_part = slab + (size_t)max_grid * GramCfg<MT>::KS * GramShape<MT>::T * 256;
  // two-element pieces: 16-byte aligned rows for f64, 8-byte aligned rows for f32
  const int vec_ok = (m % 2 == 0) && (ldx % 2 == 0) && ((reinterpret_cast<uintptr_t>(X) & (2 * sizeof(TX) - 1)) == 0);
  const int lm = vec_ok ? ((m == 16 * MT) ? 2 : 1) : 0;
#define SG_LAUNCH(LM)                                                                                         \
  hipLaunchKernelGGL((stats_gram_kernel<MT, LM, TX>), dim3(grid), dim3(GramCfg<MT>::NW * 64), 0, st, X, ldx, (int)m, \
                     center, plan, rowmean, stat_part, slab, rowsum)
  // own-means lane: centre mode 1 on packed, 16-byte-aligned rows of exactly 16 MT columns, MFMA-bound widths only
  // (below m = 128 the pass is HBM-bound and the extra statistics launch would cost more than the VALU work it saves)
  using RTL = RowTile<MT, GramCfg<MT>::R, GramShape<MT>::MP, GramCfg<MT>::NW, GramCfg<MT>::LPRMAX, TX>;
  static const bool own_on = [] { const char *e = getenv("SPR_GRAM_OWN"); return !(e && e[0] == '0'); }();
  if (MT >= 8 && lm == 2 && (center == 1 || center == 2) && own_on && (sizeof(TX) * ldx) % 16 == 0 &&
      (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
    if constexpr (MT >= 8) {
      if (center == 1)
        hipLaunchKernelGGL((stats_gram_own_kernel<MT, TX, false, false>), dim3(grid), dim3(GramCfg<MT>::NW * 64), 0, st, X, ldx,
                           plan, rowmean, slab, rowsum);
      else if (!rowsum)   // external means: a column slice of a wider matrix, centred with the means of the full rows
        hipLaunchKernelGGL((stats_gram_own_kernel<MT, TX, true, false>), dim3(grid), dim3(GramCfg<MT>::NW * 64), 0, st, X, ldx,
                           plan, rowmean, slab, rowsum);
      else                // external per-row constants that are NOT the means: the raw row sums come out as well
        hipLaunchKernelGGL((stats_gram_own_kernel<MT, TX, true, true>), dim3(grid), dim3(GramCfg<MT>::NW * 64), 0, st, X, ldx,
                           plan, rowmean, slab, rowsum);
      SPR_LAUNCH_CHECK();
      // statistics of the row means (own or external: the finalize call merges whatever the slots hold)
      hipLaunchKernelGGL(gram_rowmean_stats_kernel, dim3(grid), dim3(RS_THREADS), 0, st, rowmean, plan, (int)RTL::ROWS_PER_IT,
                         stat_part);
      SPR_LAUNCH_CHECK();
      return SPR_OK;
    }
  }
  if (lm == 2) SG_LAUNCH(2);
  else if (lm == 1) SG_LAUNCH(1);
  else SG_LAUNCH(0);
#undef SG_LAUNCH
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

template <int MT>
int launch_finalize(int64_t n_rows, int32_t m, int64_t row0, int64_t n_points, int32_t n_features,
                    double *fstats, double *gram, int ldg, int origin, const void *ws, size_t ws_bytes,
                    hipStream_t st) {
  using RT = RowTile<MT, GramCfg<MT>::R, GramShape<MT>::MP, GramCfg<MT>::NW, GramCfg<MT>::LPRMAX>;
  SPR_REQUIRE(ws_bytes >= workspace_bytes<MT>(n_features), SPR_E_WORKSPACE,
              "spr_stats_gram_finalize_f64: workspace %zu < %zu", ws_bytes, workspace_bytes<MT>(n_features));
  SegPlan plan = make_plan<MT>(n_rows, row0, n_points, n_features);
  const int64_t max_grid = (int64_t)occupancy_wgs<MT>() + n_features;
  const double *slab = static_cast<const double *>(ws);
  const double *stat_part = slab + (size_t)max_grid * GramCfg<MT>::KS * GramShape<MT>::T * 256;
  hipLaunchKernelGGL(gram_finalize_kernel<MT>, dim3(GramShape<MT>::T, n_features), dim3(1024), 0, st, slab,
                     stat_part, (int)m, plan, (int)RT::ROWS_PER_IT, gram, fstats, ldg, origin);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

int check_args(const char *who, const void *X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
               int64_t n_points, int32_t n_features) {
  SPR_REQUIRE(X != nullptr, SPR_E_INVALID, "%s: X is NULL", who);
  SPR_REQUIRE(n_rows > 0 && m > 0 && ldx >= m, SPR_E_INVALID, "%s: bad shape n_rows=%lld m=%d ldx=%lld", who,
              (long long)n_rows, m, (long long)ldx);
  SPR_REQUIRE(n_points > 0 && n_features > 0 && row0 >= 0, SPR_E_INVALID, "%s: bad feature layout", who);
  SPR_REQUIRE(row0 + n_rows <= n_points * (int64_t)n_features, SPR_E_INVALID,
              "%s: rows [%lld,%lld) exceed n_points*n_features=%lld", who, (long long)row0,
              (long long)(row0 + n_rows), (long long)(n_points * (int64_t)n_features));
  SPR_REQUIRE(m <= SPR_MAX_M, SPR_E_UNSUPPORTED, "%s: m=%d > %d not built", who, m, SPR_MAX_M);
  return SPR_OK;
}

}  // namespace

#define SPR_DISPATCH_MT(mt, CALL)            \
  switch (mt) {                              \
    case 1: CALL(1); break;                  \
    case 2: CALL(2); break;                  \
    case 3: CALL(3); break;                  \
    case 4: CALL(4); break;                  \
    case 6: CALL(6); break;                  \
    case 8: CALL(8); break;                  \
    case 12: CALL(12); break;                \
    case 16: CALL(16); break;                \
    default: break;                          \
  }

extern "C" size_t spr_stats_gram_workspace(int32_t m, int32_t n_features) {
  if (m <= 0 || m > SPR_MAX_M || n_features <= 0) return 0;
  size_t out = 0;
#define WS_CALL(MTV) out = workspace_bytes<MTV>(n_features)
  SPR_DISPATCH_MT(spr_round_mt(m), WS_CALL)
#undef WS_CALL
  return out;
}

extern "C" int spr_stats_gram_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                  int64_t n_points, int32_t n_features, int32_t center, double *d_rowmean,
                                  void *d_workspace, size_t workspace_bytes_, void *stream) {
  int rc = check_args("spr_stats_gram_f64", d_X, n_rows, m, ldx, row0, n_points, n_features);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_rowmean && d_workspace, SPR_E_INVALID, "spr_stats_gram_f64: NULL output/workspace");
  rc = SPR_E_UNSUPPORTED;
#define RUN_CALL(MTV)                                                                                \
  rc = launch<MTV>(d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_rowmean, d_workspace, \
                   workspace_bytes_, static_cast<hipStream_t>(stream))
  SPR_DISPATCH_MT(spr_round_mt(m), RUN_CALL)
#undef RUN_CALL
  return rc;
}

extern "C" int spr_stats_gram_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                  int64_t n_points, int32_t n_features, int32_t center, double *d_rowmean,
                                  void *d_workspace, size_t workspace_bytes_, void *stream) {
  int rc = check_args("spr_stats_gram_x32", d_X, n_rows, m, ldx, row0, n_points, n_features);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_rowmean && d_workspace, SPR_E_INVALID, "spr_stats_gram_x32: NULL output/workspace");
  rc = SPR_E_UNSUPPORTED;
#define RUN_CALL(MTV)                                                                                \
  rc = launch<MTV>(d_X, n_rows, m, ldx, row0, n_points, n_features, center, d_rowmean, d_workspace, \
                   workspace_bytes_, static_cast<hipStream_t>(stream))
  SPR_DISPATCH_MT(spr_round_mt(m), RUN_CALL)
#undef RUN_CALL
  return rc;
}

// The same pass for a column slice whose rows are shifted by given per-row constants that are NOT their means (d_shift,
// centre mode 2) -- the raw sums of the slice's rows come out in d_rowsum, so that the caller can form the true means
// (spr_gram_shift_finish_f64, gram_wide.hip).
extern "C" int spr_stats_gram_shifted_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                          int64_t n_points, int32_t n_features, const double *d_shift,
                                          double *d_rowsum, void *d_workspace, size_t workspace_bytes_, void *stream) {
  int rc = check_args("spr_stats_gram_shifted_f64", d_X, n_rows, m, ldx, row0, n_points, n_features);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_shift && d_rowsum && d_workspace, SPR_E_INVALID, "spr_stats_gram_shifted_f64: NULL pointer");
  rc = SPR_E_UNSUPPORTED;
#define RUN_CALL(MTV)                                                                                                    \
  rc = launch<MTV>(d_X, n_rows, m, ldx, row0, n_points, n_features, 2, const_cast<double *>(d_shift), d_workspace,       \
                   workspace_bytes_, static_cast<hipStream_t>(stream), d_rowsum)
  SPR_DISPATCH_MT(spr_round_mt(m), RUN_CALL)
#undef RUN_CALL
  return rc;
}

extern "C" int spr_stats_gram_shifted_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                          int64_t n_points, int32_t n_features, const double *d_shift,
                                          double *d_rowsum, void *d_workspace, size_t workspace_bytes_, void *stream) {
  int rc = check_args("spr_stats_gram_shifted_x32", d_X, n_rows, m, ldx, row0, n_points, n_features);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_shift && d_rowsum && d_workspace, SPR_E_INVALID, "spr_stats_gram_shifted_x32: NULL pointer");
  rc = SPR_E_UNSUPPORTED;
#define RUN_CALL(MTV)                                                                                                    \
  rc = launch<MTV>(d_X, n_rows, m, ldx, row0, n_points, n_features, 2, const_cast<double *>(d_shift), d_workspace,       \
                   workspace_bytes_, static_cast<hipStream_t>(stream), d_rowsum)
  SPR_DISPATCH_MT(spr_round_mt(m), RUN_CALL)
#undef RUN_CALL
  return rc;
}

extern "C" int spr_stats_gram_finalize_f64(int64_t n_rows, int32_t m, int64_t row0, int64_t n_points,
                                           int32_t n_features, const void *d_workspace, size_t workspace_bytes_,
                                           double *d_fstats, double *d_gram, int32_t ldg, int32_t origin,
                                           void *stream) {
  int rc = check_args("spr_stats_gram_finalize_f64", d_workspace, n_rows, m, m, row0, n_points, n_features);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_fstats && d_gram, SPR_E_INVALID, "spr_stats_gram_finalize_f64: NULL output");
  SPR_REQUIRE(origin >= 0 && ldg >= origin + m, SPR_E_INVALID, "spr_stats_gram_finalize_f64: block (%d,%d)+%d outside ldg=%d",
              origin, origin, m, ldg);
  rc = SPR_E_UNSUPPORTED;
#define FIN_CALL(MTV)                                                                                 \
  rc = launch_finalize<MTV>(n_rows, m, row0, n_points, n_features, d_fstats, d_gram, ldg, origin, d_workspace, \
                            workspace_bytes_, static_cast<hipStream_t>(stream))
  SPR_DISPATCH_MT(spr_round_mt(m), FIN_CALL)
#undef FIN_CALL
  return rc;
}
