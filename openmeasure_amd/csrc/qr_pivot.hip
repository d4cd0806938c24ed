// K6: QR column pivoting of Ur^T -- greedy max-residual-norm row selection, candidate-set form.
//
// dgeqp3 on the r x n matrix Ur^T (sparse_sensing.py:739) picks, at step j, the column with
// the largest residual norm after projecting out the j columns already chosen; only the
// pivot ORDER is used by the reference (:740-743).  Per step the winning row u_p gives the
// residual direction q_j = (I - Q Q^T) u_p / |..| in the r-dimensional coefficient space
// (classical Gram-Schmidt applied twice), and every row's squared residual norm drops by
// (u_i . q_j)^2.  Ties go to the lowest global row index, as LAPACK's idamax does.
//
// Sweeping all of Ur once per step costs r passes over n*r*8 bytes.  Residual norms only
// decrease, so a stale norm is an upper bound: a full sweep ("refresh") leaves exact norms
// and, per sweep block, its QR_TOPT largest rows; their union is the candidate set (a few
// thousand rows, L2-resident) and tau = the largest norm any NON-candidate can have.  Steps
// are then taken on the candidates alone -- exact arithmetic on exact copies of their rows --
// and a step is certified only while its winner's residual is strictly above tau (the first
// step after a refresh is always exact: every block's maximum is a candidate).  After at most
// QR_BATCH steps, or at the first uncertified step, the accepted directions are applied to
// all rows in ONE multi-direction sweep and a new candidate set is drawn.  Same pivots as the
// step-per-sweep algorithm, r/QR_BATCH-ish passes over Ur instead of r.
//
// Candidate record (one per rank, all-gathered between steps when sharded), r+3 doubles:
//   [0] best residual norm^2   [1] its global row (as double, exact below 2^53)
//   [2] runner-up norm^2 among this rank's candidates   [3..3+r) the row of Ur
#include <stdlib.h>

#include <type_traits>

#include "rowtile.hpp"

namespace {

constexpr int QR_THREADS = 256;
#ifndef QR_EPOCH_DEEP
#define QR_EPOCH_DEEP 0   // 1: epoch sweeps keep a third register set (two panels ahead) where it fits; 0: two sets, more waves
#endif
#ifndef QR_ABLATE
#define QR_ABLATE 0   // diagnostics (tools/ablate.sh, wrong results): 1 = sweeps keep no candidate lists, 2 = nor touch the norm vector
#endif
#ifndef QR_TOPT_N
#define QR_TOPT_N 16   // 16 instead of 8: 5 sweeps instead of 6 at config 3 (55.6 -> 49.7 ms, tools/topt_ab.sh); a batch of
                       // 32 directions on top of that certified [18,16,14,10,6]: still 5 sweeps, each slower
#endif
constexpr int QR_TOPT = QR_TOPT_N;  // rows kept per sweep block
constexpr int QR_MAX_BLOCKS = 1024; // sweep grid cap -> at most 16384 candidates
constexpr int QR_BATCH = 16;        // directions applied per refresh sweep (one MFMA tile of columns)

struct Best {
  double v1; int64_t i1; double v2;
  __device__ inline void init() { v1 = -2.0; i1 = INT64_MAX; v2 = -2.0; }
  __device__ inline void push(double v, int64_t i) {
    if (v > v1 || (v == v1 && i < i1)) { v2 = v1; v1 = v; i1 = i; }
    else if (v > v2) v2 = v;
  }
  __device__ inline void merge(double ov1, int64_t oi1, double ov2) {
    if (ov1 > v1 || (ov1 == v1 && oi1 < i1)) {
      v2 = (v1 > ov2) ? v1 : ov2; v1 = ov1; i1 = oi1;
    } else {
      if (ov1 > v2) v2 = ov1;
    }
  }
};

// candidate step: dot product of a row (read by LPR lanes, two doubles each) with the new direction held in
// registers, then v <- max(v - d^2, 0)
template <int LPR>
__device__ inline double downdate(double v, f64x2 u, double q0, double q1) {
  double d = u.x * q0 + u.y * q1;
  d = group_sum_t<LPR>(d);
  v -= d * d;
  return v < 0.0 ? 0.0 : v;
}

// TU: storage type of the basis (f64, or f32 widened on load); every norm and dot product is f64
template <typename TU>
__device__ inline f64x2 load_row_piece(const TU *__restrict__ rp, int k0, int r, bool vec_ok, bool valid) {
  f64x2 t = {0.0, 0.0};
  if (valid) {
    if (vec_ok) {
      if (k0 < r) t = widen(*reinterpret_cast<const typename PieceOf<TU>::type *>(rp + k0));
    } else {
      if (k0 < r) t.x = (double)rp[k0];
      if (k0 + 1 < r) t.y = (double)rp[k0 + 1];
    }
  }
  return t;
}

// Per-lane sorted list of the QR_TOPT largest (value, global row) pairs seen, and its block-level merge.
// TI: index type of the per-lane entries -- the global row as long long, or (epoch sweeps: fewer than 2^31 local rows) the
// LOCAL row as int with the shard's first row added when the lists are staged: 16 VGPRs less, a third wave per SIMD.
template <typename TI>
struct TopListT {
  double tv[QR_TOPT];
  TI ti[QR_TOPT];
  long long base = 0;            // added to every index at the block merge
  __device__ inline void init() {
#pragma unroll
    for (int k = 0; k < QR_TOPT; ++k) { tv[k] = -2.0; ti[k] = -1; }
  }
  // rows reach a lane in increasing index order, so "strictly greater" keeps the lowest index among equals
  __device__ inline void insert(double v, TI gi, bool mine) {
    const bool ins = mine && (v > tv[QR_TOPT - 1]);
    if (__any(ins)) {
#pragma unroll
      for (int k = QR_TOPT - 1; k >= 0; --k) {
        const bool here = ins && (v > tv[k]);
        const bool above = (k > 0) ? (v > tv[k > 0 ? k - 1 : 0]) : false;
        const double nv = above ? tv[k > 0 ? k - 1 : 0] : v;
        const TI ni = above ? ti[k > 0 ? k - 1 : 0] : gi;
        tv[k] = here ? nv : tv[k];
        ti[k] = here ? ni : ti[k];
      }
    }
  }
  // Block-level merge, two levels: every wave extracts the QR_TOPT best of its own 64 x QR_TOPT entries (QR_TOPT
  // rounds of arg-max, all waves in parallel), then wave 0 merges the waves' short lists; result sorted to
  // out[QR_TOPT][2].  Order: value descending, lowest global row first among equals.
  __device__ static inline void argmax_round(const double *sval, const long long *sidx, int first, int count, int lane,
                                             double &bv, long long &bi, int &bp) {
    bv = -3.0; bi = INT64_MAX; bp = -1;
    for (int e = first + lane; e < first + count; e += 64) {
      const double v = sval[e]; const long long i = sidx[e];
      if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; bp = e; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(bv, o, 64);
      const long long oi = __shfl_xor(bi, o, 64);
      const int op = __shfl_xor(bp, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; bp = op; }
    }
  }
  // Only `owner` lanes ever inserted rows (SPW of them per wave, numbered slot = 0..SPW-1): the staging area is
  // QR_THREADS/64 x SPW x QR_TOPT entries, not one list per thread.
  template <int SPW>
  __device__ inline void block_merge(double *sval, long long *sidx, double *out, bool owner, int slot) {
    constexpr int NWV = QR_THREADS / 64;
    __shared__ double wv[NWV * QR_TOPT];
    __shared__ long long wi[NWV * QR_TOPT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (owner) {
#pragma unroll
      for (int k = 0; k < QR_TOPT; ++k) {
        sval[(wave * SPW + slot) * QR_TOPT + k] = tv[k];
        sidx[(wave * SPW + slot) * QR_TOPT + k] = ti[k] >= 0 ? (long long)ti[k] + base : -1;
      }
    }
    __syncthreads();
    for (int round = 0; round < QR_TOPT; ++round) {       // level 1: this wave's own SPW x QR_TOPT entries
      double bv; long long bi; int bp;
      argmax_round(sval, sidx, wave * SPW * QR_TOPT, SPW * QR_TOPT, lane, bv, bi, bp);
      if (lane == 0) {
        wv[wave * QR_TOPT + round] = bv;
        wi[wave * QR_TOPT + round] = bi;
        if (bp >= 0) sval[bp] = -3.0;
      }
      __builtin_amdgcn_wave_barrier();   // one wave per region: LDS accesses of a wave are issued and serviced in order
    }
    __syncthreads();
    if (wave == 0) {                                      // level 2: NWV short lists
      for (int round = 0; round < QR_TOPT; ++round) {
        double bv; long long bi; int bp;
        argmax_round(wv, wi, 0, NWV * QR_TOPT, lane, bv, bi, bp);
        if (lane == 0) {
          out[2 * round] = bv > -2.5 ? bv : -2.0;
          out[2 * round + 1] = (double)(bv > -2.5 ? bi : -1);
          if (bp >= 0) wv[bp] = -3.0;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
};
using TopList = TopListT<long long>;

// Who writes a row's norm at the end of a sweep block (all three kernel forms).  The 16 x 16 MFMA result of a wave's
// 16-row block leaves lane (g = lane >> 4, c = lane & 15) with the entries (row g + 4 q, column c), q = 0..3.
//   refresh: after the 16-lane sums every lane of group g holds the squared projections of rows g + 4 q -- lane (g, c < 4)
//            takes row g + 4 c (its register q = c);
//   init:    the diagonal entry of row c sits in lane (g = c & 3, c), register q = c >> 2 -- that lane takes row c.
// Either way the 16 owners of a block read (refresh) and write 16 CONSECUTIVE doubles of the norm vector with one load and
// one store instruction per wave.  The first form of these kernels let lane (g, 0) write its four rows with four stores of
// four scattered 8-byte pieces each: at config 3 the 0.72 GB norm vector then cost 1.8 ms of a 9.4 ms sweep
// (tools/ablate.sh -DQR_ABLATE=2) -- partial-line writes, four instructions per 128-byte line.
template <bool INIT>
struct RowOwner {
  int off, q;        // row of the block this lane reads / writes, register holding its value
  bool own;
  __device__ inline RowOwner(int lane) {
    const int g = lane >> 4, c = lane & 15;
    if (INIT) { off = c; q = c >> 2; own = (c & 3) == g; }
    else { off = g + 4 * (c & 3); q = c & 3; own = c < 4; }     // lanes c >= 4 mirror an owner's address: loads only
  }
  __device__ inline int slot(int lane) const { return INIT ? ((lane >> 4) * 4 + ((lane & 15) >> 2)) : ((lane >> 4) * 4 + (lane & 3)); }
  __device__ inline double pick(double v0, double v1, double v2, double v3) const {
    return q == 0 ? v0 : q == 1 ? v1 : q == 2 ? v2 : v3;
  }
};
constexpr int QR_SPW = 16;   // owner lanes (and candidate lists) per wave

// Refresh sweep, MFMA form: nrm <- nrm - sum_t (u_i . q_t)^2 for up to 16 directions at once.
// Panels of 64 rows of Ur go to LDS raw (rowtile.hpp staging, double-buffered, loads of the panel
// after next issued behind the stores); wave w multiplies its 16-row block with Q^T (directions as
// the 16 MFMA columns, fragments in registers): r/4 v_mfma_f64_16x16x4_f64 per block, far below the
// HBM time of the panel, where the VALU form spends 8 x (fma + 5-step butterfly) per row pair.
// The squared products are summed over the 16 direction lanes with a DPP butterfly; lanes 0/16/32/48
// of a wave own rows a, a+4, a+8, a+12 of the block (increasing order, as TopList needs).
// INIT (first sweep of a placement): the same pass with the block itself as the second operand -- the diagonal of
// U_blk U_blk^T are the squared row norms -- so the initial norms stream at the same rate as the refreshes.
template <int MTR, int VEC, typename TU, bool INIT>
__global__ __launch_bounds__(QR_THREADS) void qr_refresh_mfma_kernel(
    const TU *__restrict__ Ur, int64_t n_rows, int r, int64_t ldu, int64_t row0,
    const double *__restrict__ Q, int nq, double *__restrict__ nrm, double *__restrict__ tops) {
  constexpr int NW = QR_THREADS / 64, R = 64;
  constexpr int MPAD = 16 * MTR, MP = MPAD + 2, KSTEPS = MPAD / 4;
  using RT = RowTile<MTR, R, MP, NW, 16, TU>;
  constexpr int SPW = QR_SPW;                              // lanes of a wave that own rows (and so a top list)
  constexpr int PANELS = 2 * R * MP, MERGE = 2 * (QR_THREADS / 64) * SPW * QR_TOPT;
  __shared__ double smem[PANELS > MERGE ? PANELS : MERGE];    // panels during the sweep, merge lists afterwards
  double *const lds0 = smem, *const lds1 = smem + R * MP;
  double *const sval = smem;
  long long *const sidx = reinterpret_cast<long long *>(smem + (QR_THREADS / 64) * SPW * QR_TOPT);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  double bfrag[KSTEPS];          // B[k][j] = Q[j][k]: direction j = lane & 15, k = 4 ks + (lane >> 4)
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int k = 4 * ks + (lane >> 4), j = lane & 15;
    bfrag[ks] = (!INIT && j < nq && k < r) ? Q[(int64_t)j * r + k] : 0.0;
  }
  TopList top;
  top.init();
  const RowOwner<INIT> who(lane);
  RT tile;
  const int64_t npanels = (n_rows + R - 1) / R;
  int64_t c = blockIdx.x;
  if (c < npanels) {
    tile.template load<VEC>(Ur, ldu, r, c * R, n_rows, wave, lane);
    tile.raw_store(lds0, r, c * R, n_rows, wave, lane);
    int64_t cn = c + gridDim.x;
    int64_t nrow0 = (cn < npanels) ? cn * R : n_rows;
    tile.template load<VEC>(Ur, ldu, r, nrow0, n_rows, wave, lane);
    int buf = 0;
    const int afrag = (lane & 15) * MP + (lane >> 4);
    // the old norm of this lane's row, two panels ahead like the rows themselves: a load issued in the iteration that
    // consumes it leaves one HBM latency exposed per panel (a panel's 16 dependent MFMAs are shorter than that)
    auto old_of = [&](int64_t row0p) {
      const int64_t rr = row0p + wave * 16 + who.off;
      return (INIT || QR_ABLATE >= 2) ? 0.0 : nrm[rr < n_rows ? rr : n_rows - 1];
    };
    double old = old_of(c * R), old_n = old_of(nrow0);
    while (c < npanels) {
      const double *cur = buf ? lds1 : lds0;
      double *nxt = buf ? lds0 : lds1;
      const int64_t c2 = cn + gridDim.x;
      const int64_t n2row0 = (c2 < npanels) ? c2 * R : n_rows;
      __syncthreads();
      const int64_t orow = c * R + wave * 16 + who.off;          // the row this lane reads / writes (RowOwner)
      const bool mine = who.own && orow < n_rows;
      const double old_n2 = old_of(n2row0);
      const double *p = cur + wave * 16 * MP + afrag;
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const double a = p[4 * ks];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, INIT ? a : bfrag[ks], acc, 0, 0, 0);
        if (ks == 0) {
#pragma unroll
          for (int it = 0; it < RT::IT; ++it) {
            tile.raw_store_pass(it, nxt, r, nrow0, n_rows, wave, lane);
            tile.template load_pass<VEC>(it, Ur, ldu, r, n2row0, n_rows, wave, lane);
          }
        }
      }
      double v;
      if (INIT) {
        v = who.pick(acc.x, acc.y, acc.z, acc.w);            // D[i][j] = u_i . u_j of the block: its diagonal
      } else {
        const double d2 = who.pick(group_sum_t<16>(acc.x * acc.x), group_sum_t<16>(acc.y * acc.y),
                                   group_sum_t<16>(acc.z * acc.z), group_sum_t<16>(acc.w * acc.w));
        v = old - d2;
        v = v < 0.0 ? 0.0 : v;
        v = old < 0.0 ? -1.0 : v;
      }
      if (QR_ABLATE >= 2) asm volatile("" ::"v"(v));
      if (QR_ABLATE < 2 && mine) nrm[orow] = v;
      if (QR_ABLATE < 1) top.insert(v, row0 + orow, mine);
      old = old_n;
      old_n = old_n2;
      buf ^= 1;
      c = cn;
      cn = c2;
      nrow0 = n2row0;
    }
  }
  __syncthreads();   // the panels are dead from here on: their LDS is reused for the merge
  top.template block_merge<SPW>(sval, sidx, tops + (int64_t)blockIdx.x * QR_TOPT * 2, who.own, who.slot(lane));
}

// Refresh / init sweep, register-direct form (r a multiple of 16, rows 16-byte aligned): the same arithmetic, the same
// row -> wave -> lane assignment and therefore the same candidate lists as qr_refresh_mfma_kernel, but the rows of Ur
// go HBM -> registers directly in the MFMA A layout instead of through an LDS panel behind a barrier.  Lane
// (i = l & 15, kk = l >> 4) owns row i of its wave's 16-row block and loads the four consecutive elements
// [16 g + 4 kk, +4) of it for g = 0 .. r/16 - 1 (one 16-byte piece of an f32 row, two of an f64 row); MFMA step 4 g + t
// contracts over column 16 g + 4 kk + t, and the direction fragments are permuted the same way once per kernel.
// Waves are independent: no LDS traffic, no barrier until the final list merge; the next block of the wave is
// requested before the current one is multiplied.  An f32-stored basis moves half the bytes per block with the
// same (small) fixed cost per block, which is what kept the LDS form at 2.9-3.6 TB/s on it.
template <int NG, typename TU, bool INIT>
__global__ __launch_bounds__(QR_THREADS) void qr_refresh_direct_kernel(
    const TU *__restrict__ Ur, int64_t n_rows, int r, int64_t ldu, int64_t row0,
    const double *__restrict__ Q, int nq, double *__restrict__ nrm, double *__restrict__ tops) {
  constexpr int R = 64;
  constexpr int SPW = QR_SPW;
  __shared__ double smem[2 * (QR_THREADS / 64) * SPW * QR_TOPT];
  double *const sval = smem;
  long long *const sidx = reinterpret_cast<long long *>(smem + (QR_THREADS / 64) * SPW * QR_TOPT);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kk = lane >> 4;
  using P4 = typename std::conditional<std::is_same<TU, float>::value, float4, double4>::type;

  double bfrag[4 * NG];          // B[k][j] = Q[j][k] at the permuted k of step 4 g + t
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      bfrag[4 * g + t] = (!INIT && li < nq) ? Q[(int64_t)li * r + 16 * g + 4 * kk + t] : 0.0;
  TopList top;
  top.init();
  const RowOwner<INIT> who(lane);
  const int64_t npanels = (n_rows + R - 1) / R;
  auto load_block = [&](int64_t c, P4 (&dst)[NG]) {
    int64_t row = c * R + wave * 16 + li;
    row = row < n_rows ? row : n_rows - 1;                  // rows past the end re-read the last row; never stored
    const TU *rp = Ur + row * ldu + 4 * kk;
#pragma unroll
    for (int g = 0; g < NG; ++g) dst[g] = *reinterpret_cast<const P4 *>(rp + 16 * g);
  };
  auto old_of = [&](int64_t cc) {                          // the old norm of this lane's row of block cc
    const int64_t rr = cc * R + wave * 16 + who.off;
    return INIT ? 0.0 : nrm[rr < n_rows ? rr : n_rows - 1];
  };
  int64_t c = blockIdx.x;
  P4 cur[NG], nxt[NG];
  double old = 0.0;
  if (c < npanels) { load_block(c, cur); old = old_of(c); }
  while (c < npanels) {
    const int64_t cn = c + gridDim.x;
    load_block(cn < npanels ? cn : c, nxt);
    const double old_n = old_of(cn < npanels ? cn : c);    // one block ahead, like the rows
    const int64_t orow = c * R + wave * 16 + who.off;      // the row this lane reads / writes (RowOwner)
    const bool mine = who.own && orow < n_rows;
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const double a4[4] = {(double)cur[g].x, (double)cur[g].y, (double)cur[g].z, (double)cur[g].w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[t], INIT ? a4[t] : bfrag[4 * g + t], acc, 0, 0, 0);
    }
    double v;
    if (INIT) {
      v = who.pick(acc.x, acc.y, acc.z, acc.w);
    } else {
      const double d2 = who.pick(group_sum_t<16>(acc.x * acc.x), group_sum_t<16>(acc.y * acc.y),
                                 group_sum_t<16>(acc.z * acc.z), group_sum_t<16>(acc.w * acc.w));
      v = old - d2;
      v = v < 0.0 ? 0.0 : v;
      v = old < 0.0 ? -1.0 : v;
    }
    if (mine) nrm[orow] = v;
    top.insert(v, row0 + orow, mine);
#pragma unroll
    for (int g = 0; g < NG; ++g) cur[g] = nxt[g];
    old = old_n;
    c = cn;
  }
  top.template block_merge<SPW>(sval, sidx, tops + (int64_t)blockIdx.x * QR_TOPT * 2, who.own, who.slot(lane));
}

// Refresh / init sweep for a basis wider than 128 columns (r <= SPR_MAX_R_WIDE; the reference pivots Ur^T for any r <= m,
// :739): the register-direct form above with a RUN-TIME loop over the 16-column groups of a row -- eight groups (128
// columns) of the wave's 16-row block are in registers at a time, the next eight are requested before these are
// multiplied -- and the <= 16 directions in LDS instead of registers: image Ql[((g 4 + t) 4 + kk) 16 + li] =
// Q[li][16 g + 4 kk + t], so the 64 lanes of a step read 64 consecutive doubles (conflict-free ds_read_b64).  Same row ->
// wave -> lane assignment, arithmetic order per row and candidate lists as the other two forms.  VEC 0: any r / alignment
// (scalar loads, columns >= r contribute zeros).
template <int NGMAX, int VEC, typename TU, bool INIT>
__global__ __launch_bounds__(QR_THREADS) void qr_refresh_wide_kernel(
    const TU *__restrict__ Ur, int64_t n_rows, int r, int64_t ldu, int64_t row0,
    const double *__restrict__ Q, int nq, double *__restrict__ nrm, double *__restrict__ tops) {
  constexpr int R = 64, SG = 8;                             // rows per workgroup step; groups per register batch
  constexpr int SPW = QR_SPW;
  __shared__ double smem[2 * (QR_THREADS / 64) * SPW * QR_TOPT];
  __shared__ double Ql[INIT ? 16 : NGMAX * 256];
  double *const sval = smem;
  long long *const sidx = reinterpret_cast<long long *>(smem + (QR_THREADS / 64) * SPW * QR_TOPT);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kk = lane >> 4;
  const int ng = (r + 15) / 16;
  using P4 = typename std::conditional<std::is_same<TU, float>::value, float4, double4>::type;

  if (!INIT) {
    for (int e = threadIdx.x; e < ng * 256; e += QR_THREADS) {
      const int j = e & 15, k4 = (e >> 4) & 3, t = (e >> 6) & 3, g = e >> 8;
      const int k = 16 * g + 4 * k4 + t;
      Ql[e] = (j < nq && k < r) ? Q[(int64_t)j * r + k] : 0.0;
    }
    __syncthreads();
  }
  TopList top;
  top.init();
  const RowOwner<INIT> who(lane);
  const int64_t npanels = (n_rows + R - 1) / R;
  auto load_batch = [&](const TU *rp, int g0, P4 (&dst)[SG]) {
#pragma unroll
    for (int u = 0; u < SG; ++u) {
      const int g = g0 + u < ng ? g0 + u : ng - 1;          // past the last group: a harmless re-read, never multiplied
      const int c0 = 16 * g + 4 * kk;
      if (VEC) {
        dst[u] = *reinterpret_cast<const P4 *>(rp + c0);
      } else {
        dst[u].x = c0 < r ? rp[c0] : (TU)0;         dst[u].y = c0 + 1 < r ? rp[c0 + 1] : (TU)0;
        dst[u].z = c0 + 2 < r ? rp[c0 + 2] : (TU)0; dst[u].w = c0 + 3 < r ? rp[c0 + 3] : (TU)0;
      }
    }
  };
  for (int64_t c = blockIdx.x; c < npanels; c += gridDim.x) {
    int64_t row = c * R + wave * 16 + li;
    row = row < n_rows ? row : n_rows - 1;                  // rows past the end re-read the last row; never stored
    const TU *rp = Ur + row * ldu;
    const int64_t orow = c * R + wave * 16 + who.off;       // the row this lane reads / writes (RowOwner)
    const bool mine = who.own && orow < n_rows;
    const double old = INIT ? 0.0 : nrm[orow < n_rows ? orow : n_rows - 1];
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    P4 cur[SG], nxt[SG];
    load_batch(rp, 0, cur);
    for (int g0 = 0; g0 < ng; g0 += SG) {
      load_batch(rp, g0 + SG, nxt);
#pragma unroll
      for (int u = 0; u < SG; ++u) {
        if (g0 + u < ng) {                                  // wave-uniform
          const double a4[4] = {(double)cur[u].x, (double)cur[u].y, (double)cur[u].z, (double)cur[u].w};
          const double *ql = Ql + (int64_t)(g0 + u) * 256 + kk * 16 + li;
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[t], INIT ? a4[t] : ql[64 * t], acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < SG; ++u) cur[u] = nxt[u];
    }
    double v;
    if (INIT) {
      v = who.pick(acc.x, acc.y, acc.z, acc.w);
    } else {
      const double d2 = who.pick(group_sum_t<16>(acc.x * acc.x), group_sum_t<16>(acc.y * acc.y),
                                 group_sum_t<16>(acc.z * acc.z), group_sum_t<16>(acc.w * acc.w));
      v = old - d2;
      v = v < 0.0 ? 0.0 : v;
      v = old < 0.0 ? -1.0 : v;
    }
    if (mine) nrm[orow] = v;
    top.insert(v, row0 + orow, mine);
  }
  top.template block_merge<SPW>(sval, sidx, tops + (int64_t)blockIdx.x * QR_TOPT * 2, who.own, who.slot(lane));
}

// First "sweep" of a placement whose squared row norms already exist (written by the projection that stored the basis,
// spr_project_norms_* / spr_project_stream_norms_*): 8 bytes per row are read instead of the whole basis.  nrm0 is copied
// to the working vector nrm (the steps down-date it) and every workgroup leaves the QR_TOPT largest rows of ITS panels --
// the same panel -> workgroup assignment as the sweep kernels (panel c belongs to workgroup c mod grid), so candidate
// set and tau are those an init sweep over the same values would have drawn.  Wave w takes every fourth panel of the
// workgroup, four panels per wave in flight; a lane meets its rows in increasing order (what TopList's tie rule needs).
__global__ __launch_bounds__(QR_THREADS) void qr_tops_from_norms_kernel(const double *__restrict__ nrm0, int64_t n_rows,
                                                                        int64_t row0, double *__restrict__ nrm,
                                                                        double *__restrict__ tops) {
  constexpr int R = 64, NWV = QR_THREADS / 64, SPW = 64;
  __shared__ double smem[2 * NWV * SPW * QR_TOPT];
  double *const sval = smem;
  long long *const sidx = reinterpret_cast<long long *>(smem + NWV * SPW * QR_TOPT);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  TopList top;
  top.init();
  const int64_t npanels = (n_rows + R - 1) / R;
  const int64_t stride = (int64_t)gridDim.x * NWV;
  int64_t c = blockIdx.x + (int64_t)wave * gridDim.x;
  for (; (c + 3 * stride + 1) * R <= n_rows; c += 4 * stride) {   // four panels, the furthest one complete
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = nrm0[(c + u * stride) * R + lane];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t row = (c + u * stride) * R + lane;
      nrm[row] = v[u];
      top.insert(v[u], row0 + row, true);
    }
  }
  for (; c < npanels; c += stride) {
    const int64_t row = c * R + lane;
    const bool mine = row < n_rows;
    const double v = nrm0[mine ? row : n_rows - 1];
    if (mine) nrm[row] = v;
    top.insert(v, row0 + row, mine);
  }
  top.template block_merge<SPW>(sval, sidx, tops + (int64_t)blockIdx.x * QR_TOPT * 2, true, lane);
}

// grid = sweep blocks: copy each block's top rows into the compact candidate arrays
template <typename TU>
__global__ __launch_bounds__(QR_THREADS) void qr_gather_kernel(
    const double *__restrict__ tops, const TU *__restrict__ Ur, int r, int64_t ldu, int64_t row0,
    int64_t n_rows, int ldc, int64_t *__restrict__ cand_idx, double *__restrict__ cand_res,
    double *__restrict__ cand_U) {
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < QR_TOPT * ldc; e += QR_THREADS) {
    const int k = e / ldc, c = e - k * ldc;
    const double v = tops[((int64_t)b * QR_TOPT + k) * 2];
    const int64_t gi = (int64_t)tops[((int64_t)b * QR_TOPT + k) * 2 + 1];
    const int64_t li = gi - row0;
    const bool ok = v >= 0.0 && li >= 0 && li < n_rows;     // -1 (already chosen) and -2 (empty) stay out
    const int64_t slot = (int64_t)b * QR_TOPT + k;
    cand_U[slot * ldc + c] = (ok && c < r) ? (double)Ur[li * ldu + c] : 0.0;
    if (c == 0) { cand_idx[slot] = ok ? gi : -1; cand_res[slot] = ok ? v : -2.0; }
  }
}

// one workgroup: tau = largest QR_TOPT-th value over the blocks; best / runner-up candidate -> record
__global__ __launch_bounds__(1024) void qr_cand_best_kernel(
    const double *__restrict__ tops, int n_blocks, const int64_t *__restrict__ cand_idx,
    const double *__restrict__ cand_res, const double *__restrict__ cand_U, int n_cand, int r, int ldc,
    double *__restrict__ tau, double *__restrict__ rec, double tau_floor) {
  __shared__ double sv1[16], sv2[16], stau[16];
  __shared__ long long si1[16];
  __shared__ int spos[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Best b; b.init();
  int pos = -1;
  for (int c0 = threadIdx.x; c0 < n_cand; c0 += 4 * 1024) {   // four candidates per trip: the loads of a trip are independent
    double v[4];
    int64_t gi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + 1024 * u;
      const bool in = c < n_cand;
      v[u] = in ? cand_res[c] : -2.0;
      gi[u] = in ? cand_idx[c] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (gi[u] >= 0) {
        const bool better = v[u] > b.v1 || (v[u] == b.v1 && gi[u] < b.i1);
        b.push(v[u], gi[u]);
        pos = better ? c0 + 1024 * u : pos;
      }
  }
  for (int o = 32; o > 0; o >>= 1) {  // wave reduce carrying the winner's slot
    const double ov1 = __shfl_xor(b.v1, o, 64);
    const long long oi1 = __shfl_xor((long long)b.i1, o, 64);
    const double ov2 = __shfl_xor(b.v2, o, 64);
    const int op = __shfl_xor(pos, o, 64);
    const bool take = ov1 > b.v1 || (ov1 == b.v1 && oi1 < b.i1);
    b.merge(ov1, oi1, ov2);
    pos = take ? op : pos;
  }
  double t = -2.0;
  if (tops)
    for (int k = threadIdx.x; k < n_blocks; k += 1024) {
      const double v = tops[((int64_t)k * QR_TOPT + QR_TOPT - 1) * 2];
      t = v > t ? v : t;
    }
  for (int o = 32; o > 0; o >>= 1) { const double ot = __shfl_xor(t, o, 64); t = ot > t ? ot : t; }
  if (lane == 0) { sv1[wave] = b.v1; si1[wave] = b.i1; sv2[wave] = b.v2; spos[wave] = pos; stau[wave] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    Best g; g.init(); int gp = -1; double gt = -2.0;
    for (int w = 0; w < 16; ++w) {
      const bool take = sv1[w] > g.v1 || (sv1[w] == g.v1 && si1[w] < g.i1);
      g.merge(sv1[w], si1[w], sv2[w]);
      gp = take ? spos[w] : gp;
      gt = stau[w] > gt ? stau[w] : gt;
    }
    rec[0] = g.v1; rec[1] = (double)g.i1; rec[2] = g.v2;
    spos[0] = gp;
    if (tau) *tau = gt > tau_floor ? gt : tau_floor;   // tau_floor: the bound of the rows a pool sweep did not visit
  }
  __syncthreads();
  const int gp = spos[0];
  for (int k = threadIdx.x; k < r; k += 1024) rec[3 + k] = (gp >= 0) ? cand_U[(int64_t)gp * ldc + k] : 0.0;
}

// The two Gram-Schmidt passes of orth_step for r <= 128, Q read ONCE: the 256 threads form a 16 x 16 grid over the (step x r)
// block of earlier directions -- thread (a, b) keeps Q[a + 16 i][b + 16 j] in registers (NJ x NJ doubles, NJ = ceil(r / 16)) --
// so that the dot products c = Q v reduce over the 16 lanes of a row group (DPP) and the update v -= Q^T c over the row groups
// (two shuffles, then the four waves through LDS).  One L2 round trip per step instead of four chains of dependent loads: the
// step kernel's last workgroup spent up to 30 us here at step 63 of a 64-column basis, on a critical path of 64+ launches
// per placement.  part: 4 * 16 * NJ doubles of LDS.  v holds the candidate row on entry (synchronised), the residual on exit.
template <int NJ>
__device__ inline void orth_tile_load(const double *__restrict__ Q, int r, int step, double (&q)[NJ][NJ]) {
  const int b = threadIdx.x & 15, a = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    const int t = a + 16 * i;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = b + 16 * j;
      q[i][j] = (t < step && k < r) ? Q[(int64_t)t * r + k] : 0.0;
    }
  }
}

template <int NJ>
__device__ inline void orth_tile_apply(const double (&q)[NJ][NJ], int r, double *v, double *part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = lane & 15;
  for (int pass = 0; pass < 2; ++pass) {
    double vk[NJ], u[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int k = b + 16 * j; vk[j] = (k < r) ? v[k] : 0.0; u[j] = 0.0; }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      double d = 0.0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) d = fma(q[i][j], vk[j], d);
      d = group_sum_t<16>(d);                                  // c[a + 16 i], in all 16 lanes of the row group
#pragma unroll
      for (int j = 0; j < NJ; ++j) u[j] = fma(d, q[i][j], u[j]);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      u[j] += __shfl_xor(u[j], 16, 64);
      u[j] += __shfl_xor(u[j], 32, 64);
      if (lane < 16) part[(wave * NJ + j) * 16 + b] = u[j];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < r; k += QR_THREADS) {
      const int j = k >> 4, bb = k & 15;
      double acc = 0.0;
#pragma unroll
      for (int w = 0; w < QR_THREADS / 64; ++w) acc += part[(w * NJ + j) * 16 + bb];
      v[k] -= acc;
    }
    __syncthreads();
  }
}

template <int NJ>
__device__ inline void orth_gs_tile(const double *__restrict__ Q, int r, int step, double *v, double *part) {
  double q[NJ][NJ];
  orth_tile_load<NJ>(Q, r, step, q);
  orth_tile_apply<NJ>(q, r, v, part);
}

// the residual in v (r doubles of LDS, complete for every thread) -> the unit direction Q[step]; red: one double per wave
__device__ inline void orth_finish(const double *v, int r, int step, double *__restrict__ Q, double *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s = 0.0;
  for (int k = threadIdx.x; k < r; k += QR_THREADS) s += v[k] * v[k];
  s = group_sum(s, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  double nn = 0.0;
  for (int w = 0; w < QR_THREADS / 64; ++w) nn += red[w];
  const double inv = (nn > 0.0) ? 1.0 / sqrt(nn) : 0.0;
  for (int k = threadIdx.x; k < r; k += QR_THREADS) Q[(int64_t)step * r + k] = v[k] * inv;
}

// one workgroup: pick the winner among the ranks' records, certify it against tau, orthogonalise,
// store q / pivot / flags.  v, c: r doubles of LDS each (c: at least 512 doubles when r <= 128 -- the tiled passes stage their
// partial sums in it); red: one double per wave; win_p: one int (all LDS).  MAXNJ: 16 MAXNJ bounds r where the caller knows it
// at compile time (the fused step kernel), so that only the tile sizes that can occur are built into it.
template <int MAXNJ = 8>
__device__ inline void orth_step(const double *__restrict__ recs, int n_rec, const double *__restrict__ taus, int n_tau,
                                 int first, int r, int step, double *__restrict__ Q, int64_t *__restrict__ piv,
                                 double *__restrict__ gap, double *__restrict__ okflag, double *v, double *c, double *red,
                                 int *win_p, int tiled) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int stride = r + 3;
  if (threadIdx.x == 0) {
    int w = 0;
    for (int i = 1; i < n_rec; ++i) {
      const double vi = recs[(int64_t)i * stride], vw = recs[(int64_t)w * stride];
      if (vi > vw || (vi == vw && recs[(int64_t)i * stride + 1] < recs[(int64_t)w * stride + 1])) w = i;
    }
    *win_p = w;
    const double bestv = recs[(int64_t)w * stride];
    double second = recs[(int64_t)w * stride + 2];
    for (int i = 0; i < n_rec; ++i)
      if (i != w && recs[(int64_t)i * stride] > second) second = recs[(int64_t)i * stride];
    double tau = -2.0;
    for (int i = 0; i < n_tau; ++i) tau = taus[i] > tau ? taus[i] : tau;
    piv[step] = (int64_t)recs[(int64_t)w * stride + 1];
    if (gap) gap[step] = (bestv > 0.0) ? (bestv - second) / bestv : 0.0;   // candidates only: a lower bound on rivals
    okflag[step] = (first || bestv > tau) ? 1.0 : 0.0;
  }
  __syncthreads();
  const double *row = recs + (int64_t)(*win_p) * stride + 3;
  for (int k = threadIdx.x; k < r; k += QR_THREADS) v[k] = row[k];
  __syncthreads();
  if (tiled && r <= 16) orth_gs_tile<1>(Q, r, step, v, c);
  else if (tiled && MAXNJ >= 2 && r <= 32) orth_gs_tile<(MAXNJ >= 2 ? 2 : 1)>(Q, r, step, v, c);
  else if (tiled && MAXNJ >= 4 && r <= 64) orth_gs_tile<(MAXNJ >= 4 ? 4 : 1)>(Q, r, step, v, c);
  else if (tiled && MAXNJ >= 8 && r <= 128) orth_gs_tile<(MAXNJ >= 8 ? 8 : 1)>(Q, r, step, v, c);
  else
  for (int pass = 0; pass < 2; ++pass) {
    for (int t = wave; t < step; t += QR_THREADS / 64) {
      double d = 0.0;
      for (int k = lane; k < r; k += 64) d += Q[(int64_t)t * r + k] * v[k];
      d = group_sum(d, 64);
      if (lane == 0) c[t] = d;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < r; k += QR_THREADS) {
      double acc = v[k];
      for (int t = 0; t < step; ++t) acc -= c[t] * Q[(int64_t)t * r + k];
      v[k] = acc;
    }
    __syncthreads();
  }
  orth_finish(v, r, step, Q, red);
}


__global__ __launch_bounds__(QR_THREADS) void qr_orth_kernel(
    const double *__restrict__ recs, int n_rec, const double *__restrict__ taus, int n_tau, int first, int r,
    int step, double *__restrict__ Q, int64_t *__restrict__ piv, double *__restrict__ gap,
    double *__restrict__ okflag, unsigned *__restrict__ zero_me, int tiled) {
  __shared__ double v[SPR_MAX_R_WIDE], c[SPR_MAX_R_WIDE];
  __shared__ double red[QR_THREADS / 64];
  __shared__ int win;
  if (zero_me && threadIdx.x == 0) *zero_me = 0u;          // ticket counter of the fused step kernels that follow
  orth_step(recs, n_rec, taus, n_tau, first, r, step, Q, piv, gap, okflag, v, c, red, &win, tiled);
}

// |p - c| < d_min with the reference's arithmetic (np.linalg.norm of the difference, :649-652)
__device__ inline bool within(const double *p, const double *c, int dim, double d_min) {
  double d2 = 0.0;
  for (int d = 0; d < dim; ++d) d2 += (c[d] - p[d]) * (c[d] - p[d]);
  return sqrt(d2) < d_min;
}

// candidate residuals <- down-dated by direction q; the pivot's own slot leaves the race (and, for GEM
// placement, so does every candidate closer than d_min to it)
template <int LPR>
__global__ __launch_bounds__(QR_THREADS) void qr_cand_downdate_kernel(
    const double *__restrict__ cand_U, int n_cand, int r, int ldc, const int64_t *__restrict__ cand_idx,
    const double *__restrict__ q, const int64_t *__restrict__ piv_ptr, double *__restrict__ cand_res,
    const double *__restrict__ xyz, int dim, int64_t n_points, double d_min) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / LPR, lig = lane % LPR;
  const int k0 = 2 * lig;
  const double q0 = (k0 < r) ? q[k0] : 0.0;
  const double q1 = (k0 + 1 < r) ? q[k0 + 1] : 0.0;
  const int64_t piv = *piv_ptr;
  double pc[3] = {0.0, 0.0, 0.0};                       // position of the pick (GEM's d_min exclusion, :649-652)
  if (xyz)
    for (int d = 0; d < dim; ++d) pc[d] = xyz[(piv % n_points) * dim + d];
  const int stride = gridDim.x * (QR_THREADS / 64) * RPW;
  for (int c0 = (blockIdx.x * (QR_THREADS / 64) + wave) * RPW; c0 < n_cand; c0 += stride) {
    const int c = c0 + grp;
    const bool valid = c < n_cand;
    const f64x2 u = load_row_piece(cand_U + (int64_t)(valid ? c : 0) * ldc, k0, r, true, valid);
    const double old = (valid && lig == 0) ? cand_res[c] : 0.0;
    double v = downdate<LPR>(old, u, q0, q1);
    if (valid && lig == 0) {
      if (old < 0.0) v = old;
      if (cand_idx[c] == piv) v = -1.0;
      if (xyz && cand_idx[c] >= 0 && within(xyz + (cand_idx[c] % n_points) * dim, pc, dim, d_min)) v = -1.0;
      cand_res[c] = v;
    }
  }
}

// One launch per candidate step (one rank, r <= 128): the down-dating above, then every workgroup leaves its best / runner-up
// and takes a ticket; the LAST one to finish merges the partials (any order gives the same result: ties go to the lowest
// global row), writes the record, and -- unless this is the last step of the batch -- certifies and orthogonalises the
// NEXT step's winner right away (orth_step), so that a step is one launch instead of three.  partial[b] = (best, its global
// row, runner-up, its candidate slot); *ticket is zero on entry (qr_orth_kernel of the batch's first step, or the last
// workgroup of the previous launch, reset it).
template <int LPR>
__global__ __launch_bounds__(QR_THREADS) void qr_step_fused_kernel(
    const double *__restrict__ cand_U, int n_cand, int r, int ldc, const int64_t *__restrict__ cand_idx,
    double *__restrict__ Q, int64_t *__restrict__ piv, double *__restrict__ gap, double *__restrict__ okflag, int step,
    int do_next, const double *__restrict__ tau, double *__restrict__ cand_res, double *__restrict__ rec,
    const double *__restrict__ xyz, int dim, int64_t n_points, double d_min, double *__restrict__ partial,
    unsigned *__restrict__ ticket, int tiled) {
  constexpr int RPW = 64 / LPR, NWV = QR_THREADS / 64;
  __shared__ double sv1[NWV], sv2[NWV];
  __shared__ long long si1[NWV];
  __shared__ int spos[NWV];
  __shared__ int s_last;
  __shared__ double ov[SPR_MAX_R], oc[4 * SPR_MAX_R], ored[NWV];   // oc: 64 NJ doubles for the tiled Gram-Schmidt passes
  __shared__ int owin;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / LPR, lig = lane % LPR;
  const int k0 = 2 * lig;
  const double *q = Q + (int64_t)step * r;
  const double q0 = (k0 < r) ? q[k0] : 0.0;
  const double q1 = (k0 + 1 < r) ? q[k0 + 1] : 0.0;
  const int64_t pv = piv[step];
  const double tau_pre = (do_next && threadIdx.x == 0) ? tau[0] : 0.0;   // for the certification in the last workgroup's tail
  double pc[3] = {0.0, 0.0, 0.0};
  if (xyz)
    for (int d = 0; d < dim; ++d) pc[d] = xyz[(pv % n_points) * dim + d];
  Best b; b.init();
  int pos = -1;
  const int stride = gridDim.x * NWV * RPW;
  for (int c0 = (blockIdx.x * NWV + wave) * RPW; c0 < n_cand; c0 += stride) {
    const int c = c0 + grp;
    const bool valid = c < n_cand;
    const f64x2 u = load_row_piece(cand_U + (int64_t)(valid ? c : 0) * ldc, k0, r, true, valid);
    const double old = (valid && lig == 0) ? cand_res[c] : 0.0;
    double v = downdate<LPR>(old, u, q0, q1);
    if (valid && lig == 0) {
      const int64_t gi = cand_idx[c];
      if (old < 0.0) v = old;
      if (gi == pv) v = -1.0;
      if (xyz && gi >= 0 && within(xyz + (gi % n_points) * dim, pc, dim, d_min)) v = -1.0;
      cand_res[c] = v;
      if (gi >= 0) {
        const bool better = v > b.v1 || (v == b.v1 && gi < b.i1);
        b.push(v, gi);
        pos = better ? c : pos;
      }
    }
  }
  auto wave_merge = [&]() {                                  // wave reduce carrying the winner's slot
    for (int o = 32; o > 0; o >>= 1) {
      const double ov1 = __shfl_xor(b.v1, o, 64);
      const long long oi1 = __shfl_xor((long long)b.i1, o, 64);
      const double ov2 = __shfl_xor(b.v2, o, 64);
      const int op = __shfl_xor(pos, o, 64);
      const bool take = ov1 > b.v1 || (ov1 == b.v1 && oi1 < b.i1);
      b.merge(ov1, oi1, ov2);
      pos = take ? op : pos;
    }
  };
  auto block_merge = [&]() {                                 // -> thread 0 holds the workgroup's result
    wave_merge();
    if (lane == 0) { sv1[wave] = b.v1; si1[wave] = b.i1; sv2[wave] = b.v2; spos[wave] = pos; }
    __syncthreads();
    if (threadIdx.x == 0) {
      Best g; g.init(); int gp = -1;
      for (int w = 0; w < NWV; ++w) {
        const bool take = sv1[w] > g.v1 || (sv1[w] == g.v1 && si1[w] < g.i1);
        g.merge(sv1[w], si1[w], sv2[w]);
        gp = take ? spos[w] : gp;
      }
      b = g; pos = gp;
    }
  };
  block_merge();
  if (threadIdx.x == 0) {
    double *pp = partial + 4 * (int64_t)blockIdx.x;
    pp[0] = b.v1; pp[1] = (double)b.i1; pp[2] = b.v2; pp[3] = (double)pos;
    __threadfence();                                         // the partial is visible before the ticket is
    const unsigned t = atomicAdd(ticket, 1u);
    s_last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // ---- last workgroup: merge the partials, write the record
  // (the directions the NEXT winner will be orthogonalised against -- rows 0 .. step of Q, all final -- are requested first
  // where they fit the registers: their L2 round trip runs under the merge)
  constexpr int NJL = LPR >= 64 ? 8 : LPR >= 32 ? 4 : LPR >= 16 ? 2 : 1;     // r <= 2 LPR <= 16 NJL
  const bool tile_next = tiled != 0 && do_next != 0;
  double qt[NJL][NJL];
  constexpr bool EARLY = NJL <= 4;                           // 64 doubles held over the merge would not fit the registers
  if (EARLY && tile_next) orth_tile_load<NJL>(Q, r, step + 1, qt);
  b.init(); pos = -1;
  for (int p = threadIdx.x; p < (int)gridDim.x; p += QR_THREADS) {
    const volatile double *pp = partial + 4 * (int64_t)p;
    const double v1 = pp[0], v2 = pp[2];
    const long long i1 = (long long)pp[1];
    const int ps = (int)pp[3];
    const bool take = v1 > b.v1 || (v1 == b.v1 && i1 < b.i1);
    b.merge(v1, i1, v2);
    pos = take ? ps : pos;
  }
  __syncthreads();                                           // the per-wave staging arrays are reused
  block_merge();
  if (threadIdx.x == 0) {
    rec[0] = b.v1; rec[1] = (double)b.i1; rec[2] = b.v2;
    spos[0] = pos;
    *ticket = 0u;
  }
  __syncthreads();
  const int gp = spos[0];
  for (int k = threadIdx.x; k < r; k += QR_THREADS) {
    const double val = (gp >= 0) ? cand_U[(int64_t)gp * ldc + k] : 0.0;
    rec[3 + k] = val;
    if (tile_next) ov[k] = val;                              // the next step's candidate row, without reading the record back
  }
  if (!do_next) return;
  if (!tile_next) {
    __syncthreads();                                         // the record is complete for every thread of this workgroup
    orth_step<NJL>(rec, 1, tau, 1, 0, r, step + 1, Q, piv, gap, okflag, ov, oc, ored, &owin, tiled);
    return;
  }
  // what orth_step does with ONE record, from the values this workgroup still holds: certify the winner against tau (requested
  // at the start of the kernel), orthogonalise its row against the directions already in registers, store the new direction
  if (threadIdx.x == 0) {
    piv[step + 1] = (int64_t)b.i1;
    if (gap) gap[step + 1] = (b.v1 > 0.0) ? (b.v1 - b.v2) / b.v1 : 0.0;
    okflag[step + 1] = (b.v1 > tau_pre) ? 1.0 : 0.0;
  }
  if (!EARLY) orth_tile_load<NJL>(Q, r, step + 1, qt);
  __syncthreads();
  orth_tile_apply<NJL>(qt, r, ov, oc);
  orth_finish(ov, r, step + 1, Q, ored);
}

// the same for rows longer than 128 entries: 64 lanes per row, each walks its column pairs
__global__ __launch_bounds__(QR_THREADS) void qr_cand_downdate_wide_kernel(
    const double *__restrict__ cand_U, int n_cand, int r, int ldc, const int64_t *__restrict__ cand_idx,
    const double *__restrict__ q, const int64_t *__restrict__ piv_ptr, double *__restrict__ cand_res,
    const double *__restrict__ xyz, int dim, int64_t n_points, double d_min) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t piv = *piv_ptr;
  double pc[3] = {0.0, 0.0, 0.0};
  if (xyz)
    for (int d = 0; d < dim; ++d) pc[d] = xyz[(piv % n_points) * dim + d];
  for (int c = blockIdx.x * (QR_THREADS / 64) + wave; c < n_cand; c += gridDim.x * (QR_THREADS / 64)) {
    const double *u = cand_U + (int64_t)c * ldc;
    double d = 0.0;
    for (int k = lane; k < r; k += 64) d += u[k] * q[k];
    d = group_sum_t<64>(d);
    if (lane == 0) {
      const double old = cand_res[c];
      double v = old - d * d;
      v = v < 0.0 ? 0.0 : v;
      if (old < 0.0) v = old;
      if (cand_idx[c] == piv) v = -1.0;
      if (xyz && cand_idx[c] >= 0 && within(xyz + (cand_idx[c] % n_points) * dim, pc, dim, d_min)) v = -1.0;
      cand_res[c] = v;
    }
  }
}

__global__ void qr_mark_kernel(const int64_t *__restrict__ piv, int n, int64_t row0, int64_t n_rows,
                               double *__restrict__ nrm) {
  const int t = threadIdx.x;
  if (t < n) {
    const int64_t li = piv[t] - row0;
    if (li >= 0 && li < n_rows) nrm[li] = -1.0;
  }
}

// GEM placement (:586-698): rows outside the search mask and rows closer than d_min to one of the picks
// piv[0..nq) leave the pool for good (nrm = -1, as for rows already chosen)
__global__ __launch_bounds__(QR_THREADS) void qr_exclude_kernel(double *__restrict__ nrm, int64_t n_rows, int64_t row0,
                                                                int64_t n_points, const uint8_t *__restrict__ mask,
                                                                const double *__restrict__ xyz, int dim,
                                                                const int64_t *__restrict__ piv, int nq, double d_min) {
  __shared__ double ctr[QR_BATCH][3];
  if (threadIdx.x < nq && xyz) {
    const int64_t g = piv[threadIdx.x];
    for (int d = 0; d < 3; ++d) ctr[threadIdx.x][d] = (d < dim && g >= 0) ? xyz[(g % n_points) * dim + d] : INFINITY;
  }
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * QR_THREADS + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * QR_THREADS) {
    bool out = mask && !mask[i];
    if (!out && xyz && nq > 0) {
      const double *p = xyz + ((row0 + i) % n_points) * dim;
      for (int t = 0; t < nq; ++t) out = out || within(p, ctr[t], dim, d_min);
    }
    if (out) nrm[i] = -1.0;
  }
}

template <typename TU>
__global__ void mask_rows_kernel(TU *__restrict__ Ur, int64_t n_rows, int r, int64_t ldu,
                                 const uint8_t *__restrict__ mask) {
  const int64_t total = n_rows * r;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / r;
    if (!mask[row]) Ur[row * ldu + (e - row * r)] = (TU)0;
  }
}

// SPR_QR_ORTH_TILE=0: the Gram-Schmidt passes of a step as chains of loads (the form before round 5's orth_gs_tile; A/B only)
int orth_tiled() {
  static const int on = [] { const char *e = getenv("SPR_QR_ORTH_TILE"); return (e && e[0] == '0') ? 0 : 1; }();
  return on;
}

int pick_lpr(int r) {
  const int half = (r + 1) / 2;
  int l = 1;
  while (l < half) l *= 2;
  return l;
}

// workgroups of a sweep over Ur (each keeps its QR_TOPT best rows: grid * QR_TOPT candidates): one per 64-row panel,
// at most 4 per CU and QR_MAX_BLOCKS
int sweep_grid(int64_t n_rows) {
  const int rows_it = 64;
  int64_t steps = (n_rows + rows_it - 1) / rows_it;
  const int cus = spr_cached_cus();
  int64_t cap = 4LL * (cus > 0 ? cus : 256);
  if (cap > QR_MAX_BLOCKS) cap = QR_MAX_BLOCKS;
  return (int)(steps < cap ? steps : cap);
}

// workspace carve-up (the candidate rows are as wide as the basis: at least SPR_MAX_R columns are always reserved, so the
// layout for r <= 128 is the one spr_qr_workspace() has always described)
struct QrWs {
  double *tops;       // [QR_MAX_BLOCKS][QR_TOPT][2]
  int64_t *cand_idx;  // [NC]
  double *cand_res;   // [NC]
  double *cand_U;     // [NC][max(r + (r & 1), SPR_MAX_R)]
  double *tau;        // [1]
  static constexpr int64_t NC = (int64_t)QR_MAX_BLOCKS * QR_TOPT;
  static int64_t width(int r) { const int w = r + (r & 1); return w > SPR_MAX_R ? w : SPR_MAX_R; }
  static size_t bytes(int r) { return sizeof(double) * ((size_t)NC * 2 + NC * 2 + NC * width(r) + 8); }
  QrWs(void *p, int r) {
    double *d = static_cast<double *>(p);
    tops = d; d += NC * 2;
    cand_idx = reinterpret_cast<int64_t *>(d); d += NC;
    cand_res = d; d += NC;
    cand_U = d; d += NC * width(r);
    tau = d;
  }
};

template <typename TU, bool INIT>
int launch_refresh(int grid, hipStream_t st, const TU *Ur, int64_t n_rows, int r, int64_t ldu, int vec_ok, int64_t row0,
                   const double *Qj, int nq, double *nrm, double *tops) {
  if (r > SPR_MAX_R) {
    const bool vec = (r % 16 == 0) && ((ldu * sizeof(TU)) % 16 == 0) && ((reinterpret_cast<uintptr_t>(Ur) & 15) == 0);
#define RW(NG, V) hipLaunchKernelGGL((qr_refresh_wide_kernel<NG, V, TU, INIT>), dim3(grid), dim3(QR_THREADS), 0, st, Ur, n_rows, r, ldu, row0, Qj, nq, nrm, tops)
#define RWV(NG) do { if (vec) RW(NG, 1); else RW(NG, 0); } while (0)
    if (INIT) RWV(16);                                        // no direction image in LDS: one instantiation serves all r
    else if (r <= 256) RWV(16);
    else if (r <= 512) RWV(32);
    else RWV(64);
#undef RWV
#undef RW
    SPR_LAUNCH_CHECK();
    return SPR_OK;
  }
  const int mtr = spr_round_mt(r);      // padded width of Ur in 16-column tiles (r <= 128 -> <= 8)
  const int lm = vec_ok ? ((r == 16 * mtr) ? 2 : 1) : 0;
  // register-direct form: whole 16-column groups, 16-byte aligned pieces (SPR_QR_DIRECT=0 keeps the LDS-panel form)
  static const bool direct_on = [] { const char *e = getenv("SPR_QR_DIRECT"); return !(e && e[0] == '0'); }();
  // measured (MI355X, 9M rows x 64, one launch): init 1.13 ms direct vs 1.39 LDS; f64 refresh with 8 directions 1.02 vs
  // 0.91 (the LDS form's 4-rows-per-instruction loads stream better); f32 basis, config-5 share: placement 138 vs 186 ms
  const bool want_direct = INIT || std::is_same<TU, float>::value;
  if (direct_on && want_direct && r % 16 == 0 && (ldu * sizeof(TU)) % 16 == 0 &&
      (reinterpret_cast<uintptr_t>(Ur) & 15) == 0) {
#define RD(NGV) hipLaunchKernelGGL((qr_refresh_direct_kernel<NGV, TU, INIT>), dim3(grid), dim3(QR_THREADS), 0, st, Ur, n_rows, r, ldu, row0, Qj, nq, nrm, tops)
    switch (r / 16) {
      case 1: RD(1); break;
      case 2: RD(2); break;
      case 3: RD(3); break;
      case 4: RD(4); break;
      case 5: RD(5); break;
      case 6: RD(6); break;
      case 7: RD(7); break;
      default: RD(8); break;
    }
#undef RD
    SPR_LAUNCH_CHECK();
    return SPR_OK;
  }
#define RF(MTV, LM) hipLaunchKernelGGL((qr_refresh_mfma_kernel<MTV, LM, TU, INIT>), dim3(grid), dim3(QR_THREADS), 0, st, Ur, n_rows, r, ldu, row0, Qj, nq, nrm, tops)
#define RFV(MTV) do { if (lm == 2) RF(MTV, 2); else if (lm == 1) RF(MTV, 1); else RF(MTV, 0); } while (0)
  switch (mtr) {
    case 1: RFV(1); break;
    case 2: RFV(2); break;
    case 3: RFV(3); break;
    case 4: RFV(4); break;
    case 6: RFV(6); break;
    default: RFV(8); break;
  }
#undef RFV
#undef RF
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

int check_ur(const char *who, const void *Ur, int64_t n_rows, int32_t r, int64_t ldu) {
  SPR_REQUIRE(Ur != nullptr, SPR_E_INVALID, "%s: Ur is NULL", who);
  SPR_REQUIRE(n_rows > 0 && r > 0 && ldu >= r, SPR_E_INVALID, "%s: bad shape n_rows=%lld r=%d ldu=%lld", who,
              (long long)n_rows, r, (long long)ldu);
  SPR_REQUIRE(r <= SPR_MAX_R_WIDE, SPR_E_UNSUPPORTED, "%s: r=%d > %d not built", who, r, SPR_MAX_R_WIDE);
  return SPR_OK;
}

// ---- Epoch sweeps: refreshes that visit only the rows that can still matter --------------------------------------
// The certified batches above end when the winner's residual reaches tau, the largest norm a non-candidate may have;
// with 16 candidates per sweep block that is after ~16 steps (tools/cert_probe.py: config 3, tau = 0.75 of the first
// winner, winners shrinking by 0.77 per 16 steps), and every refresh re-reads the whole basis.  But a row whose norm is
// far below the winners cannot be picked for many steps: its STALE norm is still an upper bound (norms only decrease).
// An epoch starts with norms nrm_e that are exact for the directions [0, j_e) (the initial norms; later a full epoch
// sweep).  The rows with nrm_e > theta form the POOL (a sorted index list); every refresh inside the epoch visits only
// the pool -- residual = nrm_e - sum over ALL directions of the epoch (u . q)^2, from scratch, so no per-row record of
// what has been applied is needed --, and the steps are certified against max(tau of the pool, theta).  When the
// winners come down to theta, ONE full sweep applies the epoch's directions to every row (rewriting nrm_e) and a new
// epoch starts with a lower theta.  The directions of an epoch (up to NT tiles of 16) sit in LDS in fragment order;
// rows go HBM -> registers in the MFMA A layout exactly as in qr_refresh_direct_kernel.
// The MFMAs of an epoch sweep for one 16-row block: NTT direction tiles side by side -- NTT independent accumulator chains, every
// A operand used for NTT MFMAs -- instead of one tile after the other (16 dependent v_mfma_f64 per tile: a wave then issues one
// MFMA per result latency, and a full sweep's time grew by 1.2-1.9 ms per extra tile, profiles/r05_epoch_sweep_ab.txt).
template <int NG, int NTT, typename P4>
__device__ inline void epoch_tiles_mfma(const P4 (&cur)[NG], const double *__restrict__ ql0, double (&q2)[4]) {
  f64x4 acc[NTT];
#pragma unroll
  for (int tl = 0; tl < NTT; ++tl) acc[tl] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const double a4[4] = {(double)cur[g].x, (double)cur[g].y, (double)cur[g].z, (double)cur[g].w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int tl = 0; tl < NTT; ++tl)
        acc[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[t], ql0[tl * NG * 256 + (g * 4 + t) * 64], acc[tl], 0, 0, 0);
    }
  }
#pragma unroll
  for (int tl = 0; tl < NTT; ++tl) {
    q2[0] = fma(acc[tl].x, acc[tl].x, q2[0]); q2[1] = fma(acc[tl].y, acc[tl].y, q2[1]);
    q2[2] = fma(acc[tl].z, acc[tl].z, q2[2]); q2[3] = fma(acc[tl].w, acc[tl].w, q2[3]);
  }
}

template <int NG, int NT, typename TU, bool POOL, bool ILP = false>
__global__ __launch_bounds__(QR_THREADS, (NG <= 4 && !ILP ? 3 : 2)) void qr_epoch_sweep_kernel(
    const TU *__restrict__ Ur, int64_t n_rows, int r, int64_t ldu, int64_t row0, const double *__restrict__ Q, int nq,
    double *__restrict__ nrm_e, double *__restrict__ nrm, const int32_t *__restrict__ pool,
    const int32_t *__restrict__ pool_n_ptr, double *__restrict__ tops) {
  constexpr int R = 64, SPW = QR_SPW;
  __shared__ double smem[2 * (QR_THREADS / 64) * SPW * QR_TOPT];
  __shared__ double Ql[NT * NG * 256];     // Ql[((tile NG + g) 4 + t) 64 + kk 16 + li] = Q[16 tile + li][16 g + 4 kk + t]
  double *const sval = smem;
  long long *const sidx = reinterpret_cast<long long *>(smem + (QR_THREADS / 64) * SPW * QR_TOPT);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, kk = lane >> 4;
  using P4 = typename std::conditional<std::is_same<TU, float>::value, float4, double4>::type;
  const int nt = (nq + 15) / 16;
  for (int e = threadIdx.x; e < nt * NG * 256; e += QR_THREADS) {
    const int j = e & 15, k4 = (e >> 4) & 3, t = (e >> 6) & 3, g = (e >> 8) % NG, tile = (e >> 8) / NG;
    const int d = 16 * tile + j;
    Ql[e] = (d < nq) ? Q[(int64_t)d * r + 16 * g + 4 * k4 + t] : 0.0;
  }
  __syncthreads();
  TopListT<int> top;               // local rows (< 2^31: spr_qr_epoch_supported); row0 is added at the merge
  top.init();
  top.base = row0;
  const RowOwner<false> who(lane);
  const int64_t n_units = POOL ? (int64_t)*pool_n_ptr : n_rows;      // rows to visit
  const int64_t npanels = (n_units + R - 1) / R;
  // row of this lane's slot in panel c (its A operand) and the row this lane owns the norm of; -1: past the end
  auto slot_row = [&](int64_t c) -> int64_t {
    const int64_t u = c * R + wave * 16 + li;
    if (u >= n_units) return -1;
    return POOL ? (int64_t)pool[u] : u;
  };
  auto load_block = [&](int64_t row, P4 (&dst)[NG]) {
    const TU *rp = Ur + (row >= 0 ? row : 0) * ldu + 4 * kk;   // past the end: a harmless re-read of row 0, never stored
#pragma unroll
    for (int g = 0; g < NG; ++g) dst[g] = *reinterpret_cast<const P4 *>(rp + 16 * g);
  };
  auto own_row = [&](int64_t row) -> int64_t {                 // lanes 0..15 hold the rows of slots 0..15
    const int lo = __shfl((int)(row & 0xffffffff), who.off, 64), hi = __shfl((int)(row >> 32), who.off, 64);
    return ((int64_t)hi << 32) | (uint32_t)lo;
  };
  // NBUF register sets per wave, used in rotation (no copies): while set u is multiplied, the rows NBUF - 1 panels ahead
  // are requested into the set that was multiplied last.  Three sets (two panels ahead) where a set is <= 32 VGPRs: with two
  // waves per SIMD one panel ahead leaves 16 MB in flight on the chip -- 4.3 TB/s for a full sweep at config 3, 10.6 ms.
  constexpr int NBUF = (QR_EPOCH_DEEP && NG * sizeof(P4) / 4 <= 32 && NG <= 6) ? 3 : 2;
  P4 buf[NBUF][NG];
  double oldv[NBUF];
  int64_t orv[NBUF];
  auto fetch = [&](int64_t cc, P4 (&dst)[NG], double &old, int64_t &orow) {
    const int64_t row = cc < npanels ? slot_row(cc) : -1;
    load_block(row, dst);
    orow = own_row(row);
    old = nrm_e[orow >= 0 ? orow : 0];
  };
  auto process = [&](const P4 (&cur)[NG], double old, int64_t orow) {
    const bool mine = who.own && orow >= 0;
    double q2[4] = {0.0, 0.0, 0.0, 0.0};                       // this lane's direction column, summed over the tiles
    if (ILP) {                                                 // the tiles' accumulator chains side by side
      const double *ql0 = Ql + kk * 16 + li;
      if (nt == 1) epoch_tiles_mfma<NG, 1, P4>(cur, ql0, q2);
      else if (nt == 2) epoch_tiles_mfma<NG, (NT >= 2 ? 2 : 1), P4>(cur, ql0, q2);
      else if (nt == 3) epoch_tiles_mfma<NG, (NT >= 3 ? 3 : 1), P4>(cur, ql0, q2);
      else epoch_tiles_mfma<NG, (NT >= 4 ? 4 : 1), P4>(cur, ql0, q2);
    } else
    for (int tile = 0; tile < nt; ++tile) {
      const double *ql = Ql + tile * NG * 256 + kk * 16 + li;
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const double a4[4] = {(double)cur[g].x, (double)cur[g].y, (double)cur[g].z, (double)cur[g].w};
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[t], ql[(g * 4 + t) * 64], acc, 0, 0, 0);
      }
      q2[0] = fma(acc.x, acc.x, q2[0]); q2[1] = fma(acc.y, acc.y, q2[1]);
      q2[2] = fma(acc.z, acc.z, q2[2]); q2[3] = fma(acc.w, acc.w, q2[3]);
    }
    const double d2 = who.pick(group_sum_t<16>(q2[0]), group_sum_t<16>(q2[1]), group_sum_t<16>(q2[2]), group_sum_t<16>(q2[3]));
    double v = old - d2;
    v = v < 0.0 ? 0.0 : v;
    v = old < 0.0 ? -1.0 : v;
    if (mine) {
      nrm[orow] = v;
      if (!POOL) nrm_e[orow] = v;                              // full sweep: the next epoch starts from these
    }
    top.insert(v, (int)orow, mine);
  };
  int64_t c = blockIdx.x;
  const int64_t gs = gridDim.x;
#pragma unroll
  for (int u = 0; u < NBUF - 1; ++u) fetch(c + u * gs, buf[u], oldv[u], orv[u]);
  while (c < npanels) {
#pragma unroll
    for (int u = 0; u < NBUF; ++u) {
      if (c < npanels) {                                       // wave-uniform
        constexpr int NB1 = NBUF - 1;
        fetch(c + NB1 * gs, buf[(u + NB1) % NBUF], oldv[(u + NB1) % NBUF], orv[(u + NB1) % NBUF]);
        process(buf[u], oldv[u], orv[u]);
        c += gs;
      }
    }
  }
  top.template block_merge<SPW>(sval, sidx, tops + (int64_t)blockIdx.x * QR_TOPT * 2, who.own, who.slot(lane));
}

// ---- pool = sorted list of the rows with nrm_e > theta: count per chunk, scan, fill (deterministic order) ------------
constexpr int PB_THREADS = 256;
constexpr int PB_MAX_BLOCKS = 2048;

__global__ __launch_bounds__(PB_THREADS) void qr_pool_count_kernel(const double *__restrict__ nrm_e, int64_t n_rows,
                                                                   int64_t chunk, double theta, int32_t *__restrict__ counts) {
  __shared__ int red[PB_THREADS / 64];
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n_rows ? lo + chunk : n_rows;
  int cnt = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += PB_THREADS) cnt += nrm_e[i] > theta;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < PB_THREADS / 64; ++w) t += red[w];
    counts[blockIdx.x] = t;
  }
}

// one workgroup: exclusive scan of the chunk counts (in place); pool_n = total, or -1 when the list would not fit.
// Thread t owns the contiguous counts [t per, (t + 1) per); wave scan by shuffles, the 16 wave totals through LDS.
__global__ __launch_bounds__(1024) void qr_pool_scan_kernel(int32_t *__restrict__ counts, int n_blocks, int64_t cap,
                                                            int32_t *__restrict__ pool_n) {
  constexpr int PER = (PB_MAX_BLOCKS + 1023) / 1024;
  __shared__ int wsum[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int v[PER], mine = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int b = t * PER + i;
    v[i] = b < n_blocks ? counts[b] : 0;
    mine += v[i];
  }
  int incl = mine;                                             // inclusive scan over the wave's threads
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int before = 0, total = 0;
  for (int w = 0; w < 16; ++w) { if (w < wave) before += wsum[w]; total += wsum[w]; }
  int run = before + incl - mine;                              // rows ahead of this thread's first chunk
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int b = t * PER + i;
    if (b < n_blocks) counts[b] = run;
    run += v[i];
  }
  if (t == 0) *pool_n = (int64_t)total <= cap ? total : -1;
}

__global__ __launch_bounds__(PB_THREADS) void qr_pool_fill_kernel(const double *__restrict__ nrm_e, int64_t n_rows,
                                                                  int64_t chunk, double theta,
                                                                  const int32_t *__restrict__ offsets,
                                                                  const int32_t *__restrict__ pool_n,
                                                                  int32_t *__restrict__ pool) {
  __shared__ int wcnt[PB_THREADS / 64];
  if (*pool_n < 0) return;                                       // would not fit: the caller falls back to full sweeps
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n_rows ? lo + chunk : n_rows;
  int64_t base = offsets[blockIdx.x];
  for (int64_t i0 = lo; i0 < hi; i0 += PB_THREADS) {             // workgroup-uniform trip count
    const int64_t i = i0 + threadIdx.x;
    const bool in = i < hi && nrm_e[i] > theta;
    const unsigned long long m = __ballot(in);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < PB_THREADS / 64; ++w) { if (w < wave) woff += wcnt[w]; tot += wcnt[w]; }
    if (in) pool[base + woff + before] = (int32_t)i;
    base += tot;
    __syncthreads();
  }
}

// candidates from the last sweep's block tops + this rank's record and tau
template <typename TU>
int build_candidates(const QrWs &w, int grid, const TU *Ur, int64_t n_rows, int r, int64_t ldu, int64_t row0,
                     double *rec, double *tau_out, hipStream_t st, double tau_floor = -2.0) {
  const int ldc = r + (r & 1);
  hipLaunchKernelGGL(qr_gather_kernel<TU>, dim3(grid), dim3(QR_THREADS), 0, st, w.tops, Ur, r, ldu, row0, n_rows, ldc,
                     w.cand_idx, w.cand_res, w.cand_U);
  SPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(qr_cand_best_kernel, dim3(1), dim3(1024), 0, st, w.tops, grid, w.cand_idx, w.cand_res,
                     w.cand_U, grid * QR_TOPT, r, ldc, tau_out, rec, tau_floor);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

}  // namespace

extern "C" size_t spr_qr_workspace(int64_t n_rows) {
  (void)n_rows;
  return QrWs::bytes(SPR_MAX_R);
}

extern "C" size_t spr_qr_workspace_r(int64_t n_rows, int32_t r) {
  (void)n_rows;
  return (r > 0 && r <= SPR_MAX_R_WIDE) ? QrWs::bytes(r) : 0;
}

extern "C" int32_t spr_qr_batch(void) { return QR_BATCH; }

template <typename TU>
static int mask_rows_entry(const char *who, TU *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, const uint8_t *d_mask,
                           void *stream) {
  int rc = check_ur(who, d_Ur, n_rows, r, ldu);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_mask != nullptr, SPR_E_INVALID, "%s: mask is NULL", who);
  const int64_t total = n_rows * r;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mask_rows_kernel<TU>, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_Ur,
                     n_rows, (int)r, ldu, d_mask);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

extern "C" int spr_mask_rows_f64(double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, const uint8_t *d_mask,
                                 void *stream) {
  return mask_rows_entry("spr_mask_rows_f64", d_Ur, n_rows, r, ldu, d_mask, stream);
}

extern "C" int spr_mask_rows_u32(float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, const uint8_t *d_mask,
                                 void *stream) {
  return mask_rows_entry("spr_mask_rows_u32", d_Ur, n_rows, r, ldu, d_mask, stream);
}

template <typename TU>
static int qr_init_entry(const char *who, const TU *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                         double *d_nrm, double *d_rec, double *d_tau, void *d_workspace, size_t workspace_bytes,
                         void *stream) {
  int rc = check_ur(who, d_Ur, n_rows, r, ldu);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_nrm && d_rec && d_tau && d_workspace, SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(workspace_bytes >= QrWs::bytes(r), SPR_E_WORKSPACE, "%s: workspace too small", who);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = sweep_grid(n_rows);
  const int vec_ok = (r % 2 == 0) && (ldu % 2 == 0) && ((reinterpret_cast<uintptr_t>(d_Ur) & (2 * sizeof(TU) - 1)) == 0);
  QrWs w(d_workspace, r);
  rc = launch_refresh<TU, true>(grid, st, d_Ur, n_rows, r, ldu, vec_ok, row0, nullptr, 0, d_nrm, w.tops);
  if (rc != SPR_OK) return rc;
  return build_candidates<TU>(w, grid, d_Ur, n_rows, r, ldu, row0, d_rec, d_tau, st);
}

extern "C" int spr_qr_init_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                               double *d_nrm, double *d_rec, double *d_tau, void *d_workspace,
                               size_t workspace_bytes, void *stream) {
  return qr_init_entry("spr_qr_init_f64", d_Ur, n_rows, r, ldu, row0, d_nrm, d_rec, d_tau, d_workspace, workspace_bytes,
                       stream);
}

extern "C" int spr_qr_init_u32(const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                               double *d_nrm, double *d_rec, double *d_tau, void *d_workspace,
                               size_t workspace_bytes, void *stream) {
  return qr_init_entry("spr_qr_init_u32", d_Ur, n_rows, r, ldu, row0, d_nrm, d_rec, d_tau, d_workspace, workspace_bytes,
                       stream);
}

// ---- the same start from squared row norms that already exist (d_nrm0, left by spr_project_norms_* /
// spr_project_stream_norms_* when it stored d_Ur): no pass over the basis.  d_nrm0 is not modified.
template <typename TU>
static int qr_init_norms_entry(const char *who, const TU *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                               const double *d_nrm0, double *d_nrm, double *d_rec, double *d_tau, void *d_workspace,
                               size_t workspace_bytes, void *stream) {
  int rc = check_ur(who, d_Ur, n_rows, r, ldu);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_nrm0 && d_nrm && d_rec && d_tau && d_workspace, SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(d_nrm0 != d_nrm, SPR_E_INVALID, "%s: the given norms and the working vector must be different buffers", who);
  SPR_REQUIRE(workspace_bytes >= QrWs::bytes(r), SPR_E_WORKSPACE, "%s: workspace too small", who);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = sweep_grid(n_rows);
  QrWs w(d_workspace, r);
  hipLaunchKernelGGL(qr_tops_from_norms_kernel, dim3(grid), dim3(QR_THREADS), 0, st, d_nrm0, n_rows, row0, d_nrm, w.tops);
  SPR_LAUNCH_CHECK();
  return build_candidates<TU>(w, grid, d_Ur, n_rows, r, ldu, row0, d_rec, d_tau, st);
}

extern "C" int spr_qr_init_norms_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                                     const double *d_nrm0, double *d_nrm, double *d_rec, double *d_tau,
                                     void *d_workspace, size_t workspace_bytes, void *stream) {
  return qr_init_norms_entry("spr_qr_init_norms_f64", d_Ur, n_rows, r, ldu, row0, d_nrm0, d_nrm, d_rec, d_tau,
                             d_workspace, workspace_bytes, stream);
}

extern "C" int spr_qr_init_norms_u32(const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                                     const double *d_nrm0, double *d_nrm, double *d_rec, double *d_tau,
                                     void *d_workspace, size_t workspace_bytes, void *stream) {
  return qr_init_norms_entry("spr_qr_init_norms_u32", d_Ur, n_rows, r, ldu, row0, d_nrm0, d_nrm, d_rec, d_tau,
                             d_workspace, workspace_bytes, stream);
}

extern "C" int spr_qr_step_f64(int64_t n_rows, int32_t r, int32_t step, const double *d_recs, int32_t n_rec,
                               const double *d_taus, int32_t n_tau, int32_t first, double *d_Q, int64_t *d_piv,
                               double *d_gap, double *d_ok, double *d_rec, const double *d_xyz, int32_t xyz_dim,
                               int64_t n_points, double d_min, void *d_workspace, size_t workspace_bytes,
                               void *stream) {
  SPR_REQUIRE(d_recs && d_taus && d_Q && d_piv && d_ok && d_rec && d_workspace, SPR_E_INVALID,
              "spr_qr_step_f64: NULL pointer");
  SPR_REQUIRE(!d_xyz || (xyz_dim >= 1 && xyz_dim <= 3 && n_points > 0), SPR_E_INVALID,
              "spr_qr_step_f64: xyz needs 1..3 columns and n_points > 0");
  SPR_REQUIRE(n_rows > 0 && r > 0 && r <= SPR_MAX_R_WIDE && step >= 0 && step < r && n_rec >= 1 && n_tau >= 1,
              SPR_E_INVALID, "spr_qr_step_f64: bad r=%d step=%d n_rec=%d", r, step, n_rec);
  SPR_REQUIRE(workspace_bytes >= QrWs::bytes(r), SPR_E_WORKSPACE, "spr_qr_step_f64: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  QrWs w(d_workspace, r);
  const int lpr = pick_lpr(r);
  const int ldc = r + (r & 1), n_cand = sweep_grid(n_rows) * QR_TOPT;
  static const bool fused_on = [] { const char *e = getenv("SPR_QR_FUSED_STEPS"); return !(e && e[0] == '0'); }();
  const bool fused = fused_on && r <= SPR_MAX_R;
  // fused: the down-dating and the search for the next best candidate in ONE launch behind the orthogonalisation (the step
  // kernel of spr_qr_steps_f64 with do_next = 0: every workgroup leaves its best, the last one merges them and writes the
  // record) instead of a down-date launch and a one-workgroup scan of all candidates (16 us) -- two launches per step of a
  // sharded placement instead of three.  The block lists of the last sweep are dead while steps run: their space holds the
  // partial results and the ticket, which qr_orth_kernel zeroes.
  double *partial = w.tops;
  unsigned *ticket = reinterpret_cast<unsigned *>(w.tops + 4 * 256);
  hipLaunchKernelGGL(qr_orth_kernel, dim3(1), dim3(QR_THREADS), 0, st, d_recs, (int)n_rec, d_taus, (int)n_tau,
                     (int)first, (int)r, (int)step, d_Q, d_piv, d_gap, d_ok, fused ? ticket : (unsigned *)nullptr, orth_tiled());
  SPR_LAUNCH_CHECK();
  const int rows_per_block = (QR_THREADS / 64) * (64 / lpr);
  int grid = (n_cand + rows_per_block - 1) / rows_per_block;
  if (grid > 256) grid = 256;
  if (fused) {
#define FS1(L) hipLaunchKernelGGL(qr_step_fused_kernel<L>, dim3(grid), dim3(QR_THREADS), 0, st, w.cand_U, n_cand, (int)r, ldc, w.cand_idx, d_Q, d_piv, d_gap, d_ok, (int)step, 0, d_taus, w.cand_res, d_rec, d_xyz, (int)xyz_dim, n_points, d_min, partial, ticket, orth_tiled()); break
    switch (lpr) {
      case 1: FS1(1);
      case 2: FS1(2);
      case 4: FS1(4);
      case 8: FS1(8);
      case 16: FS1(16);
      case 32: FS1(32);
      default: FS1(64);
    }
#undef FS1
    SPR_LAUNCH_CHECK();
    return SPR_OK;
  }
  if (r > SPR_MAX_R) {
    int gw = (n_cand + QR_THREADS / 64 - 1) / (QR_THREADS / 64);
    if (gw > 1024) gw = 1024;
    hipLaunchKernelGGL(qr_cand_downdate_wide_kernel, dim3(gw), dim3(QR_THREADS), 0, st, w.cand_U, n_cand, (int)r, ldc,
                       w.cand_idx, d_Q + (int64_t)step * r, d_piv + step, w.cand_res, d_xyz, (int)xyz_dim, n_points, d_min);
    SPR_LAUNCH_CHECK();
    hipLaunchKernelGGL(qr_cand_best_kernel, dim3(1), dim3(1024), 0, st, (const double *)nullptr, 0, w.cand_idx,
                       w.cand_res, w.cand_U, n_cand, (int)r, ldc, (double *)nullptr, d_rec, -2.0);
    SPR_LAUNCH_CHECK();
    return SPR_OK;
  }
#define CD(L) hipLaunchKernelGGL(qr_cand_downdate_kernel<L>, dim3(grid), dim3(QR_THREADS), 0, st, w.cand_U, n_cand, (int)r, ldc, w.cand_idx, d_Q + (int64_t)step * r, d_piv + step, w.cand_res, d_xyz, (int)xyz_dim, n_points, d_min); break
  switch (lpr) {
    case 1: CD(1);
    case 2: CD(2);
    case 4: CD(4);
    case 8: CD(8);
    case 16: CD(16);
    case 32: CD(32);
    default: CD(64);
  }
#undef CD
  SPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(qr_cand_best_kernel, dim3(1), dim3(1024), 0, st, (const double *)nullptr, 0, w.cand_idx,
                     w.cand_res, w.cand_U, n_cand, (int)r, ldc, (double *)nullptr, d_rec, -2.0);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

// Single-GPU convenience: n_steps candidate-set steps (step0, step0+1, ...) in one call, the rank's own record and tau
// feeding every step -- the same launches as n_steps calls of spr_qr_step_f64 with d_recs = d_rec, n_rec = 1, first =
// (t == 0 && first_exact), without the per-call host overhead that dominates small placements.  first_exact = 0 after a
// POOL sweep: its candidates are only the best rows of the pool, so the first step is certified against tau like the rest.
extern "C" int spr_qr_steps_f64(int64_t n_rows, int32_t r, int32_t step0, int32_t n_steps, int32_t first_exact,
                                const double *d_tau, double *d_Q, int64_t *d_piv, double *d_gap, double *d_ok, double *d_rec,
                                const double *d_xyz, int32_t xyz_dim, int64_t n_points, double d_min,
                                void *d_workspace, size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(n_steps >= 1 && step0 >= 0 && step0 + n_steps <= r, SPR_E_INVALID,
              "spr_qr_steps_f64: steps [%d, %d) outside [0, %d)", step0, step0 + n_steps, r);
  static const bool fused_on = [] { const char *e = getenv("SPR_QR_FUSED_STEPS"); return !(e && e[0] == '0'); }();
  if (fused_on && r <= SPR_MAX_R) {
    // one launch per step (qr_step_fused_kernel) behind the first step's orthogonalisation
    SPR_REQUIRE(d_tau && d_Q && d_piv && d_ok && d_rec && d_workspace, SPR_E_INVALID, "spr_qr_steps_f64: NULL pointer");
    SPR_REQUIRE(!d_xyz || (xyz_dim >= 1 && xyz_dim <= 3 && n_points > 0), SPR_E_INVALID,
                "spr_qr_steps_f64: xyz needs 1..3 columns and n_points > 0");
    SPR_REQUIRE(n_rows > 0 && r > 0, SPR_E_INVALID, "spr_qr_steps_f64: bad shape");
    SPR_REQUIRE(workspace_bytes >= QrWs::bytes(r), SPR_E_WORKSPACE, "spr_qr_steps_f64: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    QrWs w(d_workspace, r);
    const int lpr = pick_lpr(r);
    const int ldc = r + (r & 1), n_cand = sweep_grid(n_rows) * QR_TOPT;
    const int rows_per_block = (QR_THREADS / 64) * (64 / lpr);
    int grid = (n_cand + rows_per_block - 1) / rows_per_block;
    if (grid > 256) grid = 256;
    // the block lists of the last sweep are dead while steps run (build_candidates has consumed them): their space
    // holds the workgroups' partial results and the ticket
    double *partial = w.tops;
    unsigned *ticket = reinterpret_cast<unsigned *>(w.tops + 4 * 256);
    hipLaunchKernelGGL(qr_orth_kernel, dim3(1), dim3(QR_THREADS), 0, st, d_rec, 1, d_tau, 1, (int)(first_exact != 0), (int)r,
                       (int)step0, d_Q, d_piv, d_gap, d_ok, ticket, orth_tiled());
    SPR_LAUNCH_CHECK();
    for (int t = 0; t < n_steps; ++t) {
      const int step = step0 + t, do_next = (t + 1 < n_steps);
#define FS(L) hipLaunchKernelGGL(qr_step_fused_kernel<L>, dim3(grid), dim3(QR_THREADS), 0, st, w.cand_U, n_cand, (int)r, ldc, w.cand_idx, d_Q, d_piv, d_gap, d_ok, step, do_next, d_tau, w.cand_res, d_rec, d_xyz, (int)xyz_dim, n_points, d_min, partial, ticket, orth_tiled()); break
      switch (lpr) {
        case 1: FS(1);
        case 2: FS(2);
        case 4: FS(4);
        case 8: FS(8);
        case 16: FS(16);
        case 32: FS(32);
        default: FS(64);
      }
#undef FS
      SPR_LAUNCH_CHECK();
    }
    return SPR_OK;
  }
  for (int t = 0; t < n_steps; ++t) {
    const int rc = spr_qr_step_f64(n_rows, r, step0 + t, d_rec, 1, d_tau, 1, t == 0 && first_exact, d_Q, d_piv, d_gap, d_ok, d_rec,
                                   d_xyz, xyz_dim, n_points, d_min, d_workspace, workspace_bytes, stream);
    if (rc != SPR_OK) return rc;
  }
  return SPR_OK;
}

extern "C" int spr_qr_exclude_f64(double *d_nrm, int64_t n_rows, int64_t row0, int64_t n_points,
                                  const uint8_t *d_mask, const double *d_xyz, int32_t xyz_dim,
                                  const int64_t *d_piv, int32_t nq, double d_min, void *stream) {
  SPR_REQUIRE(d_nrm && n_rows > 0 && row0 >= 0 && n_points > 0, SPR_E_INVALID, "spr_qr_exclude_f64: bad arguments");
  SPR_REQUIRE(nq >= 0 && nq <= QR_BATCH && (nq == 0 || (d_xyz && d_piv)), SPR_E_INVALID,
              "spr_qr_exclude_f64: nq=%d picks need xyz and piv (at most %d per call)", nq, QR_BATCH);
  SPR_REQUIRE(!d_xyz || (xyz_dim >= 1 && xyz_dim <= 3), SPR_E_INVALID, "spr_qr_exclude_f64: xyz needs 1..3 columns");
  if (!d_mask && nq == 0) return SPR_OK;
  int64_t blocks = (n_rows + QR_THREADS - 1) / QR_THREADS;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(qr_exclude_kernel, dim3((int)blocks), dim3(QR_THREADS), 0, static_cast<hipStream_t>(stream), d_nrm,
                     n_rows, row0, n_points, d_mask, d_xyz, (int)xyz_dim, d_piv, (int)nq, d_min);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

template <typename TU>
static int qr_refresh_entry(const char *who, const TU *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                            const double *d_Q, const int64_t *d_piv, int32_t j0, int32_t nq, double *d_nrm,
                            double *d_rec, double *d_tau, void *d_workspace, size_t workspace_bytes, void *stream) {
  int rc = check_ur(who, d_Ur, n_rows, r, ldu);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_Q && d_piv && d_nrm && d_rec && d_tau && d_workspace, SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(j0 >= 0 && nq >= 1 && nq <= QR_BATCH && j0 + nq <= r, SPR_E_INVALID, "%s: bad j0=%d nq=%d", who, j0, nq);
  SPR_REQUIRE(workspace_bytes >= QrWs::bytes(r), SPR_E_WORKSPACE, "%s: workspace too small", who);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = sweep_grid(n_rows);
  const int vec_ok = (r % 2 == 0) && (ldu % 2 == 0) && ((reinterpret_cast<uintptr_t>(d_Ur) & (2 * sizeof(TU) - 1)) == 0);
  QrWs w(d_workspace, r);
  hipLaunchKernelGGL(qr_mark_kernel, dim3(1), dim3(64), 0, st, d_piv + j0, (int)nq, row0, n_rows, d_nrm);
  SPR_LAUNCH_CHECK();
  rc = launch_refresh<TU, false>(grid, st, d_Ur, n_rows, (int)r, ldu, vec_ok, row0, d_Q + (int64_t)j0 * r, (int)nq, d_nrm, w.tops);
  if (rc != SPR_OK) return rc;
  return build_candidates<TU>(w, grid, d_Ur, n_rows, r, ldu, row0, d_rec, d_tau, st);
}

extern "C" int spr_qr_refresh_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                                  const double *d_Q, const int64_t *d_piv, int32_t j0, int32_t nq, double *d_nrm,
                                  double *d_rec, double *d_tau, void *d_workspace, size_t workspace_bytes,
                                  void *stream) {
  return qr_refresh_entry("spr_qr_refresh_f64", d_Ur, n_rows, r, ldu, row0, d_Q, d_piv, j0, nq, d_nrm, d_rec, d_tau,
                          d_workspace, workspace_bytes, stream);
}

extern "C" int spr_qr_refresh_u32(const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                                  const double *d_Q, const int64_t *d_piv, int32_t j0, int32_t nq, double *d_nrm,
                                  double *d_rec, double *d_tau, void *d_workspace, size_t workspace_bytes,
                                  void *stream) {
  return qr_refresh_entry("spr_qr_refresh_u32", d_Ur, n_rows, r, ldu, row0, d_Q, d_piv, j0, nq, d_nrm, d_rec, d_tau,
                          d_workspace, workspace_bytes, stream);
}

// ---- epoch sweeps (see qr_epoch_sweep_kernel) ---------------------------------------------------------------------
extern "C" int32_t spr_qr_epoch_supported(int32_t r, int64_t ldu, const void *d_Ur, int32_t u_is_f32, int64_t n_rows) {
  const size_t es = u_is_f32 ? sizeof(float) : sizeof(double);
  return r >= 16 && r <= SPR_MAX_R && r % 16 == 0 && (ldu * es) % 16 == 0 && (reinterpret_cast<uintptr_t>(d_Ur) & 15) == 0 &&
         n_rows > 0 && n_rows < INT32_MAX;
}

extern "C" int32_t spr_qr_epoch_max_directions(int32_t r) {          // directions one epoch sweep can apply
  const int ng = r / 16;
  return 16 * (ng < 4 ? ng : ng <= 6 ? 4 : 3);
}

extern "C" size_t spr_qr_pool_workspace(void) { return sizeof(int32_t) * (PB_MAX_BLOCKS + 4); }

// d_pool[0 .. *d_pool_n) <- the local rows with d_nrm_e > theta, ascending; *d_pool_n = -1 when more than cap qualify
extern "C" int spr_qr_pool_build(const double *d_nrm_e, int64_t n_rows, double theta, int32_t *d_pool, int64_t cap,
                                 int32_t *d_pool_n, void *d_workspace, size_t workspace_bytes, void *stream) {
  SPR_REQUIRE(d_nrm_e && d_pool && d_pool_n && d_workspace, SPR_E_INVALID, "spr_qr_pool_build: NULL pointer");
  SPR_REQUIRE(n_rows > 0 && n_rows < INT32_MAX && cap > 0, SPR_E_INVALID, "spr_qr_pool_build: bad shape");
  SPR_REQUIRE(workspace_bytes >= spr_qr_pool_workspace(), SPR_E_WORKSPACE, "spr_qr_pool_build: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t chunk = (n_rows + PB_MAX_BLOCKS - 1) / PB_MAX_BLOCKS;
  chunk = (chunk + PB_THREADS - 1) / PB_THREADS * PB_THREADS;
  const int blocks = (int)((n_rows + chunk - 1) / chunk);
  int32_t *counts = static_cast<int32_t *>(d_workspace);
  hipLaunchKernelGGL(qr_pool_count_kernel, dim3(blocks), dim3(PB_THREADS), 0, st, d_nrm_e, n_rows, chunk, theta, counts);
  SPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(qr_pool_scan_kernel, dim3(1), dim3(1024), 0, st, counts, blocks, cap, d_pool_n);
  SPR_LAUNCH_CHECK();
  hipLaunchKernelGGL(qr_pool_fill_kernel, dim3(blocks), dim3(PB_THREADS), 0, st, d_nrm_e, n_rows, chunk, theta, counts,
                     d_pool_n, d_pool);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

template <typename TU>
static int qr_epoch_entry(const char *who, const TU *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                          const double *d_Q, const int64_t *d_piv, int32_t j_e, int32_t j, int32_t j_mark,
                          double *d_nrm_e, double *d_nrm, const int32_t *d_pool, const int32_t *d_pool_n,
                          int64_t pool_n_host, double tau_floor, double *d_rec, double *d_tau, void *d_workspace,
                          size_t workspace_bytes, void *stream) {
  int rc = check_ur(who, d_Ur, n_rows, r, ldu);
  if (rc != SPR_OK) return rc;
  SPR_REQUIRE(d_Q && d_piv && d_nrm_e && d_nrm && d_rec && d_tau && d_workspace, SPR_E_INVALID, "%s: NULL pointer", who);
  SPR_REQUIRE(spr_qr_epoch_supported(r, ldu, d_Ur, std::is_same<TU, float>::value, n_rows), SPR_E_UNSUPPORTED,
              "%s: r=%d / alignment outside the epoch sweep's range (r a multiple of 16 up to %d, 16-byte rows)", who, r, SPR_MAX_R);
  SPR_REQUIRE(j_e >= 0 && j > j_e && j <= r && j - j_e <= spr_qr_epoch_max_directions(r) && j_mark >= 0 && j_mark <= j,
              SPR_E_INVALID, "%s: bad direction range [%d, %d) (at most %d per sweep), marks from %d", who, j_e, j,
              spr_qr_epoch_max_directions(r), j_mark);
  SPR_REQUIRE(!d_pool || (d_pool_n && pool_n_host >= 1 && pool_n_host <= n_rows), SPR_E_INVALID, "%s: bad pool", who);
  SPR_REQUIRE(workspace_bytes >= QrWs::bytes(r), SPR_E_WORKSPACE, "%s: workspace too small", who);
  hipStream_t st = static_cast<hipStream_t>(stream);
  QrWs w(d_workspace, r);
  // the picks since the last sweep leave the race for good, in the epoch norms and in the bounds
  for (int m0 = j_mark; m0 < j; m0 += 64) {
    const int nm = j - m0 < 64 ? j - m0 : 64;
    hipLaunchKernelGGL(qr_mark_kernel, dim3(1), dim3(64), 0, st, d_piv + m0, nm, row0, n_rows, d_nrm_e);
    hipLaunchKernelGGL(qr_mark_kernel, dim3(1), dim3(64), 0, st, d_piv + m0, nm, row0, n_rows, d_nrm);
  }
  SPR_LAUNCH_CHECK();
  // always the grid of a full sweep: the step kernels walk sweep_grid(n_rows) x QR_TOPT candidate slots, so the workgroups
  // a small pool leaves without rows must still write their (empty) lists over the previous sweep's
  const int grid = sweep_grid(n_rows);
  (void)pool_n_host;
  const double *Qe = d_Q + (int64_t)j_e * r;
  const int nq = j - j_e;
  // full sweeps of bases up to 64 columns run the tiles' MFMA chains side by side (template flag ILP: -11 % at one tile, -13 % at
  // two, -5 % at three, profiles/r05_epoch_sweep_ab.txt; wider bases gain nothing -- r = 128: 19.0 vs 19.1 ms, MFMA-bound at 0.82
  // of the peak -- and pool sweeps neither); SPR_QR_EPOCH_ILP=0: one tile after the other everywhere, 2: pool sweeps too (A/B)
  static int ilp_env = -1;
  if (ilp_env < 0) {
    const char *e = getenv("SPR_QR_EPOCH_ILP");
    ilp_env = e ? atoi(e) : 1;
  }
  const bool ilp_form = ilp_env && !d_pool;
  const bool ilp_pool = ilp_env == 2;                          // SPR_QR_EPOCH_ILP=2: the pool sweeps as well (A/B)
#define ES(NGV, NTV)                                                                                                     \
  do {                                                                                                                   \
    if (d_pool && ilp_pool && NGV <= 4)                                                                                  \
      hipLaunchKernelGGL((qr_epoch_sweep_kernel<NGV, NTV, TU, true, (NGV <= 4)>), dim3(grid), dim3(QR_THREADS), 0, st,    \
                         d_Ur, n_rows, (int)r, ldu, row0, Qe, nq, d_nrm_e, d_nrm, d_pool, d_pool_n, w.tops);             \
    else if (d_pool)                                                                                                     \
      hipLaunchKernelGGL((qr_epoch_sweep_kernel<NGV, NTV, TU, true>), dim3(grid), dim3(QR_THREADS), 0, st, d_Ur, n_rows,  \
                         (int)r, ldu, row0, Qe, nq, d_nrm_e, d_nrm, d_pool, d_pool_n, w.tops);                           \
    else if (ilp_form && NGV <= 4)                                                                                       \
      hipLaunchKernelGGL((qr_epoch_sweep_kernel<NGV, NTV, TU, false, (NGV <= 4)>), dim3(grid), dim3(QR_THREADS), 0, st,   \
                         d_Ur, n_rows, (int)r, ldu, row0, Qe, nq, d_nrm_e, d_nrm, d_pool, d_pool_n, w.tops);             \
    else                                                                                                                 \
      hipLaunchKernelGGL((qr_epoch_sweep_kernel<NGV, NTV, TU, false>), dim3(grid), dim3(QR_THREADS), 0, st, d_Ur, n_rows, \
                         (int)r, ldu, row0, Qe, nq, d_nrm_e, d_nrm, d_pool, d_pool_n, w.tops);                           \
  } while (0)
  switch (r / 16) {
    case 1: ES(1, 1); break;
    case 2: ES(2, 2); break;
    case 3: ES(3, 3); break;
    case 4: ES(4, 4); break;
    case 5: ES(5, 4); break;
    case 6: ES(6, 4); break;
    case 7: ES(7, 3); break;      // three tiles: 2 x (48 KB image + lists) fit a CU's LDS -- two workgroups per CU
    default: ES(8, 3); break;
  }
#undef ES
  SPR_LAUNCH_CHECK();
  return build_candidates<TU>(w, grid, d_Ur, n_rows, r, ldu, row0, d_rec, d_tau, st, tau_floor);
}

extern "C" int spr_qr_epoch_sweep_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                                      const double *d_Q, const int64_t *d_piv, int32_t j_e, int32_t j, int32_t j_mark,
                                      double *d_nrm_e, double *d_nrm, const int32_t *d_pool, const int32_t *d_pool_n,
                                      int64_t pool_n_host, double tau_floor, double *d_rec, double *d_tau,
                                      void *d_workspace, size_t workspace_bytes, void *stream) {
  return qr_epoch_entry("spr_qr_epoch_sweep_f64", d_Ur, n_rows, r, ldu, row0, d_Q, d_piv, j_e, j, j_mark, d_nrm_e, d_nrm,
                        d_pool, d_pool_n, pool_n_host, tau_floor, d_rec, d_tau, d_workspace, workspace_bytes, stream);
}

extern "C" int spr_qr_epoch_sweep_u32(const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                                      const double *d_Q, const int64_t *d_piv, int32_t j_e, int32_t j, int32_t j_mark,
                                      double *d_nrm_e, double *d_nrm, const int32_t *d_pool, const int32_t *d_pool_n,
                                      int64_t pool_n_host, double tau_floor, double *d_rec, double *d_tau,
                                      void *d_workspace, size_t workspace_bytes, void *stream) {
  return qr_epoch_entry("spr_qr_epoch_sweep_u32", d_Ur, n_rows, r, ldu, row0, d_Q, d_piv, j_e, j, j_mark, d_nrm_e, d_nrm,
                        d_pool, d_pool_n, pool_n_host, tau_floor, d_rec, d_tau, d_workspace, workspace_bytes, stream);
}
