/*
 * spr_hip.h -- C ABI of libspr_hip.so: the MI355X (gfx950) kernels under
 * openmeasure_amd.sparse_sensing.SPR.
 *
 * The reference (burn-research/OpenMEASURE, src/openmeasure/sparse_sensing.py) has no
 * FFI; its "operator interface" for this path is the set of NumPy/SciPy calls listed in
 * SURVEY.md section 2 (K1..K11).  Each entry point below replaces one of those call
 * sites and says which (file:line of sparse_sensing.py).  The Python class in
 * openmeasure_amd/rom.py (imported as openmeasure_amd.sparse_sensing) binds them with ctypes; INTEGRATION.md shows the
 * same binding as a maintainer of the reference would add it.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HBM) owned by the caller; the
 *     library allocates nothing (one exception: spr_p2p_alloc, whose interprocess handle needs
 *     the base pointer of an allocation) and keeps no state between calls;
 *   - matrices are row-major float64; "ld*" is the row stride in elements;
 *   - rows of the snapshot matrix are feature-major (global row = f*n_points + cell,
 *     sparse_sensing.py:110); a rank holds the contiguous global rows
 *     [row0, row0+n_rows) and passes the GLOBAL n_points so kernels can tell the
 *     feature of every local row;
 *   - stream is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued asynchronously on it, nothing synchronises the host;
 *   - return value: 0 = ok, <0 = error (SPR_E_*); spr_last_error() gives the text for
 *     the calling thread.  No C++ exception crosses this boundary.
 *
 * Environment switches read by the library (A/B measurements only -- every setting computes the same results through
 * another kernel of the same family; each is read once per process, at the first call that consults it):
 *   SPR_GRAM_OWN=0            spr_stats_gram_*: the generic row staging instead of the own-means lane (m >= 128)
 *   SPR_PROJECT_WS=0          spr_project_*: the general (register-resident W) kernel also where the W-stationary one fits
 *   SPR_RECONSTRUCT_DIRECT=0|1|3   spr_reconstruct_*: LDS panels everywhere | register-direct rows for f32 bases and
 *                             f64 bases wider than 96 columns (default) | register-direct everywhere
 *   SPR_QR_DIRECT=0           spr_qr_init_* / spr_qr_refresh_*: LDS-panel sweeps only
 *   SPR_QR_FUSED_STEPS=0      spr_qr_steps_f64: three launches per candidate step instead of the fused one
 *   SPR_RESERVE_CUS=k         every persistent grid is sized for k compute units fewer (multi-GPU diagnostic: leaves CUs to
 *                             RCCL's kernel, which cannot share one with the Gram / projection workgroups)
 *   SPR_QR_EPOCH_ILP=0|2      spr_qr_epoch_sweep_*: direction tiles one after the other in full sweeps too | side by side in pool
 *                             sweeps as well (default 1: full sweeps of bases up to 64 columns side by side)
 *   SPR_WS_WG_PER_CU=k        spr_project_*: workgroups per CU of the W-stationary kernel at m = 64 (1..4, default 2)
 *   SPR_QR_ORTH_TILE=0        spr_qr_step_f64 / spr_qr_steps_f64: the Gram-Schmidt passes of a step as chains of loads instead of
 *                             the register-tiled form (A/B only)
 *   SPR_P2P_BLIT=1            spr_p2p_copy / spr_field_gather_p2p: blit kernels instead of the SDMA engines
 *   SPR_P2P_PROBE=1           spr_field_gather_p2p prints the host time of each of its runtime calls (tools/p2p_push_probe.py)
 *   SPR_RCCL_LIBRARY=<path>   spr_comm_*: the RCCL library to load (default: the one already in the process, else librccl.so.1)
 * and by the Python layer (openmeasure_amd/): SPR_PROJECT_STREAM=1 (streamed-W projection for every shape),
 * SPR_GAP_FILLER=1|0 (ROM.gap_filler, opt-in: a filler launch in fit()'s host gap), SPR_GATHER=auto|p2p|rccl (field exchange of
 * sharded objects), SPR_GATHER_TRIAL=0 (no first-exchange trial of the two exchanges: 'auto' = p2p whenever available),
 * SPR_DEFER_RECONSTRUCT=0|1 (ROM.defer_reconstruct), SPR_NATIVE_COMM=1 (fit()'s all-reduce through spr_fit_gram_pass / this library's
 * own communicator), SPR_P2P_STREAMS / SPR_P2P_BUFFERS / SPR_P2P_MEMORY / SPR_P2P_JOIN_TIMEOUT_S / SPR_P2P_RELEASE_TIMEOUT_S
 * (openmeasure_amd/p2p.py), SPR_DL_KERNEL=0 (small downloads
 * by copy + event), SPR_PINNED_RESULT_GB=<g> (budget of page-locked
 * memory for host results still alive, default 8; 0 = pageable copies only), SPR_TRACE=1 (per-phase wall clock of fit(),
 * synchronising), SPR_HIP_LIBRARY=<path> (another build of this library).  None of them is needed in production.
 */
#ifndef SPR_HIP_H
#define SPR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPR_OK 0
#define SPR_E_INVALID (-1)   /* bad argument (shape, alignment, NULL)            */
#define SPR_E_UNSUPPORTED (-2) /* shape outside what the kernels are built for   */
#define SPR_E_HIP (-3)       /* a HIP runtime call failed                        */
#define SPR_E_WORKSPACE (-4) /* workspace too small                              */

#define SPR_MAX_M 256        /* snapshots (columns of X) one symmetric-Gram / register-resident projection launch handles */
#define SPR_MAX_M_WIDE 512   /* ... and the two-slice Gram path that forms the row means itself (spr_gram_cross, centre 1);
                                wider matrices go as slice pairs (spr_gram_cross_pair), any m                       */
#define SPR_MAX_R 128        /* retained modes / sensors ONE launch handles; callers go in column groups beyond      */
#define SPR_MAX_R_STREAM 256 /* ... columns ONE spr_project_stream_f64 / _x32_f64out launch takes (float64 output, 16-byte-aligned rows,
                                m % 4 == 0, no row norms): all m columns of fit()'s refinement pass at m <= 256 with X read once */
#define SPR_MAX_R_WIDE 1024  /* ... the widest basis the placement / solve kernels accept (r <= m in the reference, :336) */

/* Bumped whenever an entry point changes its argument list or meaning (round 3 -> 4: spr_qr_steps_f64 gained
 * first_exact, new entry points arrived; round 5, still 4: the CU-free field exchange spr_p2p_* / spr_field_gather_p2p*,
 * spr_field_unstage_blocks_f64; 4 -> 5 (round 6): spr_field_gather_p2p gained d_status and the poison bit, the communicator
 * entry points spr_comm_* arrived).  A binding written for another value must refuse to
 * call into this library: openmeasure_amd/_lib.py compares spr_abi_version() with the value its prototypes were
 * written for. */
#define SPR_ABI_VERSION 5
int spr_abi_version(void);
const char *spr_last_error(void);
/* number of compute units of the current device (used to size persistent grids) */
int spr_device_cus(int *out_cus);
/* Small host -> device upload executed as a kernel on `stream` (no copy-engine dependency in front of the next
 * kernel): h_pinned_src is page-locked, device-visible host memory (hipHostMalloc / torch pin_memory), n_bytes a
 * multiple of 8.  Carries the small operands the reference keeps as host ndarrays next to X -- W = V_r S_r^-1
 * (sparse_sensing.py:272-279), X_scl per feature (:115), the coefficient vectors of reconstruct (:371) -- to
 * the device.  The source may be rewritten once work queued after this call on the stream has completed. */
int spr_upload_bytes(void *d_dst, const void *h_pinned_src, int64_t n_bytes, void *stream);
/* ... and back: n_bytes (multiple of 8, <= 4 MiB) from the device into page-locked, device-visible host memory, by a kernel on
 * `stream`, which then stores ticket_value at h_pinned_ticket (a 64-bit word of such memory as well) with system-scope release:
 * the host polls the ticket instead of blocking on an event.  Carries the scaled m x m Gram matrix and the feature statistics
 * to the host eigen-solve that replaces the V factor of np.linalg.svd (sparse_sensing.py:272) -- 32 KB at BASELINE config 2. */
int spr_download_bytes(void *h_pinned_dst, const void *d_src, int64_t n_bytes, void *h_pinned_ticket, uint64_t ticket_value,
                       void *stream);
/* HOST-side helper (host pointers, no device work): unit eigenvectors of the symmetric tridiagonal matrix (h_d[m], h_e[m-1]) for
 * the r eigenvalues h_lam, all at once -- the inverse iterations of LAPACK's dstein (dlagtf / dlagts) with the eigenvalue index
 * as the vectorised dimension, without dstein's re-orthogonalisation inside clusters (check the result; fit() falls back to
 * dstein).  h_Z: m x r row-major, column j belongs to h_lam[j].  Part of the host eigen-solve of the m x m Gram matrix between
 * the two passes of fit() -- the reference's np.linalg.svd call site (sparse_sensing.py:272), of which :336 keeps r vectors. */
int spr_host_tridiag_vectors(const double *h_d, const double *h_e, int32_t m, const double *h_lam, int32_t r, double *h_Z,
                             int32_t iterations);
/* ... and the whole route in one host call: h_G (m x m, symmetric) -> h_lam[m] (all eigenvalues, DESCENDING) and h_V (m x r
 * row-major: the unit eigenvectors of the r largest, column j belonging to h_lam[j]).  fn_dsytrd / fn_dsterf / fn_dormtr: the
 * addresses of the caller's LAPACK routines with the Fortran calling convention without hidden string lengths, as SciPy exports
 * them (scipy.linalg.cython_lapack.__pyx_capi__) -- this library links no LAPACK.  Returns 0; 1 when LAPACK reports a failure;
 * 2 when the vectors miss their orthonormality checks (close eigenvalues): the caller then takes dstein / dsyevd.  < 0: SPR_E_*. */
int spr_host_eig_top(const double *h_G, int32_t m, int32_t r, double *h_lam, double *h_V, void *fn_dsytrd, void *fn_dsterf,
                     void *fn_dormtr);
/* The same idea for the SVD that ends fit()'s conditioning refinement pass (np.linalg.svd of an m x m factor, openmeasure_amd/rom.py
 * _refine_spectrum; reference: the accuracy of np.linalg.svd(X0), :272): ALL singular values (h_S, descending) and the r leading
 * RIGHT singular vectors (h_V, m x r row-major) of the row-major m x m matrix h_M -- dgebrd, dbdsdc (values only), the batched
 * inverse iteration on the Golub-Kahan form of the bidiagonal matrix, dormbr; LAPACK again through the caller's function
 * pointers.  0 ok, 1 LAPACK failed, 2 the vectors failed their orthonormality checks (take dgesdd). */
int spr_host_svd_top(const double *h_M, int32_t m, int32_t r, double *h_S, double *h_V, void *fn_dgebrd, void *fn_dbdsdc,
                     void *fn_dormbr);

/* ---- K1 + K3a : fused row mean, per-feature statistics, per-feature Gram -----------
 * Replaces np.average(x, axis=1) (:112), np.std(x) (:115), the materialised
 * X0 = (X - X_cnt)/X_scl (:169) and the X0^T X0 half of np.linalg.svd (:272).
 * Two calls on the same stream and workspace:
 *   spr_stats_gram_f64           the one read of X: d_rowmean[i] = mean_j X[i,j] for every
 *                                local row, per-workgroup Gram / Welford partials -> workspace;
 *   spr_stats_gram_finalize_f64  fixed-order reduction of the partials (bitwise reproducible,
 *                                no float atomics).  For every feature f over its LOCAL rows:
 *     d_fstats[3f..3f+2] = (count, mean, M2) of the row means (Welford/Chan form),
 *     d_gram[f*m*m ..]   = sum_i (x_i - mean_i)(x_i - mean_i)^T          (m x m, full).
 * The caller combines: var_f = (trace(G_f) + m*M2_f) / (count_f*m) -> X_scl; G = sum_f G_f/var_f.
 * Across ranks d_gram is summed (all-reduce) and d_fstats Chan-merged.
 * center = 1: as above.  center = 0: rows are taken as they are (d_rowmean is written as
 * zeros) -- the plain X^T X needed by ROM.decomposition(X0) on a caller-supplied X0 (:272).
 * Workspace: spr_stats_gram_workspace() bytes, 16-byte aligned.  m <= SPR_MAX_M. */
size_t spr_stats_gram_workspace(int32_t m, int32_t n_features);
int spr_stats_gram_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx,
                       int64_t row0, int64_t n_points, int32_t n_features, int32_t center,
                       double *d_rowmean, void *d_workspace, size_t workspace_bytes, void *stream);
int spr_stats_gram_finalize_f64(int64_t n_rows, int32_t m, int64_t row0, int64_t n_points,
                                int32_t n_features, const void *d_workspace, size_t workspace_bytes,
                                double *d_fstats, double *d_gram, int32_t ldg, int32_t origin,
                                void *stream);
/* ldg / origin: the m x m block is written at (origin, origin) of per-feature ldg x ldg matrices
 * (ldg = m, origin = 0 for a stand-alone Gram).
 *
 * 256 < m <= 512 (column-split path, csrc/gram_wide.hip): columns A = [0,256), B = [256,m).
 *   spr_gram_cross_f64    A^T B and its transpose into ldg = m matrices.  center = 1: the means of the FULL rows are
 *                         formed in this pass (the panels hold whole rows) and WRITTEN to d_rowmean; run it first,
 *                         then spr_rowmean_stats_f64 for the per-feature statistics (count, mean, M2 of the row
 *                         means: reads the n means, not X).  center = 2: means READ from d_rowmean; 0: none;
 *   spr_rowstats_f64      (alternative to center = 1 above) row means of the full rows + per-feature statistics in a
 *                         separate read of X;
 *   spr_stats_gram_f64    on the slices (d_X, 256) and (d_X + 256, m - 256), ldx = full stride,
 *                         center = 2: the row means are READ from d_rowmean (no statistics; give
 *                         the finalize call a scratch d_fstats), finalize with ldg = m and
 *                         origin = 0 / 256. */
size_t spr_rowstats_workspace(int32_t n_features);
int spr_rowstats_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                     int64_t n_points, int32_t n_features, double *d_rowmean, double *d_fstats,
                     void *d_workspace, size_t workspace_bytes, void *stream);
size_t spr_gram_cross_workspace(int32_t m, int32_t n_features);
int spr_gram_cross_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                       int64_t n_points, int32_t n_features, int32_t center, double *d_rowmean,
                       double *d_gram, void *d_workspace, size_t workspace_bytes, void *stream);
int spr_rowmean_stats_f64(const double *d_rowmean, int64_t n_rows, int64_t row0, int64_t n_points,
                          int32_t n_features, double *d_fstats, void *d_workspace, size_t workspace_bytes,
                          void *stream);   /* workspace: spr_rowstats_workspace(n_features) */
/* m > 512 (any snapshot count, :272 accepts any X0): the columns are cut into slices of 256 (the last one 1..256 wide);
 *   spr_rowstats_f64 first (row means of the full rows + feature statistics: one read of X), then, all with the row
 *   means READ from d_rowmean (centre mode 2; 0 for rows taken as they are),
 *   spr_stats_gram_f64 + finalize (ldg = m, origin = 256 i) on every slice i -> diagonal blocks,
 *   spr_gram_cross_pair_f64 on every pair i < j: A = the 256 columns at col_a = 256 i, B = the w_b columns at
 *   col_b = 256 j; block (col_a.., col_b..) and its mirror are written into the per-feature ldg x ldg matrices.
 *   Workspace of a pair: spr_gram_cross_workspace(256 + w_b, n_features). */
int spr_gram_cross_pair_f64(const double *d_X, int64_t n_rows, int32_t col_a, int32_t col_b, int32_t w_b,
                            int32_t ldg, int64_t ldx, int64_t row0, int64_t n_points, int32_t n_features,
                            int32_t center, double *d_rowmean, double *d_gram, void *d_workspace,
                            size_t workspace_bytes, void *stream);

/* ---- K3b : device-side spectrum for m <= spr_spectrum_max_m() (= 64) ------------------------
 * Replaces, for small snapshot counts, the host eigen-solve behind np.linalg.svd (:272), the block
 * statistics merge (np.std :115 and the other scale_type formulas :117-161 that derive from block
 * mean / variance), A = (diag(S) Vt)^T (:273) and the explained variance (:274-275), so that fit()
 * needs no host synchronisation.  d_gram [F][m][m] and d_fstats_all [n_ranks][F][3] are the
 * (all-reduced / all-gathered) outputs of spr_stats_gram_finalize_f64.  scale_code: 0 'std',
 * 1 'none', 2 'pareto', 3 'vast', 4 'level', 5 'variance', 6 'poisson', 7 'l2-norm'.
 * Outputs: d_feat [F][5] = (count, block mean, block variance, scale, variance of the row-centred values =
 * trace(G_f) / (count m)); d_scale, d_inv_scale
 * [F]; d_lam, d_S, d_expvar [m] (descending); d_V [m][m] (columns = eigenvectors, largest-magnitude
 * entry positive); d_W = V_r S_r^-1 and d_Ar = V_r S_r [m][r]; d_info = (Jacobi sweeps, off^2, diag^2). */
int32_t spr_spectrum_max_m(void);
int spr_spectrum_f64(const double *d_gram, const double *d_fstats_all, int32_t n_ranks,
                     int32_t n_features, int32_t m, int32_t scale_code, int32_t r, double *d_feat,
                     double *d_scale, double *d_inv_scale, double *d_lam, double *d_S, double *d_expvar,
                     double *d_V, double *d_W, double *d_Ar, double *d_info, void *stream);

/* ---- K3a' : per-feature Gram blocks -> the scaled m x m Gram matrix --------------------
 * G = sum_f G_f / X_scl_f^2 is the Gram matrix of the reference's X0 = (X - X_cnt)/X_scl (:169) whose SVD :272 takes.
 * Inputs as for spr_spectrum_f64 (all-reduced d_gram [F][m][m], all-gathered d_fstats_all [n_ranks][F][3], scale_code
 * 0..7); outputs d_G [m][m], d_feat [F][5] = (count, block mean, block variance (:115), scale, variance of the
 * row-centred values), and the scales
 * d_scale / d_inv_scale [F] that the projection and reconstruction kernels read.  Replaces the host merge of fit():
 * one download of m^2 + 5 F doubles instead of F m^2, no upload. */
int spr_gram_combine_f64(const double *d_gram, const double *d_fstats_all, int32_t n_ranks,
                         int32_t n_features, int32_t m, int32_t scale_code, double *d_G, double *d_feat,
                         double *d_scale, double *d_inv_scale, void *stream);

/* ---- K1 + K3a for 256 < m <= 512 without a pass for the full-row means (np.average :112 + Gram half of :272) ----
 * The Gram matrix of row-centred data is P G_s P, P = I - 1 1^T / m, for the Gram matrix G_s of rows shifted by any
 * per-row constant.  The three launches of the wide path therefore all shift by the mean of the FIRST 256 columns, which
 * the first symmetric launch (spr_stats_gram_f64 on that slice, center = 1) forms anyway:
 *   spr_stats_gram_shifted_*   the symmetric pass on another column slice with the given shifts (centre mode 2) that also
 *                              writes the RAW row sums of its slice (d_rowsum[n_rows]);
 *   spr_gram_cross_*           center = 2 with the same shifts (no row sums);
 *   spr_gram_shift_finish_f64  d_rowmean <- (w_a d_rowmean + d_rowsum_b) / m (the means of the full rows) and
 *                              G_f <- P G_f P for the F matrices. */
int spr_stats_gram_shifted_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                               int64_t n_points, int32_t n_features, const double *d_shift, double *d_rowsum,
                               void *d_workspace, size_t workspace_bytes, void *stream);
int spr_gram_shift_finish_f64(double *d_rowmean, const double *d_rowsum_b, int64_t n_rows, int32_t w_a, int32_t m,
                              double *d_gram, int32_t n_features, void *stream);

/* ---- K4 : basis projection  Ur = X0 . W,  W = V_r Sigma_r^-1 (m x r) ----------------
 * Replaces the U factor of np.linalg.svd (:272) and the truncation U[:, :r] (:336).
 * Second read of X.  center = 1: d_rowmean (the row means written by spr_stats_gram_f64)
 * is removed in the epilogue as x.W - mean*(1^T W); the per-feature 1/X_scl
 * (d_inv_scale[n_features]) is applied there too, so X0 never exists.  center = 0:
 * rows are used as they are (d_rowmean may be NULL).  accumulate = 1 adds to d_Ur instead of
 * overwriting it: a wider X (m <= 512) is projected as two column slices, (d_X, d_W) and
 * (d_X + 256, d_W + 256 r), the second one accumulating. */
int spr_project_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx,
                    int64_t row0, int64_t n_points, int32_t n_features, int32_t center,
                    const double *d_inv_scale, const double *d_rowmean, const double *d_W, int32_t r,
                    double *d_Ur, int64_t ldu, int32_t accumulate, void *stream);

/* ---- K4s : the same projection for ANY snapshot count m (csrc/project_stream.hip) -----------------------------
 * Replaces the U factor of np.linalg.svd (:272) / U[:, :r] (:336) when W = V_r Sigma_r^-1 no longer fits a workgroup's
 * registers or LDS (m > 256; BASELINE config 5 has m = 512, r = 128).  ONE launch over the full contraction length:
 * W streams through LDS in k-chunks from a permuted image the call builds in the workspace, the accumulators live
 * across all chunks and d_Ur is written once (rounded once when it is float).  m is a run-time loop count -- no upper
 * bound; r <= SPR_MAX_R per call (the caller projects wider bases in column groups: d_W = the group's columns packed
 * m x r, d_Ur + column offset, same ldu).  center: 0 rows as they are, 1 row means removed in the epilogue
 * (x.W - mean (1^T W)), 2 row means subtracted from the operand before the multiplication (data whose mean dwarfs its
 * fluctuation).  Workspace: spr_project_stream_workspace(m, r, x_is_f32) bytes, 16-byte aligned. */
size_t spr_project_stream_workspace(int32_t m, int32_t r, int32_t x_is_f32);
int spr_project_stream_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                           int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                           const double *d_rowmean, const double *d_W, int32_t r, double *d_Ur, int64_t ldu,
                           void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- K4 + the first sweep of K6 : projection that also leaves the squared row norms of the basis ---------------
 * optimal_placement (:739) starts from |Ur[i, :]|^2 of every row; the plain route reads all of Ur once more for them
 * (spr_qr_init_*).  These variants write d_rownorm2[n_rows] -- the norms of the values AS STORED, i.e. rounded to the
 * basis type first -- from the accumulators of the projection that stores d_Ur, and spr_qr_init_norms_* starts the
 * pivoting from that vector (8 bytes per row instead of r values).  Same arguments and results otherwise.
 * spr_project_norms_*        only the W-stationary form produces norms (m = 64 / 128 / 192 / 256 packed rows, 16-byte aligned,
 *                            r <= 64, n_rows >= 4096): spr_project_norms_supported() says whether a shape qualifies;
 *                            SPR_E_UNSUPPORTED and nothing launched otherwise.
 * spr_project_stream_norms_* every shape spr_project_stream_* takes (any m, r <= SPR_MAX_R per call). */
int32_t spr_project_norms_supported(int32_t m, int32_t r, int64_t n_rows, int64_t ldx, const void *d_X,
                                    int32_t x_is_f32);
int spr_project_norms_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                          int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                          const double *d_rowmean, const double *d_W, int32_t r, double *d_Ur, int64_t ldu,
                          double *d_rownorm2, void *stream);
int spr_project_stream_norms_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                 int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                                 const double *d_rowmean, const double *d_W, int32_t r, double *d_Ur, int64_t ldu,
                                 double *d_rownorm2, void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- K2 / K11 as stand-alone calls (ROM.scale_data's return value, ROM.unscale_data) --
 * spr_scale_rows:  X0 = (X - rowmean) * inv_scale[feature]   (:169), n_rows x m.
 * spr_unscale:     x  = scale[feature] * x0 + rowmean        (:235), n_rows; with d_rowscale != NULL
 *                  the per-row factor is used instead (the sampled form, :233). */
int spr_scale_rows_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                       int64_t n_points, int32_t n_features, const double *d_rowmean,
                       const double *d_inv_scale, double *d_X0, int64_t ldo, void *stream);
int spr_unscale_f64(const double *d_x0, int64_t n_rows, int64_t row0, int64_t n_points,
                    int32_t n_features, const double *d_rowmean, const double *d_scale,
                    const double *d_rowscale, double *d_x, void *stream);

/* ---- per-feature min / max of the raw block: np.max(x), np.min(x) of scale_type 'range' (:128)
 * and 'max' (:135).  d_minmax[2f] = min, [2f+1] = max over the LOCAL rows of feature f (+inf/-inf
 * when a feature has no local rows; ranks combine with min / max).  One extra read of X. */
size_t spr_feature_minmax_workspace(int32_t n_features);
int spr_feature_minmax_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                           int64_t n_points, int32_t n_features, double *d_minmax,
                           void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- per-feature order statistics of the raw block: np.median(x) of scale_type 'median' (:140-141).
 * Radix selection on order-preserving keys (key(x) = bits(x) ^ sign for x >= 0, ~bits(x) for x < 0): one call
 * counts, for every feature f with local rows and for two targets t (lower / upper middle element), the keys
 * whose bits above position shift+bits equal those of d_prefix[2f+t], by the `bits`-wide digit at position
 * `shift` (1 <= bits <= 13; shift + bits == 64 counts every key).  Counts are ADDED to
 * d_hist[f][t][1 << bits] (caller zeroes it; ranks sum them).  two_targets = 0 when both prefixes are equal
 * (one histogram is built and written to both halves).  Five calls (13+13+13+13+12 bits) pin down both keys;
 * the caller walks the cumulative counts between calls.  One read of X per call. */
int spr_feature_digit_hist_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                               int64_t n_points, int32_t n_features, const uint64_t *d_prefix, int32_t shift,
                               int32_t bits, int32_t two_targets, uint64_t *d_hist, void *stream);

/* ---- axis_cnt=None (np.average(x, axis=None), :112): scalar centre per feature ----------------
 * spr_colsums_f64: d_out[f][0][m] = sum_i (x_i - mean_i), d_out[f][1][m] = sum_i mean_i (x_i - mean_i)
 * over the LOCAL rows of feature f -- the two vectors that turn the row-centred Gram blocks into the
 * Gram blocks of (X - mu_f) (csrc/scale.hip).  One extra read of X, only for this option.
 * spr_fill_feature_f64: d_out[i] = d_values[feature of row i] (the new X_cnt column). */
size_t spr_colsums_workspace(int32_t m, int32_t n_features);
int spr_colsums_f64(const double *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                    int64_t n_points, int32_t n_features, const double *d_rowmean, double *d_out,
                    void *d_workspace, size_t workspace_bytes, void *stream);
int spr_fill_feature_f64(double *d_out, int64_t n_rows, int64_t row0, int64_t n_points, int32_t n_features,
                         const double *d_values, void *stream);

/* ---- K10 + K11 : reconstruction  x = X_scl * (Ur a) + X_cnt -------------------------
 * Replaces Ur @ Ar.T (:371) and unscale_data (:235, :372-373) in one streaming pass.
 * d_A is n_p x r row-major (the Ar argument); output d_Xrec is COLUMN-major
 * (n_p columns of ldo >= n_rows doubles each). d_scale[n_features] = X_scl per feature;
 * d_rowscale (optional) = one factor per row instead, for reconstruct(sampling=S) (:365-368)
 * where the rows are S.Ur, the centre S.X_cnt and the factor S.X_scl. */
int spr_reconstruct_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu,
                        int64_t row0, int64_t n_points, int32_t n_features,
                        const double *d_rowmean, const double *d_scale, const double *d_rowscale,
                        const double *d_A, int32_t n_p, double *d_Xrec, int64_t ldo,
                        void *stream);
/* Sharded reconstruct() with n_p > 1 coefficient vectors: the one all-gather of the ranks' (n_p, n_loc) result blocks
 * leaves d_stage[world][n_p][n_loc]; this copies it into the layout the reference returns (:371-375: the vectors as columns
 * of the WHOLE field), d_out[v * ldo + q * n_loc + i] = d_stage[q][v][i].  (One vector needs nothing: the staged blocks are
 * the field.) */
int spr_field_unstage_f64(const double *d_stage, int32_t world, int32_t n_p, int64_t n_loc, double *d_out, int64_t ldo,
                          void *stream);
/* ... and for row blocks of DIFFERENT sizes (the gather then carries blocks padded to n_max rows):
 * d_out[v * ldo + off_q + i] = d_stage[q][v][i] for i < rows_q, with (off_q, rows_q) = d_layout[2q], d_layout[2q+1]
 * (int64, on the device; off_q = first global row of rank q's block minus that of rank 0's). */
int spr_field_unstage_blocks_f64(const double *d_stage, int32_t world, int32_t n_p, int64_t n_max, const int64_t *d_layout,
                                 double *d_out, int64_t ldo, void *stream);

/* ---- the same exchange WITHOUT compute units (csrc/p2p.hip, round 5) -------------------------------------------------
 * RCCL's all-gather runs a device kernel that cannot share a compute unit with the Gram / projection workgroups, so a field
 * gather left in flight under the next fit() only progresses where a CU is free.  On one node the ranks can instead map each
 * other's copy of the field (interprocess handles) and WRITE their block into it with the SDMA engines.
 *   spr_p2p_alloc / spr_p2p_free   a device buffer of n_bytes (multiple of 4096) and its interprocess handle
 *                                  (spr_p2p_handle_bytes() bytes at h_handle).  kind 0: hipMalloc (coarse-grained: coherent
 *                                  between GPUs at kernel boundaries); 1: fine-grained (coherent at system scope while
 *                                  kernels run: the counters); 2: uncached.  The ONE allocation this library makes: a
 *                                  handle can only be taken of the base pointer of an allocation.
 *   spr_p2p_open / spr_p2p_close   map / unmap a buffer exported by ANOTHER process of this node (peer access is enabled
 *                                  lazily); the handle bytes travel through any channel the caller has (torch.distributed).
 *   spr_p2p_device_id / _peer_access   the PCI bus id of the current device (>= 16 bytes at h_buf), and whether the current
 *                                  device IS the one with a given bus id or may access its memory (*h_can = 1 / 0; 0 also
 *                                  for a device this process cannot see).  Exchanged next to the handles: a write into
 *                                  mapped memory of a GPU without peer access is a memory fault, not an error code.
 *   spr_p2p_signal / spr_p2p_wait  hipStreamWriteValue64 / hipStreamWaitValue64(>=) on a 64-bit counter inside such a
 *                                  buffer (own or mapped); used on the copy streams.
 *   spr_p2p_flags_set / _wait      ONE single-wave kernel that raises / awaits n <= 128 counters (host table of device
 *                                  pointers, passed to the kernel by value): what the compute stream uses.  The wait gives
 *                                  up after timeout_s seconds of the device's wall clock -- or at once on a counter that
 *                                  carries the poison bit, spr_p2p_poison_bit() = 1 << 62: "this exchange is broken" -- and
 *                                  leaves (index + 1, value seen) in the two words at d_status (NULL: nowhere) -- every
 *                                  wave reaches its exit.
 *   spr_p2p_copy                   one device-to-device copy through the SDMA engines (hipMemcpyDeviceToDeviceNoCU).
 *   spr_field_gather_p2p           this rank's block of the field -- columns [first, first + n_loc) of the n_p rows of its
 *                                  own (n_p, ldo) copy d_field -- into the same place of every peer's copy: per peer p, on
 *                                  streams[p] (streams may repeat; the peers of one stream are served together: one wait
 *                                  kernel, their copies, one kernel for their counters): wait until *d_release_flag[p] >=
 *                                  release_value (a counter in THIS rank's memory that peer p raises when it no longer reads
 *                                  what its copy held; 0 = no wait; a single-wave kernel that gives up after
 *                                  release_timeout_s), n_p copies of n_loc doubles, *d_peer_arrive_flag[p] = arrive_value (a
 *                                  counter in peer p's memory), and, if given, *d_pushed_flag[p] = arrive_value (this rank's
 *                                  own: "the push to p has left").  A release wait that gives up leaves (p + 1, value seen) in
 *                                  the two words at d_status (page-locked host or device memory the kernels can reach; NULL:
 *                                  nowhere) and -- an SDMA copy cannot be taken back -- every arrival counter raised while
 *                                  *d_status != 0 carries the poison bit: the peer's join of this gather and this rank's own
 *                                  both fail, nobody is handed a field that was overwritten while it may still have been read.
 *                                  The caller orders streams[p] behind the kernel that wrote the block -- with events, or
 *                                  (d_ready_flag != NULL) with a counter of its own that it raises to ready_value on its
 *                                  compute stream behind that kernel (spr_p2p_flags_set: system-scope release): the copy
 *                                  streams' wait kernels then await it too (index n_peers in the status words if it never
 *                                  comes) and no event of the compute stream is needed.
 *   spr_field_gather_p2p_join      `stream` waits (one kernel) until every one of the n counters -- the peers' arrivals and
 *                                  this rank's own pushed counters -- has reached arrive_value.
 *   spr_field_gather_p2p_release   *d_peer_release_flag[p] = value for every peer (one kernel), behind everything enqueued
 *                                  on `stream` so far (the consumers of the previous field).
 * All pointer tables are HOST arrays of device pointers.  Replaces the remote half of Ur @ Ar.T being whole on every caller
 * (sparse_sensing.py:371-375) next to torch.distributed's all_gather; SPR_P2P_BLIT=1 swaps the SDMA copies for blit kernels
 * (A/B only). */
size_t spr_p2p_handle_bytes(void);
int spr_p2p_alloc(size_t n_bytes, int32_t kind, void **d_ptr, void *h_handle);
int spr_p2p_free(void *d_ptr);
int spr_p2p_open(const void *h_handle, void **d_mapped);
int spr_p2p_close(void *d_mapped);
int spr_p2p_device_id(char *h_buf, int32_t n_buf);
int spr_p2p_peer_access(const char *h_bus_id, int32_t *h_can);
int spr_p2p_signal(void *d_flag, uint64_t value, void *stream);
int spr_p2p_wait(void *d_flag, uint64_t value, void *stream);
int spr_p2p_flags_set(void *const *d_flags, int32_t n, uint64_t value, void *stream);
int spr_p2p_flags_wait(void *const *d_flags, int32_t n, uint64_t value, double timeout_s, void *d_status, void *stream);
int spr_p2p_copy(void *d_dst, const void *d_src, int64_t n_bytes, void *stream);
uint64_t spr_p2p_poison_bit(void);
int spr_field_gather_p2p(const double *d_field, int64_t ldo, int32_t n_p, int64_t first, int64_t n_loc, int32_t n_peers,
                         void *const *d_peer_field, void *const *d_release_flag, uint64_t release_value,
                         double release_timeout_s, void *const *d_peer_arrive_flag, uint64_t arrive_value,
                         void *const *d_pushed_flag, void *const *streams, void *d_status, void *d_ready_flag,
                         uint64_t ready_value);
int spr_field_gather_p2p_join(void *const *d_flags, int32_t n_flags, uint64_t arrive_value, double timeout_s, void *d_status,
                              void *stream);
int spr_field_gather_p2p_release(void *const *d_peer_release_flag, int32_t n_peers, uint64_t value, void *stream);

/* ---- the collectives of the sharded path behind this ABI (csrc/comm.hip, round 6) -------------------------------------------
 * north_star: "a single RCCL all-reduce over xGMI for the Gram matrix and a final all-gather for the reconstructed field"; the
 * reference has neither (one process, NumPy).  RCCL is reached through dlopen -- the copy that is already in the process
 * (PyTorch's), else librccl.so.1 of the ROCm installation, else SPR_RCCL_LIBRARY=<path> -- so this library links nothing but the
 * HIP runtime; where no RCCL can be loaded these calls return SPR_E_UNSUPPORTED with the loader's text.
 *   spr_comm_unique_id   (one rank) fills spr_comm_unique_id_bytes() bytes; the caller carries them to the other ranks through
 *                        any channel it has (a file, MPI, torch.distributed);
 *   spr_comm_init        COLLECTIVE over the `world` callers: a communicator on the CURRENT device; spr_comm_destroy frees it;
 *   spr_allreduce_f64    in-place sum of `count` doubles over the ranks, enqueued on `stream` (_i64: 64-bit integers -- the
 *                        digit histograms of the median scaling, :140-141);
 *   spr_allgather        rank q's bytes_per_rank bytes land at d_recv + q * bytes_per_rank on every rank (the field of
 *                        reconstruct(), sparse_sensing.py:371-375: whole on every caller);
 *   spr_fit_gram_pass    the first pass of fit() WITH its collective as one enqueue (SURVEY 8(b) "spr_fit_stats_gram (includes
 *                        collectives)"): spr_stats_gram + finalize into the rank's slots of d_buf = [F m m Gram | world x F x 3
 *                        statistics | world first rows] (spr_fit_gram_pass_buffer() bytes; zeroed here), the all-reduce of d_buf
 *                        over `comm` (NULL: one rank, no collective), spr_gram_combine_f64 -> d_G [m][m], d_feat [F][5],
 *                        d_scale / d_inv_scale [F].  Replaces np.average / np.std / X0 = (X - cnt)/scl / the X0^T X0 half of
 *                        np.linalg.svd (:112, :115, :169, :272) for a row block of a sharded X.  m <= SPR_MAX_M_WIDE (beyond 256 the
 *                        column-split path with its three launches, BASELINE config 5's m = 512); workspace:
 *                        spr_fit_gram_pass_workspace(m, n_features, n_rows) bytes, 256-byte aligned; scale_code as
 *                        for spr_gram_combine_f64; x_is_f32: d_X is float (storage only).  Afterwards d_buf still holds what
 *                        every rank contributed: rows per (rank, feature) = d_buf[F m m + (q F + f) 3], first rows at the end. */
size_t spr_comm_unique_id_bytes(void);
int spr_comm_unique_id(void *h_id);
int spr_comm_init(const void *h_id, int32_t rank, int32_t world, void **comm);
int spr_comm_destroy(void *comm);
int spr_comm_info(void *comm, int32_t *rank, int32_t *world);
const char *spr_comm_library(void);
int spr_allreduce_f64(void *comm, double *d_buf, int64_t count, void *stream);
int spr_allreduce_i64(void *comm, int64_t *d_buf, int64_t count, void *stream);
int spr_allgather(void *comm, const void *d_send, void *d_recv, int64_t bytes_per_rank, void *stream);
size_t spr_fit_gram_pass_buffer(int32_t m, int32_t n_features, int32_t world);
size_t spr_fit_gram_pass_workspace(int32_t m, int32_t n_features, int64_t n_rows);
int spr_fit_gram_pass(void *comm, const void *d_X, int32_t x_is_f32, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                      int64_t n_points, int32_t n_features, int32_t scale_code, double *d_rowmean, double *d_buf,
                      size_t buf_bytes, double *d_G, double *d_feat, double *d_scale, double *d_inv_scale, void *d_workspace,
                      size_t workspace_bytes, void *stream);

/* ---- K6 : QR column pivoting of Ur^T (sensor selection) -----------------------------
 * Replaces scipy.linalg.qr(Ur.T, pivoting=True) (:739) -- only the first s pivots are
 * used by the reference (:740-743).  Greedy max-residual-norm selection with norm
 * down-dating, identical in exact arithmetic to dgeqp3's choice; ties go to the lowest
 * global row index (LAPACK idamax).  Candidate-set form (csrc/qr_pivot.hip): a full sweep
 * over Ur leaves exact residual norms, a candidate set (the largest rows of every sweep
 * block) and tau = the largest norm a non-candidate can have; steps run on the candidates
 * and are certified while the winner's residual is strictly above tau.
 *
 * spr_mask_rows     Ur[~mask,:] = 0 in place (:737-738); d_mask is n_rows bytes.
 * spr_qr_init       d_nrm[i] = |Ur[i,:]|^2, candidate set, this rank's best record
 *                   d_rec[0..r+2] = (value, global row as double, runner-up, row of Ur) and
 *                   d_tau[0].
 * spr_qr_step       given the ranks' records d_recs[n_rec][r+3] and taus d_taus[n_tau]: picks
 *                   the winner, writes d_piv[step], the direction d_Q[step][r], d_gap[step]
 *                   (relative gap to the best rival candidate) and d_ok[step] (1 = certified:
 *                   first step after a sweep, or winner > tau), down-dates the candidate
 *                   residuals and leaves this rank's next record in d_rec.
 * spr_qr_refresh    applies the nq <= spr_qr_batch() accepted directions Q[j0..j0+nq) to all
 *                   rows in one sweep (pivots marked), then new candidates / d_rec / d_tau.
 * Driver loop (host): up to spr_qr_batch() steps, read d_ok once, keep the certified prefix
 * (always >= 1 step), refresh, repeat.  Single GPU: d_recs == d_rec, d_taus == d_tau.
 * Workspace: spr_qr_workspace(n_rows) bytes, the same buffer for all calls of one run.
 *
 * calc_type='gem' (SPR.gem, :586-698) runs on the same machinery: the conditional variance of a row given the
 * rows already picked is the squared residual of the row, centred over its r entries, after projecting out the
 * centred picks -- i.e. the pivoting above with the direction 1/sqrt(r) applied first (host puts it in Q[0],
 * piv[0] = -1, one refresh) and the steps numbered from 1.  Two extras:
 * spr_qr_exclude     d_nrm[i] = -1 (out of the pool for good) for rows with d_mask[i] == 0 (search mask, :617)
 *                    and for rows whose position d_xyz[(row0+i) % n_points][xyz_dim] lies closer than d_min to
 *                    the position of one of the picks d_piv[0..nq) (:649-652); call it before the refresh that
 *                    applies those picks.  d_mask / d_xyz may be NULL.
 * spr_qr_step        with d_xyz != NULL also drops the candidates closer than d_min to the step's pick. */
#define SPR_QR_REC_LEN(r) ((r) + 3)
size_t spr_qr_workspace(int64_t n_rows);                 /* r <= SPR_MAX_R */
size_t spr_qr_workspace_r(int64_t n_rows, int32_t r);    /* any r <= SPR_MAX_R_WIDE (0 beyond) */
int32_t spr_qr_batch(void);
int spr_mask_rows_f64(double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu,
                      const uint8_t *d_mask, void *stream);
int spr_qr_init_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu,
                    int64_t row0, double *d_nrm, double *d_rec, double *d_tau,
                    void *d_workspace, size_t workspace_bytes, void *stream);
/* spr_qr_init with the squared row norms given (d_nrm0, written by spr_project_norms_* / spr_project_stream_norms_* for this
 * very d_Ur; not modified): no pass over the basis.  d_nrm is the working vector the steps down-date. */
int spr_qr_init_norms_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                          const double *d_nrm0, double *d_nrm, double *d_rec, double *d_tau,
                          void *d_workspace, size_t workspace_bytes, void *stream);
int spr_qr_step_f64(int64_t n_rows, int32_t r, int32_t step, const double *d_recs, int32_t n_rec,
                    const double *d_taus, int32_t n_tau, int32_t first, double *d_Q,
                    int64_t *d_piv, double *d_gap, double *d_ok, double *d_rec,
                    const double *d_xyz, int32_t xyz_dim, int64_t n_points, double d_min,
                    void *d_workspace, size_t workspace_bytes, void *stream);
/* single GPU: n_steps consecutive steps in one call (d_recs = d_rec, n_rec = 1, first = first_exact for the call's
 * first step: 1 after spr_qr_init_* / spr_qr_refresh_* / a full epoch sweep, 0 after a pool sweep) */
int spr_qr_steps_f64(int64_t n_rows, int32_t r, int32_t step0, int32_t n_steps, int32_t first_exact,
                     const double *d_tau, double *d_Q, int64_t *d_piv, double *d_gap, double *d_ok, double *d_rec,
                     const double *d_xyz, int32_t xyz_dim, int64_t n_points, double d_min,
                     void *d_workspace, size_t workspace_bytes, void *stream);
/* ---- K6, epoch sweeps: refreshes that only visit the rows that can still be picked (csrc/qr_pivot.hip) -----------
 * Same pivots as the refresh-per-batch scheme above (:739), fewer passes over Ur.  d_nrm_e holds norms that are exact
 * for the directions [0, j_e) (a copy of the initial norms, later rewritten by full sweeps); the rows with d_nrm_e >
 * theta form the POOL (spr_qr_pool_build: sorted local row list).  A pool sweep recomputes d_nrm = d_nrm_e - sum over the
 * epoch's directions [j_e, j) of (u . q)^2 for the pool's rows only and certifies the following steps against
 * max(tau of the pool, tau_floor = theta); a full sweep (d_pool = NULL) does it for every row, rewrites d_nrm_e and
 * starts the next epoch at j.  j_mark: the picks d_piv[j_mark, j) have not been taken out of the race yet.
 * Range: r a multiple of 16 up to SPR_MAX_R, 16-byte aligned rows, fewer than 2^31 local rows
 * (spr_qr_epoch_supported); at most spr_qr_epoch_max_directions(r) directions per sweep. */
int32_t spr_qr_epoch_supported(int32_t r, int64_t ldu, const void *d_Ur, int32_t u_is_f32, int64_t n_rows);
int32_t spr_qr_epoch_max_directions(int32_t r);
size_t spr_qr_pool_workspace(void);
int spr_qr_pool_build(const double *d_nrm_e, int64_t n_rows, double theta, int32_t *d_pool, int64_t cap,
                      int32_t *d_pool_n, void *d_workspace, size_t workspace_bytes, void *stream);
int spr_qr_epoch_sweep_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                           const double *d_Q, const int64_t *d_piv, int32_t j_e, int32_t j, int32_t j_mark,
                           double *d_nrm_e, double *d_nrm, const int32_t *d_pool, const int32_t *d_pool_n,
                           int64_t pool_n_host, double tau_floor, double *d_rec, double *d_tau,
                           void *d_workspace, size_t workspace_bytes, void *stream);
int spr_qr_exclude_f64(double *d_nrm, int64_t n_rows, int64_t row0, int64_t n_points,
                       const uint8_t *d_mask, const double *d_xyz, int32_t xyz_dim,
                       const int64_t *d_piv, int32_t nq, double d_min, void *stream);
int spr_qr_refresh_f64(const double *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                       const double *d_Q, const int64_t *d_piv, int32_t j0, int32_t nq,
                       double *d_nrm, double *d_rec, double *d_tau,
                       void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- K7 + K8 : Theta = C . Ur and cnt = C . X_cnt for a CSR measurement matrix ------
 * Replaces C.dot(self.Ur) (:797) and self.C.dot(self.X_cnt[:,0]) (:573).  The one-hot C
 * of optimal_placement is the 1-nnz-per-row case (a row gather).  Column indices are
 * GLOBAL rows; entries outside [row0,row0+n_rows) are skipped, so per-rank results are
 * partial sums to be all-reduced.  d_Theta is s x r row-major, d_cnt has s entries.
 * d_scl (optional, s entries) = C . X_scl[:,0] (sampling @ X_scl, :233) from the per-feature
 * d_scale[n_features] and the global n_points. */
int spr_measure_csr_f64(const int64_t *d_indptr, const int64_t *d_indices,
                        const double *d_vals, int32_t s, const double *d_Ur,
                        int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                        const double *d_rowmean, const double *d_scale, int64_t n_points,
                        int32_t n_features, double *d_Theta, double *d_cnt, double *d_scl,
                        void *stream);

/* ---- K8 + K9 : scale_vector + (weighted) least squares ------------------------------
 * Replaces scale_vector (:571-582) and the OLS branch of predict (:868-878) for n_p
 * measurement vectors at once.  d_y is n_p x s x 3 (value, std-dev, feature id).
 * Per vector: y0 = ((y-cnt)/scl, sigma/scl); W = I if every sigma is 0 else diag(1/y0_sigma);
 * normal equations (W Theta)^T (W Theta) a = (W Theta)^T W y0 by f64 MFMA + Cholesky.
 * Outputs: d_Ar, d_Ar_sigma (n_p x r), d_y0 (n_p x s x 2, may be NULL),
 * d_info (n_p x 2: [0] = 0 ok / 1 Cholesky breakdown / 2 a weight 1/sigma that is not finite -- an uncertainty that is zero or
 * NaN for SOME sensors makes W = diag(1/0) at :872 and np.linalg.pinv raises LinAlgError --, [1] = (max L_jj / min L_jj)^2). */
int spr_solve_ols_f64(const double *d_Theta, int32_t s, int32_t r, const double *d_cnt,
                      const double *d_scale, int32_t n_features, const double *d_y,
                      int32_t n_p, double *d_Ar, double *d_Ar_sigma, double *d_y0,
                      double *d_info, void *stream);

/* r > SPR_MAX_R (up to SPR_MAX_R_WIDE): the same normal-equations solve with its matrices in a workspace of
 * spr_solve_ols_workspace(s, r, n_p) bytes (L2-resident; one 1024-thread workgroup per vector), same outputs. */
size_t spr_solve_ols_workspace(int32_t s, int32_t r, int32_t n_p);
int spr_solve_ols_wide_f64(const double *d_Theta, int32_t s, int32_t r, const double *d_cnt,
                           const double *d_scale, int32_t n_features, const double *d_y, int32_t n_p,
                           double *d_Ar, double *d_Ar_sigma, double *d_y0, double *d_info, void *d_workspace,
                           size_t workspace_bytes, void *stream);

/* ---- K9b : the same solve with the reference's pseudo-inverse semantics ---------------
 * np.linalg.pinv(W @ Theta) (:873, :877; SVD, rcond = 1e-15) returns the MINIMUM-NORM least-squares solution when
 * W Theta is rank deficient or has fewer rows than columns (s < r: every GEM placement, :660-668).  Same inputs and
 * scaling as spr_solve_ols_f64; per vector the rows of [W Theta | W y0 | y0_sigma] are reduced to an r x (r+2)
 * triangular factor (streaming Householder QR, only when s > r) and a one-sided Jacobi SVD of that factor gives
 * a = sum_{sigma_i > rcond sigma_max} v_i (u_i^T b) / sigma_i.  predict() takes this path when s < r, on Cholesky
 * breakdown, or when the fast path reports cond^2 > 1e13.  s_cnt = entries of d_cnt (must equal s).
 * d_info (n_p x 4): [0] Jacobi sweeps (negative = not converged, or a non-finite weight 1/sigma as in spr_solve_ols_f64:
 * the reference's pinv raises LinAlgError for both), [1] rank kept, [2] sigma_max, [3] smallest
 * singular value kept. */
int spr_solve_pinv_f64(const double *d_Theta, int32_t s, int32_t r, const double *d_cnt, int32_t s_cnt,
                       const double *d_scale, int32_t n_features, const double *d_y, int32_t n_p,
                       double rcond, double *d_Ar, double *d_Ar_sigma, double *d_y0, double *d_info,
                       void *stream);

/* r > SPR_MAX_R (the reference keeps any r <= m modes, :336): the same algorithm with its factor in a workspace of
 * spr_solve_pinv_workspace(r, n_p) bytes (L2-resident; one 1024-thread workgroup per vector), r <= SPR_MAX_R_WIDE. */
size_t spr_solve_pinv_workspace(int32_t r, int32_t n_p);
int spr_solve_pinv_wide_f64(const double *d_Theta, int32_t s, int32_t r, const double *d_cnt, int32_t s_cnt,
                            const double *d_scale, int32_t n_features, const double *d_y, int32_t n_p,
                            double rcond, double *d_Ar, double *d_Ar_sigma, double *d_y0, double *d_info,
                            void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- synthetic snapshot matrices (benchmark input) -------------------
 * X[i,j] = (f+1) * ( sum_k L[i,k] R[k,j] + eps * N[i,j] ) + 10 f, with L, N standard
 * normal from a counter-based generator keyed by (seed, GLOBAL row, column), so any row
 * range is reproducible on any rank count.  d_R is k x ldr (ldr >= col0+ncols).
 * Columns [col0, col0+ncols) of the virtual matrix are written to d_X (n_rows x ldx). */
int spr_synth_f64(double *d_X, int64_t n_rows, int32_t ncols, int64_t ldx, int64_t row0,
                  int64_t n_points, int32_t col0, const double *d_R, int32_t k, int32_t ldr,
                  double eps, uint64_t seed, void *stream);
/* the same generator evaluated at arbitrary global rows for one column (held-out state) */
int spr_synth_gather_f64(const int64_t *d_rows, int32_t n, int64_t n_points, int32_t col,
                         const double *d_R, int32_t k, int32_t ldr, double eps,
                         uint64_t seed, double *d_out, void *stream);

/* ---- f32 STORAGE (BASELINE config 5: 50M cells x 16 features x 512 snapshots does not fit 8 x 288 GB in f64) ----
 * The reference keeps X in whatever float dtype the caller passes (sparse_sensing.py:74); its X_cnt / X_scl are float64
 * (np.zeros, :106-107), so X0 = (X - X_cnt)/X_scl (:169) and the U of its SVD (:272) are float64 whatever that dtype is.
 * These entry points take the snapshot shard as float (suffix _x32), widen every element on load and do ALL arithmetic
 * in f64 exactly like their _f64 twins -- same arguments, same outputs (row means, statistics, Gram blocks, norms, Theta,
 * fields stay double).  The basis is float64 by default (spr_project_x32_f64out, spr_project_stream_x32_f64out); storing
 * it as float as well is a STORAGE OPTION the reference does not have (spr_project_x32 / spr_project_stream_x32 write the
 * f64 result rounded once; the _u32 twins read such a basis), for shards whose f64 basis would not fit.  Two-element
 * pieces: rows 8-byte aligned and an even m / ld take the vector path.  With a float basis, parity of the sensors is
 * checked against both the reference's choice (pivots of the f64 basis) and dgeqp3 on the stored basis widened to f64. */
int spr_stats_gram_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                       int64_t n_points, int32_t n_features, int32_t center, double *d_rowmean,
                       void *d_workspace, size_t workspace_bytes, void *stream);
int spr_rowstats_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                     int64_t n_points, int32_t n_features, double *d_rowmean, double *d_fstats,
                     void *d_workspace, size_t workspace_bytes, void *stream);
int spr_gram_cross_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                       int64_t n_points, int32_t n_features, int32_t center, double *d_rowmean,
                       double *d_gram, void *d_workspace, size_t workspace_bytes, void *stream);
int spr_gram_cross_pair_x32(const float *d_X, int64_t n_rows, int32_t col_a, int32_t col_b, int32_t w_b,
                            int32_t ldg, int64_t ldx, int64_t row0, int64_t n_points, int32_t n_features,
                            int32_t center, double *d_rowmean, double *d_gram, void *d_workspace,
                            size_t workspace_bytes, void *stream);
int spr_project_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                    int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                    const double *d_rowmean, const double *d_W, int32_t r, float *d_Ur, int64_t ldu,
                    int32_t accumulate, void *stream);
/* f32 shard, f64 result: for the column slices of a wide X (m > 256), whose partial sums cancel by up to
 * sigma_1/sigma_r between the slices -- accumulate them in an f64 block of rows, round to f32 once afterwards */
int spr_project_x32_f64out(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                           int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                           const double *d_rowmean, const double *d_W, int32_t r, double *d_Ur, int64_t ldu,
                           int32_t accumulate, void *stream);
/* ... and its last slice: adds the f64 partial sums d_acc_in (row stride lda) and stores the f32 total in d_Ur */
int spr_project_x32_acc(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                        int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                        const double *d_rowmean, const double *d_W, int32_t r, const double *d_acc_in,
                        int64_t lda, float *d_Ur, int64_t ldu, void *stream);
/* streamed-W projection (any m) of an f32 shard: float basis, or double basis (the reference's dtype for a float32 X:
 * X_cnt / X_scl are float64, :106-107, so X0 = (X - X_cnt)/X_scl and U are float64 whatever the dtype of X) */
int spr_project_stream_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                           int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                           const double *d_rowmean, const double *d_W, int32_t r, float *d_Ur, int64_t ldu,
                           void *d_workspace, size_t workspace_bytes, void *stream);
int spr_project_stream_x32_f64out(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                  int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                                  const double *d_rowmean, const double *d_W, int32_t r, double *d_Ur, int64_t ldu,
                                  void *d_workspace, size_t workspace_bytes, void *stream);
int spr_project_norms_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                          int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                          const double *d_rowmean, const double *d_W, int32_t r, float *d_Ur, int64_t ldu,
                          double *d_rownorm2, void *stream);
int spr_project_norms_x32_f64out(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                 int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                                 const double *d_rowmean, const double *d_W, int32_t r, double *d_Ur, int64_t ldu,
                                 double *d_rownorm2, void *stream);
int spr_project_stream_norms_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                 int64_t n_points, int32_t n_features, int32_t center, const double *d_inv_scale,
                                 const double *d_rowmean, const double *d_W, int32_t r, float *d_Ur, int64_t ldu,
                                 double *d_rownorm2, void *d_workspace, size_t workspace_bytes, void *stream);
int spr_project_stream_norms_x32_f64out(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                                        int64_t n_points, int32_t n_features, int32_t center,
                                        const double *d_inv_scale, const double *d_rowmean, const double *d_W,
                                        int32_t r, double *d_Ur, int64_t ldu, double *d_rownorm2, void *d_workspace,
                                        size_t workspace_bytes, void *stream);
int spr_stats_gram_shifted_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                               int64_t n_points, int32_t n_features, const double *d_shift, double *d_rowsum,
                               void *d_workspace, size_t workspace_bytes, void *stream);
int spr_scale_rows_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                       int64_t n_points, int32_t n_features, const double *d_rowmean,
                       const double *d_inv_scale, double *d_X0, int64_t ldo, void *stream);
int spr_feature_minmax_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                           int64_t n_points, int32_t n_features, double *d_minmax,
                           void *d_workspace, size_t workspace_bytes, void *stream);
int spr_colsums_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                    int64_t n_points, int32_t n_features, const double *d_rowmean, double *d_out,
                    void *d_workspace, size_t workspace_bytes, void *stream);
int spr_feature_digit_hist_x32(const float *d_X, int64_t n_rows, int32_t m, int64_t ldx, int64_t row0,
                               int64_t n_points, int32_t n_features, const uint64_t *d_prefix, int32_t shift,
                               int32_t bits, int32_t two_targets, uint64_t *d_hist, void *stream);
int spr_synth_f32(float *d_X, int64_t n_rows, int32_t ncols, int64_t ldx, int64_t row0,
                  int64_t n_points, int32_t col0, const double *d_R, int32_t k, int32_t ldr,
                  double eps, uint64_t seed, void *stream);
int spr_reconstruct_u32(const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                        int64_t n_points, int32_t n_features, const double *d_rowmean,
                        const double *d_scale, const double *d_rowscale, const double *d_A, int32_t n_p,
                        double *d_Xrec, int64_t ldo, void *stream);
int spr_mask_rows_u32(float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu,
                      const uint8_t *d_mask, void *stream);
int spr_qr_init_u32(const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu,
                    int64_t row0, double *d_nrm, double *d_rec, double *d_tau,
                    void *d_workspace, size_t workspace_bytes, void *stream);
int spr_qr_init_norms_u32(const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                          const double *d_nrm0, double *d_nrm, double *d_rec, double *d_tau,
                          void *d_workspace, size_t workspace_bytes, void *stream);
int spr_qr_epoch_sweep_u32(const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                           const double *d_Q, const int64_t *d_piv, int32_t j_e, int32_t j, int32_t j_mark,
                           double *d_nrm_e, double *d_nrm, const int32_t *d_pool, const int32_t *d_pool_n,
                           int64_t pool_n_host, double tau_floor, double *d_rec, double *d_tau,
                           void *d_workspace, size_t workspace_bytes, void *stream);
int spr_qr_refresh_u32(const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu, int64_t row0,
                       const double *d_Q, const int64_t *d_piv, int32_t j0, int32_t nq,
                       double *d_nrm, double *d_rec, double *d_tau,
                       void *d_workspace, size_t workspace_bytes, void *stream);
int spr_measure_csr_u32(const int64_t *d_indptr, const int64_t *d_indices, const double *d_vals,
                        int32_t s, const float *d_Ur, int64_t n_rows, int32_t r, int64_t ldu,
                        int64_t row0, const double *d_rowmean, const double *d_scale, int64_t n_points,
                        int32_t n_features, double *d_Theta, double *d_cnt, double *d_scl, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPR_HIP_H */
