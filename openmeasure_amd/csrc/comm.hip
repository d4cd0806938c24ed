// Collectives behind the C ABI (round 6; SURVEY.md 8(b): spr_fit_stats_gram "(includes collectives)").
//
// north_star: "a single RCCL all-reduce over xGMI for the Gram matrix and a final all-gather for the reconstructed field".  Until
// round 6 every RCCL call of the sharded path lived in torch.distributed, so a caller of include/spr_hip.h written in C could
// not run it.  This file puts the communicator, the two collectives and the whole first pass of fit() -- Gram kernel, finalize,
// all-reduce, statistics merge + scaled sum -- behind the library's own entry points: one enqueue on the caller's stream,
// nothing of the host between the kernels and the collective.
//
// RCCL is reached through dlopen: the library that is ALREADY in the process (PyTorch ships its own librccl.so and two copies
// of RCCL in one process must not both own the GPU's IPC state) or, in a process without one, librccl.so.1 of the ROCm
// installation.  libspr_hip.so itself therefore still links nothing but the HIP runtime, loads on a machine without RCCL and in
// the CPU-side sanitizer build; only spr_comm_* fail there, with SPR_E_UNSUPPORTED and the loader's text.
#include <dlfcn.h>
#include <link.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>   // types and enumerators only: every call goes through the table below

#include "common.hpp"

namespace {
struct Rccl {
  void *handle = nullptr;
  char where[256] = "";
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

int find_loaded(struct dl_phdr_info *info, size_t, void *out) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
    strncpy(static_cast<char *>(out), info->dlpi_name, 255);
    return 1;
  }
  return 0;
}

struct Loaded {
  Rccl r;
  bool ok = false;
  char why[400] = "";
};

// one attempt per process (function-local static: initialised once, also with several threads calling in)
Loaded load_rccl() {
  Loaded L;
  void *h = nullptr;
  char path[256] = "";
  if (const char *e = getenv("SPR_RCCL_LIBRARY")) {
    h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
    strncpy(path, e, 255);
  }
  if (!h && dl_iterate_phdr(find_loaded, path)) h = dlopen(path, RTLD_NOW | RTLD_NOLOAD);   // the copy the process has
  static const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (int i = 0; !h && i < 3; ++i) {
    h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    strncpy(path, names[i], 255);
  }
  if (!h) {
    const char *de = dlerror();
    snprintf(L.why, sizeof L.why, "spr_comm: no RCCL library could be loaded (%s); SPR_RCCL_LIBRARY=<path> names one",
             de ? de : "no loader message");
    return L;
  }
  L.r.handle = h;
  strncpy(L.r.where, path, 255);
#define SPR_SYM(field, name)                                                                  \
  L.r.field = reinterpret_cast<decltype(L.r.field)>(dlsym(h, name));                          \
  if (!L.r.field) {                                                                           \
    snprintf(L.why, sizeof L.why, "spr_comm: %s has no symbol %s", path, name);               \
    return L;                                                                                 \
  }
  SPR_SYM(GetUniqueId, "ncclGetUniqueId")
  SPR_SYM(CommInitRank, "ncclCommInitRank")
  SPR_SYM(CommDestroy, "ncclCommDestroy")
  SPR_SYM(AllReduce, "ncclAllReduce")
  SPR_SYM(AllGather, "ncclAllGather")
  SPR_SYM(GetErrorString, "ncclGetErrorString")
#undef SPR_SYM
  L.ok = true;
  return L;
}

// -> the table, or nullptr with the reason in spr_last_error()
const Rccl *rccl() {
  static const Loaded L = load_rccl();
  if (L.ok) return &L.r;
  spr_set_error("%s", L.why);
  return nullptr;
}

struct SprComm {
  ncclComm_t comm;
  int rank, world;
};
// the communicators this library has handed out: a handle is looked up here, never dereferenced on trust (a caller's stale or
// foreign pointer gets SPR_E_INVALID, not a fault)
constexpr int kMaxComms = 64;
SprComm *g_comms[kMaxComms] = {nullptr};
pthread_mutex_t g_comms_lock = PTHREAD_MUTEX_INITIALIZER;

#define SPR_RCCL_TRY(api, expr)                                                                                    \
  do {                                                                                                             \
    ncclResult_t r__ = (expr);                                                                                     \
    if (r__ != ncclSuccess) {                                                                                      \
      spr_set_error("%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r__), __FILE__, __LINE__);               \
      return SPR_E_HIP;                                                                                            \
    }                                                                                                              \
  } while (0)

SprComm *as_comm(void *p) {
  if (!p) return nullptr;
  SprComm *found = nullptr;
  pthread_mutex_lock(&g_comms_lock);
  for (int i = 0; i < kMaxComms; ++i)
    if (g_comms[i] == p) found = g_comms[i];
  pthread_mutex_unlock(&g_comms_lock);
  return found;
}

__global__ void set_slot_kernel(double *__restrict__ p, double v) { *p = v; }
}  // namespace

extern "C" size_t spr_comm_unique_id_bytes(void) { return sizeof(ncclUniqueId); }

extern "C" int spr_comm_unique_id(void *h_id) {
  SPR_REQUIRE(h_id, SPR_E_INVALID, "spr_comm_unique_id: NULL output");
  const Rccl *api = rccl();
  if (!api) return SPR_E_UNSUPPORTED;
  ncclUniqueId id;
  SPR_RCCL_TRY(api, api->GetUniqueId(&id));
  memcpy(h_id, &id, sizeof id);
  return SPR_OK;
}

extern "C" int spr_comm_init(const void *h_id, int32_t rank, int32_t world, void **comm) {
  SPR_REQUIRE(h_id && comm, SPR_E_INVALID, "spr_comm_init: NULL pointer");
  SPR_REQUIRE(world >= 1 && rank >= 0 && rank < world, SPR_E_INVALID, "spr_comm_init: rank=%d world=%d", rank, world);
  const Rccl *api = rccl();
  if (!api) return SPR_E_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, h_id, sizeof id);
  ncclComm_t c = nullptr;
  SPR_RCCL_TRY(api, api->CommInitRank(&c, world, id, rank));   // on the CURRENT device; collective over the `world` callers
  SprComm *out = static_cast<SprComm *>(malloc(sizeof(SprComm)));
  if (!out) {
    api->CommDestroy(c);
    spr_set_error("spr_comm_init: out of host memory");
    return SPR_E_HIP;
  }
  out->comm = c;
  out->rank = rank;
  out->world = world;
  int slot = -1;
  pthread_mutex_lock(&g_comms_lock);
  for (int i = 0; i < kMaxComms && slot < 0; ++i)
    if (!g_comms[i]) g_comms[slot = i] = out;
  pthread_mutex_unlock(&g_comms_lock);
  if (slot < 0) {
    api->CommDestroy(c);
    free(out);
    spr_set_error("spr_comm_init: %d communicators are alive already", kMaxComms);
    return SPR_E_UNSUPPORTED;
  }
  *comm = out;
  return SPR_OK;
}

extern "C" int spr_comm_destroy(void *comm) {
  SprComm *c = as_comm(comm);
  SPR_REQUIRE(c, SPR_E_INVALID, "spr_comm_destroy: not a communicator of this library");
  const Rccl *api = rccl();
  if (!api) return SPR_E_UNSUPPORTED;
  pthread_mutex_lock(&g_comms_lock);
  for (int i = 0; i < kMaxComms; ++i)
    if (g_comms[i] == c) g_comms[i] = nullptr;
  pthread_mutex_unlock(&g_comms_lock);
  ncclResult_t r = api->CommDestroy(c->comm);
  free(c);
  if (r != ncclSuccess) {
    spr_set_error("ncclCommDestroy failed: %s", api->GetErrorString(r));
    return SPR_E_HIP;
  }
  return SPR_OK;
}

extern "C" int spr_comm_info(void *comm, int32_t *rank, int32_t *world) {
  SprComm *c = as_comm(comm);
  SPR_REQUIRE(c && rank && world, SPR_E_INVALID, "spr_comm_info: not a communicator of this library / NULL output");
  *rank = c->rank;
  *world = c->world;
  return SPR_OK;
}

extern "C" const char *spr_comm_library(void) {
  const Rccl *api = rccl();
  return api ? api->where : "";
}

extern "C" int spr_allreduce_f64(void *comm, double *d_buf, int64_t count, void *stream) {
  SprComm *c = as_comm(comm);
  SPR_REQUIRE(c, SPR_E_INVALID, "spr_allreduce_f64: not a communicator of this library");
  SPR_REQUIRE(d_buf && count > 0, SPR_E_INVALID, "spr_allreduce_f64: NULL buffer / count=%lld", (long long)count);
  const Rccl *api = rccl();
  if (!api) return SPR_E_UNSUPPORTED;
  SPR_RCCL_TRY(api, api->AllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, c->comm, static_cast<hipStream_t>(stream)));
  return SPR_OK;
}

extern "C" int spr_allreduce_i64(void *comm, int64_t *d_buf, int64_t count, void *stream) {
  SprComm *c = as_comm(comm);
  SPR_REQUIRE(c, SPR_E_INVALID, "spr_allreduce_i64: not a communicator of this library");
  SPR_REQUIRE(d_buf && count > 0, SPR_E_INVALID, "spr_allreduce_i64: NULL buffer / count=%lld", (long long)count);
  const Rccl *api = rccl();
  if (!api) return SPR_E_UNSUPPORTED;
  SPR_RCCL_TRY(api, api->AllReduce(d_buf, d_buf, (size_t)count, ncclInt64, ncclSum, c->comm, static_cast<hipStream_t>(stream)));
  return SPR_OK;
}

extern "C" int spr_allgather(void *comm, const void *d_send, void *d_recv, int64_t bytes_per_rank, void *stream) {
  SprComm *c = as_comm(comm);
  SPR_REQUIRE(c, SPR_E_INVALID, "spr_allgather: not a communicator of this library");
  SPR_REQUIRE(d_send && d_recv && bytes_per_rank > 0, SPR_E_INVALID, "spr_allgather: NULL buffer / bytes_per_rank=%lld",
              (long long)bytes_per_rank);
  const Rccl *api = rccl();
  if (!api) return SPR_E_UNSUPPORTED;
  SPR_RCCL_TRY(api, api->AllGather(d_send, d_recv, (size_t)bytes_per_rank, ncclChar, c->comm, static_cast<hipStream_t>(stream)));
  return SPR_OK;
}

// ---- the first pass of fit() with its collective, as ONE enqueue -------------------------------------------------------------
// Buffer layout (doubles; the very buffer openmeasure_amd/_shard.py all-reduces through torch.distributed):
//   [ F m m   per-feature Gram blocks | world x F x 3  (count, mean, M2) per rank and feature | world  first global row per rank ]
// every rank writes its own slots and zeros elsewhere, so that the ONE sum hands every rank all ranks' statistics and row blocks
// in rank order (adding zeros is exact).
extern "C" size_t spr_fit_gram_pass_buffer(int32_t m, int32_t n_features, int32_t world) {
  if (m < 1 || n_features < 1 || world < 1) return 0;
  return ((size_t)n_features * m * m + (size_t)world * n_features * 3 + (size_t)world) * sizeof(double);
}

// Workspace of spr_fit_gram_pass: m <= 256 the Gram workspace; 256 < m <= 512 (the column-split path: BASELINE config 5 has m = 512)
// the larger of the Gram / cross / row-statistics workspaces, then n_rows doubles (raw row sums of the second slice) and 3 F doubles.
namespace {
size_t align256(size_t b) { return (b + 255) / 256 * 256; }
size_t wide_kernel_ws(int32_t m, int32_t F) {
  size_t a = spr_stats_gram_workspace(SPR_MAX_M, F), b = spr_stats_gram_workspace(m - SPR_MAX_M, F);
  size_t c = spr_gram_cross_workspace(m, F), d = spr_rowstats_workspace(F);
  size_t w = a > b ? a : b;
  if (c > w) w = c;
  if (d > w) w = d;
  return align256(w);
}
}  // namespace

extern "C" size_t spr_fit_gram_pass_workspace(int32_t m, int32_t n_features, int64_t n_rows) {
  if (m < 1 || n_features < 1 || n_rows < 1 || m > SPR_MAX_M_WIDE) return 0;
  if (m <= SPR_MAX_M) return spr_stats_gram_workspace(m, n_features);
  return wide_kernel_ws(m, n_features) + align256((size_t)n_rows * sizeof(double)) + align256((size_t)n_features * 3 * sizeof(double));
}

extern "C" int spr_fit_gram_pass(void *comm, const void *d_X, int32_t x_is_f32, int64_t n_rows, int32_t m, int64_t ldx,
                                 int64_t row0, int64_t n_points, int32_t n_features, int32_t scale_code, double *d_rowmean,
                                 double *d_buf, size_t buf_bytes, double *d_G, double *d_feat, double *d_scale,
                                 double *d_inv_scale, void *d_workspace, size_t workspace_bytes, void *stream) {
  int rank = 0, world = 1;
  SprComm *c = nullptr;
  if (comm) {
    c = as_comm(comm);
    SPR_REQUIRE(c, SPR_E_INVALID, "spr_fit_gram_pass: not a communicator of this library");
    rank = c->rank;
    world = c->world;
  }
  SPR_REQUIRE(d_X && d_rowmean && d_buf && d_G && d_feat && d_scale && d_inv_scale, SPR_E_INVALID,
              "spr_fit_gram_pass: NULL pointer");
  SPR_REQUIRE(m >= 1 && m <= SPR_MAX_M_WIDE, SPR_E_UNSUPPORTED,
              "spr_fit_gram_pass: m=%d (1..%d: wider matrices go slice by slice, spr_rowstats / spr_gram_cross_pair)", m, SPR_MAX_M_WIDE);
  SPR_REQUIRE(n_features >= 1 && n_rows >= 1, SPR_E_INVALID, "spr_fit_gram_pass: n_rows=%lld n_features=%d", (long long)n_rows,
              n_features);
  const size_t need = spr_fit_gram_pass_buffer(m, n_features, world);
  SPR_REQUIRE(buf_bytes >= need, SPR_E_WORKSPACE, "spr_fit_gram_pass: buffer of %zu bytes, %zu needed", buf_bytes, need);
  const size_t ws_need = spr_fit_gram_pass_workspace(m, n_features, n_rows);
  SPR_REQUIRE(d_workspace && workspace_bytes >= ws_need, SPR_E_WORKSPACE, "spr_fit_gram_pass: workspace of %zu bytes, %zu needed",
              workspace_bytes, ws_need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t n_gram = (size_t)n_features * m * m;
  double *fstats_all = d_buf + n_gram;
  double *fstats_mine = fstats_all + (size_t)rank * n_features * 3;
  double *rows_all = fstats_all + (size_t)world * n_features * 3;
  SPR_HIP_TRY(hipMemsetAsync(d_buf, 0, need, st));
  int rc;
  if (m <= SPR_MAX_M) {
    rc = x_is_f32 ? spr_stats_gram_x32(static_cast<const float *>(d_X), n_rows, m, ldx, row0, n_points, n_features, 1, d_rowmean,
                                       d_workspace, workspace_bytes, stream)
                  : spr_stats_gram_f64(static_cast<const double *>(d_X), n_rows, m, ldx, row0, n_points, n_features, 1,
                                       d_rowmean, d_workspace, workspace_bytes, stream);
    if (rc) return rc;
    rc = spr_stats_gram_finalize_f64(n_rows, m, row0, n_points, n_features, d_workspace, workspace_bytes, fstats_mine, d_buf, m, 0,
                                     stream);
    if (rc) return rc;
  } else {
    // 256 < m <= 512: columns A = [0, 256), B = [256, m).  Every launch shifts the rows by the mean of their FIRST 256 columns --
    // formed for free by the symmetric launch on A --, P G P afterwards gives the Gram blocks of the row-centred data and the
    // true row means (include/spr_hip.h, "K1 + K3a for 256 < m <= 512"); the statistics of the row means come last.
    const int mA = SPR_MAX_M, mB = m - SPR_MAX_M;
    const size_t kws = wide_kernel_ws(m, n_features);
    char *base = static_cast<char *>(d_workspace);
    double *rowsum_b = reinterpret_cast<double *>(base + kws);
    double *scratch = reinterpret_cast<double *>(base + kws + align256((size_t)n_rows * sizeof(double)));
    const size_t esz = x_is_f32 ? sizeof(float) : sizeof(double);
    const char *xb = static_cast<const char *>(d_X) + (size_t)mA * esz;
    rc = x_is_f32 ? spr_stats_gram_x32(static_cast<const float *>(d_X), n_rows, mA, ldx, row0, n_points, n_features, 1, d_rowmean,
                                       d_workspace, kws, stream)
                  : spr_stats_gram_f64(static_cast<const double *>(d_X), n_rows, mA, ldx, row0, n_points, n_features, 1,
                                       d_rowmean, d_workspace, kws, stream);
    if (rc) return rc;
    rc = spr_stats_gram_finalize_f64(n_rows, mA, row0, n_points, n_features, d_workspace, kws, scratch, d_buf, m, 0, stream);
    if (rc) return rc;
    rc = x_is_f32 ? spr_gram_cross_x32(static_cast<const float *>(d_X), n_rows, m, ldx, row0, n_points, n_features, 2, d_rowmean,
                                       d_buf, d_workspace, kws, stream)
                  : spr_gram_cross_f64(static_cast<const double *>(d_X), n_rows, m, ldx, row0, n_points, n_features, 2, d_rowmean,
                                       d_buf, d_workspace, kws, stream);
    if (rc) return rc;
    rc = x_is_f32 ? spr_stats_gram_shifted_x32(reinterpret_cast<const float *>(xb), n_rows, mB, ldx, row0, n_points, n_features,
                                               d_rowmean, rowsum_b, d_workspace, kws, stream)
                  : spr_stats_gram_shifted_f64(reinterpret_cast<const double *>(xb), n_rows, mB, ldx, row0, n_points, n_features,
                                               d_rowmean, rowsum_b, d_workspace, kws, stream);
    if (rc) return rc;
    rc = spr_stats_gram_finalize_f64(n_rows, mB, row0, n_points, n_features, d_workspace, kws, scratch, d_buf, m, mA, stream);
    if (rc) return rc;
    rc = spr_gram_shift_finish_f64(d_rowmean, rowsum_b, n_rows, mA, m, d_buf, n_features, stream);
    if (rc) return rc;
    rc = spr_rowmean_stats_f64(d_rowmean, n_rows, row0, n_points, n_features, fstats_mine, d_workspace, kws, stream);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(set_slot_kernel, dim3(1), dim3(1), 0, st, rows_all + rank, (double)row0);   // exact below 2^53
  SPR_LAUNCH_CHECK();
  if (c) {   // (a one-rank communicator reduces onto itself: the same call path, which is what one GPU can test)
    const Rccl *api = rccl();
    if (!api) return SPR_E_UNSUPPORTED;
    SPR_RCCL_TRY(api, api->AllReduce(d_buf, d_buf, need / sizeof(double), ncclDouble, ncclSum, c->comm, st));
  }
  return spr_gram_combine_f64(d_buf, fstats_all, world, n_features, m, scale_code, d_G, d_feat, d_scale, d_inv_scale, stream);
}
