// CU-free exchange of the reconstructed field between the ranks of one node (round 5).
//
// reconstruct() of a row-sharded SPR ends with every rank handing its (n_p, n_loc) block of the field to every other rank
// (north_star: "a final all-gather for the reconstructed field"; reference: the (n, n_p) array Ur @ Ar.T of
// sparse_sensing.py:371-375, whole on every caller).  RCCL's all-gather does that with a device kernel of 256 threads,
// 261-280 VGPRs per wave and 19.7 KB of LDS, which cannot share a compute unit with the Gram or projection workgroups
// (DESIGN.md 5c): left in flight under the next fit() it only progresses where a CU is free.  This file moves the same bytes
// without a single wave:
//
//   * every rank owns one buffer from hipMalloc (the one allocation this library makes: an interprocess handle needs the
//     BASE pointer of an allocation, which a sub-allocating caller cannot give) and exports it (hipIpcGetMemHandle); the
//     other ranks of the node map it (hipIpcOpenMemHandle, peer access enabled lazily);
//   * a rank's block goes straight from its own copy of the field into the same place of every peer's copy with
//     hipMemcpyAsync(..., hipMemcpyDeviceToDeviceNoCU): the SDMA engines, one stream per peer so the copies to different
//     peers use different engines / xGMI links;
//   * arrival and buffer release are 64-bit counters in exported fine-grained memory: raised with hipStreamWriteValue64 behind
//     the copies (copy streams), set / awaited on the compute stream by ONE single-wave kernel each -- microseconds of one
//     wave, nothing that occupies a compute unit while the bytes move.
//
// Measured on one MI355X, two processes (tools/archive/lab/p2p_probe.hip, profiles/r05_p2p_probe.txt): a NoCU copy into the other
// process's buffer runs at 61 GB/s and leaves a CU-filling kernel's time unchanged (24.87 ms with and without), the same
// copy as a blit kernel takes 15 ms next to that kernel; write/wait values work on IPC-mapped device memory.
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "common.hpp"

namespace {
hipMemcpyKind p2p_kind() {
  // SPR_P2P_BLIT=1: the default device-to-device copy (blit kernels on compute units) instead of the SDMA engines -- A/B only
  static int blit = -1;
  if (blit < 0) {
    const char *e = getenv("SPR_P2P_BLIT");
    blit = (e && atoi(e) == 1) ? 1 : 0;
  }
  return blit ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToDeviceNoCU;
}
// SPR_P2P_PROBE=1: spr_field_gather_p2p prints the host time of each of its runtime calls (tools/p2p_push_probe.py)
bool p2p_probe() {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("SPR_P2P_PROBE");
    on = (e && atoi(e) == 1) ? 1 : 0;
  }
  return on == 1;
}
double host_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
}  // namespace

extern "C" size_t spr_p2p_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

extern "C" int spr_p2p_alloc(size_t n_bytes, int32_t kind, void **d_ptr, void *h_handle) {
  SPR_REQUIRE(d_ptr && h_handle, SPR_E_INVALID, "spr_p2p_alloc: NULL output");
  SPR_REQUIRE(n_bytes > 0 && n_bytes % 4096 == 0, SPR_E_INVALID, "spr_p2p_alloc: n_bytes=%zu must be a positive multiple of 4096",
              n_bytes);
  SPR_REQUIRE(kind >= 0 && kind <= 2, SPR_E_INVALID, "spr_p2p_alloc: kind=%d (0 coarse-grained, 1 fine-grained, 2 uncached)", kind);
  void *p = nullptr;
  if (kind == 0)
    SPR_HIP_TRY(hipMalloc(&p, n_bytes));
  else
    SPR_HIP_TRY(hipExtMallocWithFlags(&p, n_bytes, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    spr_set_error("hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 is needed on hosts with dmabuf IPC only)",
                  hipGetErrorString(e));
    return SPR_E_HIP;
  }
  memcpy(h_handle, &h, sizeof h);
  *d_ptr = p;
  return SPR_OK;
}

extern "C" int spr_p2p_free(void *d_ptr) {
  if (d_ptr) SPR_HIP_TRY(hipFree(d_ptr));
  return SPR_OK;
}

extern "C" int spr_p2p_open(const void *h_handle, void **d_mapped) {
  SPR_REQUIRE(h_handle && d_mapped, SPR_E_INVALID, "spr_p2p_open: NULL pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, h_handle, sizeof h);
  void *p = nullptr;
  SPR_HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  *d_mapped = p;
  return SPR_OK;
}

// Which GPU is this process on, in a form another process can compare: the PCI bus id of the current device ("0000:c1:00.0").
// Ranks exchange it next to their handles, so that nobody maps -- and then WRITES to -- memory of a device its own GPU has no
// peer access to (or cannot even see: HIP_VISIBLE_DEVICES differs per process): such a write is a memory fault, not an error code.
extern "C" int spr_p2p_device_id(char *h_buf, int32_t n_buf) {
  SPR_REQUIRE(h_buf && n_buf >= 16, SPR_E_INVALID, "spr_p2p_device_id: buffer of at least 16 bytes");
  int dev = 0;
  SPR_HIP_TRY(hipGetDevice(&dev));
  SPR_HIP_TRY(hipDeviceGetPCIBusId(h_buf, n_buf, dev));
  return SPR_OK;
}

// *h_can = 1 when the device with that bus id IS the current device or the current device may access its memory
// (hipDeviceCanAccessPeer), 0 when not -- including a device this process cannot see.
extern "C" int spr_p2p_peer_access(const char *h_bus_id, int32_t *h_can) {
  SPR_REQUIRE(h_bus_id && h_can, SPR_E_INVALID, "spr_p2p_peer_access: NULL pointer");
  *h_can = 0;
  int dev = 0, peer = -1;
  SPR_HIP_TRY(hipGetDevice(&dev));
  if (hipDeviceGetByPCIBusId(&peer, h_bus_id) != hipSuccess) {
    (void)hipGetLastError();                                   // not visible here: an answer, not a failure
    return SPR_OK;
  }
  if (peer == dev) {
    *h_can = 1;
    return SPR_OK;
  }
  int can = 0;
  SPR_HIP_TRY(hipDeviceCanAccessPeer(&can, dev, peer));
  *h_can = can ? 1 : 0;
  return SPR_OK;
}

extern "C" int spr_p2p_close(void *d_mapped) {
  if (d_mapped) SPR_HIP_TRY(hipIpcCloseMemHandle(d_mapped));
  return SPR_OK;
}

extern "C" int spr_p2p_signal(void *d_flag, uint64_t value, void *stream) {
  SPR_REQUIRE(d_flag && (uintptr_t)d_flag % 8 == 0, SPR_E_INVALID, "spr_p2p_signal: NULL / unaligned flag");
  SPR_HIP_TRY(hipStreamWriteValue64(static_cast<hipStream_t>(stream), d_flag, value, 0));
  return SPR_OK;
}

extern "C" int spr_p2p_wait(void *d_flag, uint64_t value, void *stream) {
  SPR_REQUIRE(d_flag && (uintptr_t)d_flag % 8 == 0, SPR_E_INVALID, "spr_p2p_wait: NULL / unaligned flag");
  SPR_HIP_TRY(hipStreamWaitValue64(static_cast<hipStream_t>(stream), d_flag, value, hipStreamWaitValueGte, ~0ull));
  return SPR_OK;
}

extern "C" int spr_p2p_copy(void *d_dst, const void *d_src, int64_t n_bytes, void *stream) {
  SPR_REQUIRE(d_dst && d_src && n_bytes > 0, SPR_E_INVALID, "spr_p2p_copy: NULL pointer / empty copy");
  SPR_HIP_TRY(hipMemcpyAsync(d_dst, d_src, (size_t)n_bytes, p2p_kind(), static_cast<hipStream_t>(stream)));
  return SPR_OK;
}

// ---- counters set and awaited by single-wave kernels ------------------------------------------------------------------
// hipStreamWriteValue64 / hipStreamWaitValue64 are themselves one tiny kernel each on this runtime (__amd_rocclr_streamOpsWrite /
// streamOpsWait in a rocprofv3 trace, 4-8 us plus 10-15 us of dispatch gap): seven of each in front of every reconstruct
// kernel cost 0.15 ms -- and 0.9 ms when the stream also had to wait for events behind SDMA copies of another hardware queue
// (profiles/r05_p2p_gap_experiments.txt).  On the COMPUTE stream the exchange therefore uses ONE kernel per release and ONE per
// join, over a table of counters passed by value; completion of this rank's own pushes is a counter as well (written on the
// copy stream behind the copies), so the compute stream never waits for an event of another queue.
//
// POISON (round 6).  Counters only ever grow and every wait is "counter >= value", so one bit far above any gather count says
// "the exchange this counter belongs to is broken": a wait that reads it fails at once (status words, like a time-out), and a
// push whose wait for the peer's release failed raises the arrival counters it owes WITH that bit -- the peer's join and the
// pusher's own join of that gather then both fail instead of handing out a field that was overwritten while it may still have
// been read, or one assembled around a block nobody vouches for.
namespace {
constexpr int kMaxFlags = 128;
constexpr unsigned long long kPoison = 1ull << 62;
struct FlagTable {
  unsigned long long *p[kMaxFlags];
  unsigned short id[kMaxFlags];      // what a failing lane reports as its index (a sub-table keeps the caller's numbering)
  unsigned short add[kMaxFlags];     // a wait on counter i is for value + add[i] (one launch may await counters of two sequences)
};

// *t.p[i] = value for all i -- OR value | POISON when the word at `gate` (NULL: no gate) is non-zero: the status word a failed
// release wait of this push has left (see spr_field_gather_p2p).
__global__ __launch_bounds__(128) void p2p_flags_set_kernel(FlagTable t, int n, unsigned long long value,
                                                            const unsigned long long *__restrict__ gate) {
  unsigned long long v = value;
  if (gate && __hip_atomic_load(gate, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 0) v |= kPoison;
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    __hip_atomic_store(t.p[i], v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Every lane polls one counter until it has reached `value`.  Exit condition every wave reaches: `timeout_ticks` of the
// 100 MHz wall clock, or a poisoned counter; a lane that gives up leaves (its index + 1) in status[0] and the value it saw in
// status[1] (bit 62 set: poisoned, otherwise timed out) -- the host reads the two words where it synchronises anyway and raises.
__global__ __launch_bounds__(128) void p2p_flags_wait_kernel(FlagTable t, int n, unsigned long long value,
                                                             unsigned long long timeout_ticks,
                                                             unsigned long long *__restrict__ status) {
  const unsigned long long t0 = wall_clock64();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    unsigned long long seen;
    bool bad = false;
    int polls = 0;
    for (;;) {
      seen = __hip_atomic_load(t.p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (seen & kPoison) { bad = true; break; }
      if (seen >= value + t.add[i]) break;
      if (wall_clock64() - t0 > timeout_ticks) { bad = true; break; }
      // back off: a counter that is about to arrive is seen within a fraction of a microsecond; a copy stream's wait for the
      // NEXT gather's block sits here for most of a step (13.8 of 21 ms on one rank's block of config 4,
      // profiles/r06_c4share8_p2p_deferred_timeline.txt) and should not poll 2 000 times per millisecond next to the Gram waves
      if (polls < 64) {
        ++polls;
        __builtin_amdgcn_s_sleep(8);
      } else {
        __builtin_amdgcn_s_sleep(127);
      }
    }
    if (bad && status) {
      __hip_atomic_store(status + 1, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(status, (unsigned long long)(t.id[i] + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: what the counters announce is visible to what runs next
}

int fill_table(FlagTable &t, void *const *ptrs, int n, const char *who) {
  SPR_REQUIRE(n >= 1 && n <= kMaxFlags && ptrs, SPR_E_INVALID, "%s: n=%d counters (1..%d) / NULL table", who, n, kMaxFlags);
  for (int i = 0; i < n; ++i) {
    SPR_REQUIRE(ptrs[i] && (uintptr_t)ptrs[i] % 8 == 0, SPR_E_INVALID, "%s: counter %d is NULL or unaligned", who, i);
    t.p[i] = static_cast<unsigned long long *>(ptrs[i]);
    t.id[i] = (unsigned short)i;
    t.add[i] = 0;
  }
  return SPR_OK;
}

int launch_wait(const FlagTable &t, int n, uint64_t value, double timeout_s, void *d_status, hipStream_t st) {
  hipLaunchKernelGGL(p2p_flags_wait_kernel, dim3(1), dim3(n > 64 ? 128 : 64), 0, st, t, n, (unsigned long long)value,
                     (unsigned long long)(timeout_s * 1e8), static_cast<unsigned long long *>(d_status));
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

int launch_set(const FlagTable &t, int n, uint64_t value, const void *d_gate, hipStream_t st) {
  hipLaunchKernelGGL(p2p_flags_set_kernel, dim3(1), dim3(n > 64 ? 128 : 64), 0, st, t, n, (unsigned long long)value,
                     static_cast<const unsigned long long *>(d_gate));
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}
}  // namespace

extern "C" uint64_t spr_p2p_poison_bit(void) { return kPoison; }

extern "C" int spr_p2p_flags_set(void *const *d_flags, int32_t n, uint64_t value, void *stream) {
  FlagTable t;
  if (int rc = fill_table(t, d_flags, n, "spr_p2p_flags_set")) return rc;
  return launch_set(t, n, value, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int spr_p2p_flags_wait(void *const *d_flags, int32_t n, uint64_t value, double timeout_s, void *d_status,
                                  void *stream) {
  FlagTable t;
  if (int rc = fill_table(t, d_flags, n, "spr_p2p_flags_wait")) return rc;
  SPR_REQUIRE(timeout_s > 0.0 && timeout_s <= 3600.0, SPR_E_INVALID, "spr_p2p_flags_wait: timeout_s=%g (0, 3600]", timeout_s);
  SPR_REQUIRE(d_status == nullptr || (uintptr_t)d_status % 8 == 0, SPR_E_INVALID, "spr_p2p_flags_wait: unaligned status");
  return launch_wait(t, n, value, timeout_s, d_status, static_cast<hipStream_t>(stream));
}

extern "C" int spr_field_gather_p2p(const double *d_field, int64_t ldo, int32_t n_p, int64_t first, int64_t n_loc,
                                    int32_t n_peers, void *const *d_peer_field, void *const *d_release_flag,
                                    uint64_t release_value, double release_timeout_s, void *const *d_peer_arrive_flag,
                                    uint64_t arrive_value, void *const *d_pushed_flag, void *const *streams, void *d_status,
                                    void *d_ready_flag, uint64_t ready_value) {
  SPR_REQUIRE(d_field && n_p >= 1 && first >= 0 && n_loc >= 0 && ldo >= first + n_loc, SPR_E_INVALID,
              "spr_field_gather_p2p: n_p=%d first=%lld n_loc=%lld ldo=%lld", n_p, (long long)first, (long long)n_loc,
              (long long)ldo);
  SPR_REQUIRE(n_peers >= 1 && 2 * n_peers <= kMaxFlags && d_peer_field && d_peer_arrive_flag && streams, SPR_E_INVALID,
              "spr_field_gather_p2p: n_peers=%d (1..%d) / NULL peer table", n_peers, kMaxFlags / 2);
  SPR_REQUIRE(release_value == 0 || (d_release_flag && release_timeout_s > 0.0 && release_timeout_s <= 3600.0), SPR_E_INVALID,
              "spr_field_gather_p2p: release flags missing / release_timeout_s=%g outside (0, 3600]", release_timeout_s);
  SPR_REQUIRE(d_status == nullptr || (uintptr_t)d_status % 8 == 0, SPR_E_INVALID, "spr_field_gather_p2p: unaligned status");
  SPR_REQUIRE((uintptr_t)d_ready_flag % 8 == 0 && (!d_ready_flag || (ready_value >= release_value && ready_value - release_value < 65536
                                                                      && release_timeout_s > 0.0 && release_timeout_s <= 3600.0)),
              SPR_E_INVALID, "spr_field_gather_p2p: unaligned ready counter / ready_value=%llu not in [release_value, release_value + 65535]",
              (unsigned long long)ready_value);
  for (int p = 0; p < n_peers; ++p) {
    SPR_REQUIRE(d_peer_field[p] && d_peer_arrive_flag[p] && (uintptr_t)d_peer_arrive_flag[p] % 8 == 0, SPR_E_INVALID,
                "spr_field_gather_p2p: peer %d has a NULL / unaligned pointer", p);
    SPR_REQUIRE(!release_value || (d_release_flag[p] && (uintptr_t)d_release_flag[p] % 8 == 0), SPR_E_INVALID,
                "spr_field_gather_p2p: release counter %d is NULL or unaligned", p);
    SPR_REQUIRE(!d_pushed_flag || !d_pushed_flag[p] || (uintptr_t)d_pushed_flag[p] % 8 == 0, SPR_E_INVALID,
                "spr_field_gather_p2p: pushed counter %d is unaligned", p);
  }
  const hipMemcpyKind kind = p2p_kind();
  const bool probe = p2p_probe();
  double t_prev = probe ? host_us() : 0.0;
  auto lap = [&](const char *what, int p) {
    if (!probe) return;
    const double t = host_us();
    fprintf(stderr, "[p2p probe] %-8s peer %2d  %8.1f us\n", what, p, t - t_prev);
    t_prev = t;
  };
  // The peers of ONE copy stream are served together: one wait kernel for their releases, their copies, one kernel that raises
  // their arrival (and this rank's pushed) counters -- 2 launches per stream instead of 3 per peer (7 peers on 3 streams:
  // 6 + 7 n_p calls instead of 21 + 7 n_p; 0.34 ms of host time per gather before).  A peer's arrival is announced when the
  // stream's last copy has landed; the join needs every block anyway.
  bool done[kMaxFlags] = {false};
  for (int p0 = 0; p0 < n_peers; ++p0) {
    if (done[p0]) continue;
    hipStream_t st = static_cast<hipStream_t>(streams[p0]);
    FlagTable rel, arr;
    int n_rel = 0, n_arr = 0;
    for (int p = p0; p < n_peers; ++p) {
      if (done[p] || static_cast<hipStream_t>(streams[p]) != st) continue;
      if (release_value) {
        rel.p[n_rel] = static_cast<unsigned long long *>(d_release_flag[p]);
        rel.add[n_rel] = 0;
        rel.id[n_rel++] = (unsigned short)p;
      }
      arr.p[n_arr] = static_cast<unsigned long long *>(d_peer_arrive_flag[p]);
      arr.add[n_arr] = 0;
      arr.id[n_arr++] = (unsigned short)p;
      if (d_pushed_flag && d_pushed_flag[p]) {   // "my push to peer p has left": a counter of THIS rank, raised behind the copies
        arr.p[n_arr] = static_cast<unsigned long long *>(d_pushed_flag[p]);
        arr.add[n_arr] = 0;
        arr.id[n_arr++] = (unsigned short)p;
      }
    }
    if (d_ready_flag) {
      // "the block is written": a counter of THIS rank that the caller raises on its compute stream behind the kernel that
      // wrote the block (spr_p2p_flags_set: a system-scope release) -- awaited here instead of an event of the compute stream:
      // one event record + one stream wait per copy stream cost 0.25 ms of host time per push while the compute stream was busy
      // (profiles/r06_p2p_push_host.txt).  Index n_peers in the status words if it never comes.
      rel.p[n_rel] = static_cast<unsigned long long *>(d_ready_flag);
      rel.add[n_rel] = (unsigned short)(ready_value - release_value);
      rel.id[n_rel++] = (unsigned short)n_peers;
    }
    if (n_rel) {
      // the peers must have let go of what their buffer held (each raises its slot when it enters its own gather).  The wait
      // is the library's own single-wave kernel, not hipStreamWaitValue64: that one polls without an exit, and a copy stream
      // left behind a peer that has died would spin until the process is killed -- every wave must reach its exit.  A wait
      // that gives up (release_timeout_s, or a poisoned counter) leaves (peer index + 1, value seen) at d_status; the copies
      // below cannot be taken back (an SDMA command is unconditional), but the arrival counters behind them are then raised
      // WITH the poison bit: neither the peer's join nor this rank's own can succeed -- both raise, no field is handed out.
      if (int rc = launch_wait(rel, n_rel, release_value, release_timeout_s, d_status, st)) return rc;
      lap("wait", p0);
    }
    for (int p = p0; p < n_peers; ++p) {
      if (done[p] || static_cast<hipStream_t>(streams[p]) != st) continue;
      done[p] = true;
      if (n_loc > 0) {
        double *dst = static_cast<double *>(d_peer_field[p]);
        if (n_p == 1 || ldo == n_loc) {          // one contiguous piece
          SPR_HIP_TRY(hipMemcpyAsync(dst + first, d_field + first, (size_t)n_loc * 8 * (ldo == n_loc ? n_p : 1), kind, st));
        } else {
          for (int v = 0; v < n_p; ++v)          // contiguous pieces: one row of the (n_p, ldo) field each
            SPR_HIP_TRY(hipMemcpyAsync(dst + (int64_t)v * ldo + first, d_field + (int64_t)v * ldo + first, (size_t)n_loc * 8, kind, st));
        }
        lap("copy", p);
      }
    }
    if (int rc = launch_set(arr, n_arr, arrive_value, d_status, st)) return rc;
    lap("arrive", p0);
  }
  return SPR_OK;
}

extern "C" int spr_field_gather_p2p_join(void *const *d_flags, int32_t n_flags, uint64_t arrive_value, double timeout_s,
                                         void *d_status, void *stream) {
  return spr_p2p_flags_wait(d_flags, n_flags, arrive_value, timeout_s, d_status, stream);
}

extern "C" int spr_field_gather_p2p_release(void *const *d_peer_release_flag, int32_t n_peers, uint64_t value, void *stream) {
  return spr_p2p_flags_set(d_peer_release_flag, n_peers, value, stream);
}
