// Error text, ABI version and device queries of libspr_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace {
thread_local char g_err[512] = "";
}

void spr_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int spr_cached_cus() {
  static int cus = 0;
  if (cus) return cus;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  cus = prop.multiProcessorCount;
  // SPR_RESERVE_CUS=k: size every persistent grid for k compute units fewer.  A diagnostic for multi-GPU runs: RCCL's device
  // kernel (261-280 VGPRs per wave, 19.7 KB of LDS) cannot share a CU with the Gram or projection workgroups, so a field
  // gather left in flight only runs beside them on CUs they do not occupy.
  if (const char *e = getenv("SPR_RESERVE_CUS")) {
    const int k = atoi(e);
    if (k > 0 && k < cus) cus -= k;
  }
  return cus;
}

extern "C" int spr_abi_version(void) { return SPR_ABI_VERSION; }

extern "C" const char *spr_last_error(void) { return g_err; }

extern "C" int spr_device_cus(int *out_cus) {
  SPR_REQUIRE(out_cus != nullptr, SPR_E_INVALID, "spr_device_cus: NULL output");
  const int c = spr_cached_cus();
  SPR_REQUIRE(c > 0, SPR_E_HIP, "spr_device_cus: no HIP device");
  *out_cus = c;
  return SPR_OK;
}

// Small host -> device upload as a kernel on the caller's stream.  A hipMemcpyAsync between two kernels puts a
// cross-queue dependency (barrier packet on the copy's signal) in front of the next kernel; when that signal is
// not yet complete as the command processor reaches the barrier, the queue was measured to resume 10, 20 or 30 ms
// later (tools/fit_probe.py: gap between the Gram and the projection kernel 3.6 ms or 14/25/34 ms).  Reading the
// pinned host buffer from a kernel keeps everything on one queue.
namespace {
__global__ __launch_bounds__(256) void upload_kernel(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst,
                                                     int64_t n_words) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * 256)
    dst[i] = __builtin_nontemporal_load(src + i);
}
}  // namespace

extern "C" int spr_upload_bytes(void *d_dst, const void *h_pinned_src, int64_t n_bytes, void *stream) {
  SPR_REQUIRE(d_dst && h_pinned_src, SPR_E_INVALID, "spr_upload_bytes: NULL pointer");
  SPR_REQUIRE(n_bytes > 0 && n_bytes % 8 == 0, SPR_E_INVALID, "spr_upload_bytes: n_bytes=%lld must be a positive multiple of 8",
              (long long)n_bytes);
  SPR_REQUIRE(((uintptr_t)d_dst | (uintptr_t)h_pinned_src) % 8 == 0, SPR_E_INVALID, "spr_upload_bytes: unaligned pointer");
  const int64_t words = n_bytes / 8;
  int64_t blocks = (words + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(upload_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint64_t *>(h_pinned_src), static_cast<uint64_t *>(d_dst), words);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}

// Small device -> host download as a kernel: the mirror of spr_upload_bytes.  The words go straight into page-locked,
// device-visible host memory and a 64-bit ticket is stored behind them (system-scope release), so the host can poll the
// ticket in its own memory instead of blocking on an event -- between the Gram pass and the projection of a small fit()
// (BASELINE config 2: 0.3 ms of a 1.5 ms step) the copy-engine launch, the event and the wake-up of a blocked thread are a
// tenth of the gap.  One workgroup: the payload is a few tens of KB (the m x m Gram matrix and 5 F statistics).
namespace {
__global__ __launch_bounds__(1024) void download_kernel(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst,
                                                        int64_t n_words, unsigned long long *__restrict__ ticket,
                                                        unsigned long long value) {
  for (int64_t i = threadIdx.x; i < n_words; i += 1024) __builtin_nontemporal_store(src[i], dst + i);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: every lane's stores are on their way before the barrier
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(ticket, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

extern "C" int spr_download_bytes(void *h_pinned_dst, const void *d_src, int64_t n_bytes, void *h_pinned_ticket,
                                  uint64_t ticket_value, void *stream) {
  SPR_REQUIRE(h_pinned_dst && d_src && h_pinned_ticket, SPR_E_INVALID, "spr_download_bytes: NULL pointer");
  SPR_REQUIRE(n_bytes > 0 && n_bytes % 8 == 0 && n_bytes <= (int64_t)1 << 22, SPR_E_INVALID,
              "spr_download_bytes: n_bytes=%lld must be a positive multiple of 8 up to 4 MiB", (long long)n_bytes);
  SPR_REQUIRE(((uintptr_t)h_pinned_dst | (uintptr_t)d_src | (uintptr_t)h_pinned_ticket) % 8 == 0, SPR_E_INVALID,
              "spr_download_bytes: unaligned pointer");
  hipLaunchKernelGGL(download_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint64_t *>(d_src), static_cast<uint64_t *>(h_pinned_dst), n_bytes / 8,
                     static_cast<unsigned long long *>(h_pinned_ticket), (unsigned long long)ticket_value);
  SPR_LAUNCH_CHECK();
  return SPR_OK;
}
