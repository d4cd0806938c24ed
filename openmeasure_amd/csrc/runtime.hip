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
