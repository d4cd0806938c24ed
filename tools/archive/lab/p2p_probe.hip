// GPU box probe (round 5): what a CU-free field gather can be built from on this pool.
//   two processes on device 0 (forked BEFORE either touches HIP), IPC-mapped buffers, hipMemcpyDeviceToDeviceNoCU copies,
//   stream write/wait values on three kinds of flag memory.   hipcc --offload-arch=gfx950 -O2 -o p2p_probe p2p_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <unistd.h>
#include <sys/socket.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <fcntl.h>
#include <chrono>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[r%d] %s -> %s (line %d)\n", g_rank, #x, hipGetErrorString(e_), __LINE__); fflush(stdout); } } while (0)
static int g_rank = -1;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void spin_fma(double* out, long iters) {
  double a = threadIdx.x * 1e-3, b = 1.000001, c = 1e-9;
  for (long i = 0; i < iters; ++i) { a = a * b + c; b = b * 0.9999999 + 1e-7; }
  if (a == 12345.678) out[0] = a + b;
}
__global__ void fill(uint64_t* p, size_t n, uint64_t v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + i;
}
__global__ void check(const uint64_t* p, size_t n, uint64_t v, unsigned long long* bad) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (p[i] != v + i) atomicAdd(bad, 1ull);
}
__global__ void wait_flag(volatile uint64_t* f, uint64_t want, unsigned long long* result) {
  unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load((uint64_t*)f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    if (wall_clock64() - t0 > 300000000ull) { *result = 2; return; }   // 100 MHz counter: 3 s
    __builtin_amdgcn_s_sleep(32);
  }
  *result = 1;
}
__global__ void set_flag(uint64_t* f, uint64_t v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

static bool stream_done_within(hipStream_t s, double sec) {
  double t0 = now();
  while (now() - t0 < sec) { if (hipStreamQuery(s) == hipSuccess) return true; std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
  return false;
}
static void xsend(int fd, const void* p, size_t n) { if (write(fd, p, n) != (ssize_t)n) { perror("write"); _exit(3); } }
static void xrecv(int fd, void* p, size_t n) { size_t g = 0; while (g < n) { ssize_t k = read(fd, (char*)p + g, n - g); if (k <= 0) { perror("read"); _exit(3); } g += k; } }
static void sync_peer(int fd) { char c = 1; xsend(fd, &c, 1); xrecv(fd, &c, 1); }

int child(int rank, int fd, uint64_t* shm) {
  g_rank = rank;
  CK(hipSetDevice(0));
  int can = -1; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("[r%d] CanUseStreamWaitValue = %d\n", rank, can);
  const size_t bytes = 512ull << 20, n64 = bytes / 8;
  uint64_t *src, *stage; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&stage, bytes));
  hipStream_t comp, comm; CK(hipStreamCreateWithFlags(&comp, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&comm, hipStreamNonBlocking));
  unsigned long long* bad; CK(hipHostMalloc(&bad, 64)); bad[0] = 0;
  fill<<<1024, 256, 0, comp>>>(src, n64, 1000 * (rank + 1)); CK(hipStreamSynchronize(comp));
  // ---- 1. same-process copies: blit vs NoCU
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int kind = 0; kind < 2; ++kind) {
    hipMemcpyKind k = kind ? hipMemcpyDeviceToDeviceNoCU : hipMemcpyDeviceToDevice;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, comm)); CK(hipMemcpyAsync(stage, src, bytes, k, comm)); CK(hipEventRecord(e1, comm)); CK(hipStreamSynchronize(comm));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rank == 0) printf("[r0] local D2D %s 512 MiB: %.3f ms = %.1f GB/s\n", kind ? "NoCU" : "default", ms, bytes / ms / 1e6);
    }
  }
  // ---- 2. compute kernel alone, then with a NoCU / default copy in flight (rank 0 only; rank 1 idles)
  sync_peer(fd);
  if (rank == 0) {
    double* o; CK(hipMalloc(&o, 8));
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t k0, k1, c0, c1; hipEventCreate(&k0); hipEventCreate(&k1); hipEventCreate(&c0); hipEventCreate(&c1);
        CK(hipEventRecord(k0, comp)); spin_fma<<<256 * 8, 256, 0, comp>>>(o, 400000); CK(hipEventRecord(k1, comp));
        if (mode) { CK(hipEventRecord(c0, comm)); CK(hipMemcpyAsync(stage, src, bytes, mode == 1 ? hipMemcpyDeviceToDeviceNoCU : hipMemcpyDeviceToDevice, comm)); CK(hipEventRecord(c1, comm)); }
        CK(hipDeviceSynchronize());
        float km = 0, cm = 0; hipEventElapsedTime(&km, k0, k1); if (mode) hipEventElapsedTime(&cm, c0, c1);
        printf("[r0] spin kernel %.3f ms  | concurrent copy (%s): %.3f ms\n", km, mode == 0 ? "none" : mode == 1 ? "NoCU" : "default blit", cm);
      }
    }
  }
  sync_peer(fd);
  // ---- 3. IPC: export stage, open the peer's
  hipIpcMemHandle_t mine, theirs; CK(hipIpcGetMemHandle(&mine, stage));
  xsend(fd, &mine, sizeof mine); xrecv(fd, &theirs, sizeof theirs);
  uint64_t* peer = nullptr; hipError_t eo = hipIpcOpenMemHandle((void**)&peer, theirs, hipIpcMemLazyEnablePeerAccess);
  printf("[r%d] hipIpcOpenMemHandle -> %s, ptr %p\n", rank, hipGetErrorString(eo), (void*)peer);
  if (eo != hipSuccess) { fflush(stdout); return 1; }
  // flags: (a) host shm registered, (b) signal memory (own process only), (c) plain device memory inside the IPC buffer (last 4 KB)
  uint64_t* shm_dev = nullptr; hipError_t er = hipHostRegister(shm, 4096, hipHostRegisterMapped);
  printf("[r%d] hipHostRegister(shm) -> %s\n", rank, hipGetErrorString(er));
  if (er == hipSuccess) CK(hipHostGetDevicePointer((void**)&shm_dev, shm, 0));
  uint64_t* sig = nullptr; hipError_t es = hipExtMallocWithFlags((void**)&sig, 4096, hipMallocSignalMemory);
  printf("[r%d] hipExtMallocWithFlags(signal) -> %s\n", rank, hipGetErrorString(es));
  if (es == hipSuccess) { hipIpcMemHandle_t hs; hipError_t e2 = hipIpcGetMemHandle(&hs, sig); printf("[r%d] IpcGetMemHandle(signal mem) -> %s\n", rank, hipGetErrorString(e2)); }
  uint64_t* fg = nullptr; hipError_t ef = hipExtMallocWithFlags((void**)&fg, 4096, hipDeviceMallocFinegrained);
  printf("[r%d] hipExtMallocWithFlags(finegrained) -> %s\n", rank, hipGetErrorString(ef));
  if (ef == hipSuccess) { hipIpcMemHandle_t hs; hipError_t e2 = hipIpcGetMemHandle(&hs, fg); printf("[r%d] IpcGetMemHandle(finegrained) -> %s\n", rank, hipGetErrorString(e2)); }
  sync_peer(fd);
  // ---- 4. cross-process push + flag, three flag mechanisms.  rank 0 pushes, rank 1 waits and checks; then roles swap
  uint64_t* my_flag_dev = stage + n64 - 512;          // inside my exported buffer
  uint64_t* peer_flag_dev = peer + n64 - 512;
  CK(hipMemset(my_flag_dev, 0, 4096)); CK(hipDeviceSynchronize());
  sync_peer(fd);
  const size_t push = 256ull << 20, pn = push / 8;
  for (int mech = 0; mech < 4; ++mech) {
    const char* names[4] = {"writeValue64 -> host shm / waitValue64 on host shm", "writeValue64 -> peer device mem / waitValue64 on own device mem",
                            "writeValue64 -> host shm / wait KERNEL polling host shm", "set KERNEL -> peer device mem / wait KERNEL polling own device mem"};
    for (int pusher = 0; pusher < 2; ++pusher) {
      uint64_t step = 10 * (mech + 1) + pusher + 1;
      uint64_t* shm_slot = shm + 8 * mech + pusher;            // host view
      uint64_t* shm_slot_dev = shm_dev ? shm_dev + 8 * mech + pusher : nullptr;
      sync_peer(fd);
      if (rank == pusher) {
        fill<<<1024, 256, 0, comp>>>(src, n64, 7777 * step); CK(hipStreamSynchronize(comp));
        double t0 = now();
        CK(hipMemcpyAsync(peer, src, push, hipMemcpyDeviceToDeviceNoCU, comm));
        hipError_t ew = hipSuccess;
        if (mech == 0 || mech == 2) ew = shm_slot_dev ? hipStreamWriteValue64(comm, shm_slot_dev, step, 0) : hipErrorInvalidValue;
        else if (mech == 1) ew = hipStreamWriteValue64(comm, peer_flag_dev + mech, step, 0);
        else { set_flag<<<1, 1, 0, comm>>>(peer_flag_dev + mech, step); ew = hipGetLastError(); }
        bool ok = stream_done_within(comm, 5.0);
        printf("[r%d] mech %d (%s): push+flag issue -> %s, comm stream %s after %.3f ms\n", rank, mech, names[mech], hipGetErrorString(ew), ok ? "done" : "NOT DONE in 5 s", 1e3 * (now() - t0));
        fflush(stdout);
        if (!ok) _exit(7);
      } else {
        unsigned long long* res = bad + 1; *res = 0; bad[0] = 0;
        hipError_t ew = hipSuccess;
        double t0 = now();
        if (mech == 0) ew = shm_slot_dev ? hipStreamWaitValue64(comp, shm_slot_dev, step, hipStreamWaitValueGte) : hipErrorInvalidValue;
        else if (mech == 1) ew = hipStreamWaitValue64(comp, my_flag_dev + mech, step, hipStreamWaitValueGte);
        else if (mech == 2) { if (shm_slot_dev) wait_flag<<<1, 1, 0, comp>>>(shm_slot_dev, step, res); else ew = hipErrorInvalidValue; }
        else wait_flag<<<1, 1, 0, comp>>>(my_flag_dev + mech, step, res);
        if (ew == hipSuccess) check<<<1024, 256, 0, comp>>>(stage, pn, 7777 * step, bad);
        bool ok = stream_done_within(comp, 6.0);
        printf("[r%d] mech %d: wait issue -> %s, compute stream %s after %.3f ms, wait-kernel verdict %llu, mismatching words %llu of %zu (host view of shm slot: %llu)\n",
               rank, mech, hipGetErrorString(ew), ok ? "done" : "NOT DONE in 6 s", 1e3 * (now() - t0), *res, bad[0], pn, (unsigned long long)*shm_slot);
        fflush(stdout);
        if (!ok) _exit(8);
      }
    }
  }
  sync_peer(fd);
  // ---- 5. rate of a cross-process NoCU push (rank 0 -> rank 1's buffer)
  if (rank == 0) for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, comm)); CK(hipMemcpyAsync(peer, src, push, hipMemcpyDeviceToDeviceNoCU, comm)); CK(hipEventRecord(e1, comm)); CK(hipStreamSynchronize(comm));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); printf("[r0] NoCU push into the peer process's buffer, 256 MiB: %.3f ms = %.1f GB/s\n", ms, push / ms / 1e6);
  }
  sync_peer(fd);
  CK(hipIpcCloseMemHandle(peer));
  sync_peer(fd);
  CK(hipFree(src)); CK(hipFree(stage));
  printf("[r%d] done\n", rank); fflush(stdout);
  return 0;
}

int main() {
  int sv[2]; if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { perror("socketpair"); return 1; }
  uint64_t* shm = (uint64_t*)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  memset(shm, 0, 4096);
  pid_t p0 = fork(); if (p0 == 0) { close(sv[1]); _exit(child(0, sv[0], shm)); }
  pid_t p1 = fork(); if (p1 == 0) { close(sv[0]); _exit(child(1, sv[1], shm)); }
  int st0 = 0, st1 = 0; waitpid(p0, &st0, 0); waitpid(p1, &st1, 0);
  printf("exit codes: %d %d\n", WEXITSTATUS(st0), WEXITSTATUS(st1));
  return (WEXITSTATUS(st0) || WEXITSTATUS(st1)) ? 1 : 0;
}
