"""The library is a plain C ABI: a C translation unit that includes include/spr_hip.h compiles
with gcc (no C++ / HIP headers needed) and links against libspr_hip.so."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SRC = r'''
#include <stdio.h>
#include <string.h>
#include "spr_hip.h"
int main(void) {
  if (spr_abi_version() != SPR_ABI_VERSION) return 1;
  /* argument validation happens before any device work */
  int rc = spr_reconstruct_f64(NULL, 10, 4, 4, 0, 10, 1, NULL, NULL, NULL, NULL, 1, NULL, 10, NULL);
  if (rc != SPR_E_INVALID) return 2;
  if (!strstr(spr_last_error(), "NULL")) return 3;
  rc = spr_project_f64((const double *)8, 10, 300, 300, 0, 10, 1, 0, (const double *)8, NULL, (const double *)8, 4,
                       (double *)8, 4, 0, NULL);
  if (rc != SPR_E_UNSUPPORTED) return 4;           /* m = 300 > SPR_MAX_M in ONE launch: refused, not mis-computed */
  /* round 6: the collectives behind the ABI -- a handle is looked up, never dereferenced on trust */
  if (spr_comm_unique_id_bytes() != 128) return 5;
  if (spr_comm_destroy((void *)4096) != SPR_E_INVALID || spr_allreduce_f64(NULL, (double *)8, 4, NULL) != SPR_E_INVALID) return 6;
  if (spr_fit_gram_pass_buffer(256, 9, 8) != (size_t)(9 * 256 * 256 + 8 * 9 * 3 + 8) * 8) return 7;
  rc = spr_fit_gram_pass(NULL, NULL, 0, 10, 4, 4, 0, 10, 1, 0, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, 0, NULL);
  if (rc != SPR_E_INVALID) return 8;
  printf("qr batch %d, workspace %zu\n", (int)spr_qr_batch(), spr_qr_workspace(1000));
  return 0;
}
'''


@pytest.mark.skipif(shutil.which('gcc') is None, reason='gcc not available')
def test_header_is_plain_c_and_library_links(tmp_path):
    lib = os.environ.get('SPR_HIP_LIBRARY') or os.path.join(ROOT, 'openmeasure_amd', 'libspr_hip.so')
    assert os.path.exists(lib)
    src = tmp_path / 't.c'
    src.write_text(C_SRC)
    exe = tmp_path / 't'
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe),
                    lib, '-Wl,-rpath,' + os.path.dirname(lib), '-Wl,-rpath,/opt/rocm/lib']
                   # the ASan build (tests/test_asan_host.py) needs the sanitizer runtime, which is preloaded at run time
                   + (['-Wl,--allow-shlib-undefined'] if 'SPR_HIP_LIBRARY' in os.environ else []), check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert 'qr batch' in out.stdout


# A caller with no Python and no torch in the process: plain C against the HIP runtime's C API, the snapshot block
# in hipMalloc'ed memory, the library's Gram pass and reconstruction checked against loops on the host.
C_GPU_SRC = r'''
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>
#include "spr_hip.h"
#define CK(x) do { if ((x) != 0) { printf("fail %s line %d: %s\n", #x, __LINE__, spr_last_error()); return 1; } } while (0)
int main(void) {
  const int64_t n_points = 2048, n = 4096; const int F = 2, m = 48, r = 6;
  double *X = (double *)malloc(sizeof(double) * n * m);
  uint64_t s = 12345;
  for (int64_t i = 0; i < n * m; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; X[i] = (double)(s >> 11) / 9007199254740992.0 + (i / (n_points * m)); }
  double *dX, *dmean, *dfs, *dgram; void *ws;
  size_t wsb = spr_stats_gram_workspace(m, F);
  CK(hipMalloc((void **)&dX, sizeof(double) * n * m)); CK(hipMalloc((void **)&dmean, sizeof(double) * n));
  CK(hipMalloc((void **)&dfs, sizeof(double) * F * 3)); CK(hipMalloc((void **)&dgram, sizeof(double) * F * m * m));
  CK(hipMalloc(&ws, wsb));
  CK(hipMemcpy(dX, X, sizeof(double) * n * m, hipMemcpyHostToDevice));
  CK(spr_stats_gram_f64(dX, n, m, m, 0, n_points, F, 1, dmean, ws, wsb, NULL));
  CK(spr_stats_gram_finalize_f64(n, m, 0, n_points, F, ws, wsb, dfs, dgram, m, 0, NULL));
  double *G = (double *)malloc(sizeof(double) * F * m * m), *mean = (double *)malloc(sizeof(double) * n);
  CK(hipMemcpy(G, dgram, sizeof(double) * F * m * m, hipMemcpyDeviceToHost));
  CK(hipMemcpy(mean, dmean, sizeof(double) * n, hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int f = 0; f < F; ++f)
    for (int a = 0; a < m; a += 7)
      for (int b = 0; b < m; b += 5) {
        double ref = 0.0;
        for (int64_t i = f * n_points; i < (f + 1) * n_points; ++i) {
          double mu = 0.0; for (int c = 0; c < m; ++c) mu += X[i * m + c]; mu /= m;
          ref += (X[i * m + a] - mu) * (X[i * m + b] - mu);
        }
        const double d = fabs(G[((int64_t)f * m + a) * m + b] - ref) / (fabs(ref) + 1.0);
        if (d > worst) worst = d;
      }
  if (worst > 1e-12) { printf("gram mismatch %g\n", worst); return 2; }
  /* reconstruction x = scale * (U a) + mean on the first r columns of X used as a basis */
  double *dU, *dA, *dsc, *dout; double a[6] = {1, -2, 0.5, 3, -1, 0.25}, sc[2] = {2.0, 0.5};
  CK(hipMalloc((void **)&dU, sizeof(double) * n * r)); CK(hipMalloc((void **)&dA, sizeof(a)));
  CK(hipMalloc((void **)&dsc, sizeof(sc))); CK(hipMalloc((void **)&dout, sizeof(double) * n));
  CK(hipMemcpy2D(dU, sizeof(double) * r, X, sizeof(double) * m, sizeof(double) * r, n, hipMemcpyHostToDevice));
  CK(hipMemcpy(dA, a, sizeof(a), hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc, sizeof(sc), hipMemcpyHostToDevice));
  CK(spr_reconstruct_f64(dU, n, r, r, 0, n_points, F, dmean, dsc, NULL, dA, 1, dout, n, NULL));
  double *out = (double *)malloc(sizeof(double) * n);
  CK(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; i += 97) {
    double d = 0.0; for (int k = 0; k < r; ++k) d += X[i * m + k] * a[k];
    const double ref = sc[i / n_points] * d + mean[i];
    if (fabs(out[i] - ref) > 1e-12 * (fabs(ref) + 1.0)) { printf("reconstruct mismatch at %lld\n", (long long)i); return 3; }
  }
  printf("C caller ok: gram rel err %.2e\n", worst);
  return 0;
}
'''


@pytest.mark.gpu
def test_plain_c_caller_on_the_gpu(tmp_path):
    """No Python, no torch in the calling process: a C program (hip_runtime_api.h + spr_hip.h) drives the Gram pass
    and the reconstruction through the C ABI on hipMalloc'ed memory and checks them against host loops."""
    if shutil.which('gcc') is None or not os.path.exists('/opt/rocm/include/hip/hip_runtime_api.h'):
        pytest.skip('gcc / HIP runtime headers not available')
    lib = os.path.join(ROOT, 'openmeasure_amd', 'libspr_hip.so')
    src = tmp_path / 'g.c'
    src.write_text(C_GPU_SRC)
    exe = tmp_path / 'g'
    subprocess.run(['gcc', '-std=gnu99', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(ROOT, 'include'),
                    '-I', '/opt/rocm/include', str(src), '-o', str(exe), lib, '-L/opt/rocm/lib', '-lamdhip64', '-lm',
                    '-Wl,-rpath,' + os.path.dirname(lib), '-Wl,-rpath,/opt/rocm/lib'], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert 'C caller ok' in out.stdout


# round 6: the SHARDED first pass of fit() from plain C -- the library's own communicator (RCCL reached through dlopen), one rank:
# spr_fit_gram_pass (Gram kernel + finalize + all-reduce + statistics merge in one enqueue) against the same steps called one by
# one without a communicator, bit for bit; spr_allreduce_f64 / spr_allgather on a one-rank communicator leave the data as it is.
C_COMM_SRC = r'''
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "spr_hip.h"
#define CK(x) do { if ((x) != 0) { printf("fail %s line %d: %s\n", #x, __LINE__, spr_last_error()); return 1; } } while (0)
int main(void) {
  const int64_t n_points = 3000, row0 = 1500, n = 4000; const int F = 3, m = 40, world = 1;   /* rows 1500..5499 of 9000: three features */
  double *X = (double *)malloc(sizeof(double) * n * m);
  uint64_t s = 777;
  for (int64_t i = 0; i < n * m; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; X[i] = (double)(s >> 11) / 9007199254740992.0 * (1.0 + (row0 + i / m) / n_points) + 3.0; }
  unsigned char id[128];
  void *comm = NULL;
  CK(spr_comm_unique_id(id));
  CK(spr_comm_init(id, 0, world, &comm));
  int32_t rank = -1, w = -1;
  CK(spr_comm_info(comm, &rank, &w));
  if (rank != 0 || w != 1) return 2;
  printf("RCCL from %s\n", spr_comm_library());
  size_t wsb = spr_fit_gram_pass_workspace(m, F, n), bufb = spr_fit_gram_pass_buffer(m, F, world);
  if (wsb != spr_stats_gram_workspace(m, F)) return 9;                    /* m <= 256: the Gram workspace */
  double *dX, *dmean, *dmean2, *dbuf, *dG, *dfeat, *dsc, *dinv, *dfs, *dgram, *dG2, *dfeat2, *dsc2, *dinv2; void *ws;
  CK(hipMalloc((void **)&dX, sizeof(double) * n * m)); CK(hipMalloc((void **)&dmean, sizeof(double) * n)); CK(hipMalloc((void **)&dmean2, sizeof(double) * n));
  CK(hipMalloc((void **)&dbuf, bufb)); CK(hipMalloc((void **)&dG, sizeof(double) * m * m)); CK(hipMalloc((void **)&dG2, sizeof(double) * m * m));
  CK(hipMalloc((void **)&dfeat, sizeof(double) * F * 5)); CK(hipMalloc((void **)&dfeat2, sizeof(double) * F * 5));
  CK(hipMalloc((void **)&dsc, sizeof(double) * F)); CK(hipMalloc((void **)&dinv, sizeof(double) * F));
  CK(hipMalloc((void **)&dsc2, sizeof(double) * F)); CK(hipMalloc((void **)&dinv2, sizeof(double) * F));
  CK(hipMalloc((void **)&dfs, sizeof(double) * F * 3)); CK(hipMalloc((void **)&dgram, sizeof(double) * F * m * m)); CK(hipMalloc(&ws, wsb));
  CK(hipMemcpy(dX, X, sizeof(double) * n * m, hipMemcpyHostToDevice));
  /* the sharded pass: one call */
  CK(spr_fit_gram_pass(comm, dX, 0, n, m, m, row0, n_points, F, 0, dmean, dbuf, bufb, dG, dfeat, dsc, dinv, ws, wsb, NULL));
  /* the same steps one by one, no communicator */
  CK(spr_stats_gram_f64(dX, n, m, m, row0, n_points, F, 1, dmean2, ws, wsb, NULL));
  CK(spr_stats_gram_finalize_f64(n, m, row0, n_points, F, ws, wsb, dfs, dgram, m, 0, NULL));
  CK(spr_gram_combine_f64(dgram, dfs, 1, F, m, 0, dG2, dfeat2, dsc2, dinv2, NULL));
  CK(hipDeviceSynchronize());
  double *a = (double *)malloc(sizeof(double) * m * m), *b = (double *)malloc(sizeof(double) * m * m);
  CK(hipMemcpy(a, dG, sizeof(double) * m * m, hipMemcpyDeviceToHost)); CK(hipMemcpy(b, dG2, sizeof(double) * m * m, hipMemcpyDeviceToHost));
  if (memcmp(a, b, sizeof(double) * m * m)) { printf("G differs\n"); return 3; }
  double f1[15], f2[15], sc1[3], sc2[3];
  CK(hipMemcpy(f1, dfeat, sizeof f1, hipMemcpyDeviceToHost)); CK(hipMemcpy(f2, dfeat2, sizeof f2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(sc1, dsc, sizeof sc1, hipMemcpyDeviceToHost)); CK(hipMemcpy(sc2, dsc2, sizeof sc2, hipMemcpyDeviceToHost));
  if (memcmp(f1, f2, sizeof f1) || memcmp(sc1, sc2, sizeof sc1)) { printf("statistics differ\n"); return 4; }
  /* what the buffer still says about the blocks: rows per feature of this rank (1500, 2500 of feature 1 -> 1500 + 3000 - ... ) and its first row */
  double *hb = (double *)malloc(bufb);
  CK(hipMemcpy(hb, dbuf, bufb, hipMemcpyDeviceToHost));
  const double *fs = hb + (size_t)F * m * m, *rows = fs + world * F * 3;
  if (fs[0] != 1500.0 || fs[3] != 2500.0 || fs[6] != 0.0 || rows[0] != (double)row0) { printf("slots: %g %g %g %g\n", fs[0], fs[3], fs[6], rows[0]); return 5; }
  if (!(f1[0] == 1500.0 && f1[5] == 2500.0 && sc1[0] > 0.0 && isfinite(a[0]))) return 6;
  /* the two collectives on their own */
  CK(spr_allreduce_f64(comm, dG, (int64_t)m * m, NULL));
  CK(spr_allgather(comm, dG, dG2, (int64_t)sizeof(double) * m * m, NULL));
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(b, dG2, sizeof(double) * m * m, hipMemcpyDeviceToHost));
  if (memcmp(a, b, sizeof(double) * m * m)) { printf("collectives changed a one-rank buffer\n"); return 7; }
  CK(spr_comm_destroy(comm));
  if (spr_comm_destroy(comm) != SPR_E_INVALID) return 8;                   /* a stale handle is refused, not followed */
  printf("C sharded fit pass ok: G[0] = %.17g, scales %g %g\n", a[0], sc1[0], sc1[1]);
  return 0;
}
'''


@pytest.mark.gpu
def test_plain_c_sharded_fit_pass_on_the_gpu(tmp_path):
    """the collectives behind the C ABI -- a C program with no Python and no torch in the process creates the
    library's communicator (one rank), runs fit()'s first pass WITH its all-reduce as one call and gets the bits of the
    unsharded steps."""
    if shutil.which('gcc') is None or not os.path.exists('/opt/rocm/include/hip/hip_runtime_api.h'):
        pytest.skip('gcc / HIP runtime headers not available')
    lib = os.path.join(ROOT, 'openmeasure_amd', 'libspr_hip.so')
    src = tmp_path / 'c.c'
    src.write_text(C_COMM_SRC)
    exe = tmp_path / 'c'
    subprocess.run(['gcc', '-std=gnu99', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(ROOT, 'include'),
                    '-I', '/opt/rocm/include', str(src), '-o', str(exe), lib, '-L/opt/rocm/lib', '-lamdhip64', '-lm',
                    '-Wl,-rpath,' + os.path.dirname(lib), '-Wl,-rpath,/opt/rocm/lib'], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=180,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    assert 'C sharded fit pass ok' in out.stdout and 'librccl' in out.stdout
