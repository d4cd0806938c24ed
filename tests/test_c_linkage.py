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
