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
  if (spr_abi_version() != 1) return 1;
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
    lib = os.path.join(ROOT, 'openmeasure_amd', 'libspr_hip.so')
    assert os.path.exists(lib)
    src = tmp_path / 't.c'
    src.write_text(C_SRC)
    exe = tmp_path / 't'
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe),
                    lib, '-Wl,-rpath,' + os.path.dirname(lib), '-Wl,-rpath,/opt/rocm/lib'], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert 'qr batch' in out.stdout
