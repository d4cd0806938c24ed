"""AddressSanitizer run of the HOST side of the C-ABI layer: `make asan` builds every source
--cuda-host-only with -fsanitize=address (about 10 s; device code objects replaced by placeholders, so the library can
validate arguments and be linked against but never launch), then the C-ABI tests and a sweep of every entry point with
invalid arguments run against it in a child interpreter with the ASan runtime preloaded.  CPU only -- GPU ASan is not
available on this pool, and this build cannot launch kernels anyway."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_LIB = os.path.join(ROOT, 'build', 'asan', 'libspr_hip_asan.so')

SWEEP = r'''
import ctypes as C, sys
sys.path.insert(0, %(root)r)
from openmeasure_amd import _lib
assert _lib.LIB_PATH.endswith('libspr_hip_asan.so'), _lib.LIB_PATH
lib = _lib.load()
assert lib.spr_abi_version() == _lib.SPR_ABI_VERSION
n = 0
for name, (res, args) in sorted(_lib.PROTOTYPES.items()):
    if res is not C.c_int or not args or name in ('spr_abi_version', 'spr_device_cus', 'spr_project_norms_supported', 'spr_qr_epoch_supported',
                                                 'spr_qr_epoch_max_directions'):   # yes/no and size queries
        continue
    if name.startswith('spr_p2p_'):           # raw pointers handed straight to the HIP runtime (allocation, IPC mapping, stream
        continue                              # memory operations): no shape to validate, and this build has no runtime to call
    if name.startswith('spr_comm_'):          # HOST pointers (the 128 bytes of the unique id) that are read / written by design, and
        continue                              # calls into RCCL; the handle look-up of the collectives IS swept below (spr_allreduce_*, ...)
    fn = getattr(lib, name)
    # all-zero arguments: NULL pointers and empty shapes must be rejected by the validation layer, with a message
    zero = [C.c_void_p(None) if a is C.c_void_p else a(0) for a in args]
    rc = fn(*zero)
    assert rc in (-1, -2), (name, rc)
    msg = lib.spr_last_error()
    assert msg and len(msg) > 4, name
    # plausible pointers (never dereferenced on the host) but impossible shapes: negative sizes
    bad = [C.c_void_p(4096) if a is C.c_void_p else a(-3) if a in (C.c_int32, C.c_int64) else a(0) for a in args]
    rc = fn(*bad)
    assert rc in (-1, -2), (name, rc)
    n += 1
# error text is thread-local and survives long format arguments
rc = lib.spr_project_f64(C.c_void_p(8), 10, 300, 300, 0, 10, 1, 0, C.c_void_p(8), None, C.c_void_p(8), 4, C.c_void_p(8), 4, 0, None)
assert rc == -2 and b'300' in lib.spr_last_error()
print('asan sweep ok:', n, 'entry points')
'''


@pytest.mark.skipif(shutil.which('hipcc') is None or shutil.which('make') is None, reason='hipcc/make not available')
def test_c_abi_host_layer_under_address_sanitizer(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip('CPU-box check (the ASan build cannot launch kernels)')
    subprocess.run(['make', '-C', os.path.join(ROOT, 'openmeasure_amd', 'csrc'), 'asan', '-j8'], check=True,
                   capture_output=True)
    rt = subprocess.run(['/opt/rocm/lib/llvm/bin/clang', '-print-file-name=libclang_rt.asan-x86_64.so'],
                        capture_output=True, text=True, check=True).stdout.strip()
    assert os.path.exists(rt) and os.path.exists(ASAN_LIB)
    env = dict(os.environ, LD_PRELOAD=rt, SPR_HIP_LIBRARY=ASAN_LIB,
               ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=87:halt_on_error=1')
    # 1. the C-ABI tests against the instrumented library
    p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-p', 'no:cacheprovider', 'tests/test_cabi.py',
                        'tests/test_c_linkage.py::test_header_is_plain_c_and_library_links'], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert 'AddressSanitizer' not in p.stdout + p.stderr
    # 2. every int-returning entry point with NULL / zero / negative arguments
    q = subprocess.run([sys.executable, '-c', SWEEP % dict(root=ROOT)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert q.returncode == 0, (q.stdout[-3000:], q.stderr[-3000:])
    assert 'asan sweep ok' in q.stdout and 'AddressSanitizer' not in q.stderr
