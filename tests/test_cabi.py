"""The C-ABI library builds, loads and exports every symbol include/spr_hip.h declares.
No compute call is made here (there is no GPU on the CPU test box)."""
import ctypes
import os
import re

import pytest

from openmeasure_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'spr_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(spr_[a-z0-9_]+)\s*\(', src)))


def test_library_exists():
    assert os.path.exists(_lib.LIB_PATH), 'run `make -C openmeasure_amd/csrc` (or __graft_entry__.build())'


def test_every_declared_symbol_is_exported_and_bound():
    names = _header_functions()
    assert len(names) >= 17
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f'{n} declared in spr_hip.h but not exported'
    assert sorted(_lib.PROTOTYPES) == names, 'ctypes prototypes drifted from the header'


def test_loader_and_error_text():
    lib = _lib.load()
    assert lib.spr_abi_version() == _lib.SPR_ABI_VERSION == 5
    # argument validation happens before any device work: a NULL matrix is rejected on a CPU-only box
    rc = lib.spr_reconstruct_f64(None, 10, 4, 4, 0, 10, 1, None, None, None, None, 1, None, 10, None)
    assert rc == -1
    assert b'NULL' in lib.spr_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, 'spr_reconstruct_f64')


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from openmeasure_amd.engine import HipEngine
    with pytest.raises(RuntimeError):
        HipEngine()
    # and nothing under the package imports the oracle or the test double
    pkg = os.path.join(ROOT, 'openmeasure_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt and 'numpy_engine' not in txt.replace(
                    'tests/numpy_engine.py', ''), f


def test_integration_doc_matches_prototypes():
    """The ctypes stub shown in INTEGRATION.md declares the same argument lists as the library binding."""
    import ctypes as C
    import os
    import re
    from openmeasure_amd import _lib
    names = {'_p': C.c_void_p, '_i32': C.c_int32, '_i64': C.c_int64, '_sz': C.c_size_t, '_dbl': C.c_double}
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'INTEGRATION.md')).read()
    found = re.findall(r'_lib\.(spr_\w+)\.argtypes = \[([^\]]*)\]', text)
    assert len(found) >= 5
    for fn, args in found:
        want = _lib.PROTOTYPES[fn][1]
        got = [names[a.strip()] for a in args.split(',') if a.strip()]
        assert got == list(want), fn


def test_loader_refuses_another_abi_version(monkeypatch):
    """A binding written for another argument layout must not call into the library (a signature that changes
    under an unchanged name would pass shifted arguments -- wild pointers on the GPU)."""
    _lib.load()
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'SPR_ABI_VERSION', _lib.SPR_ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match='ABI version'):
        _lib.load()
