"""Cross block of the wide Gram path (csrc/gram_wide.hip) in centre mode 1 (it forms the full-row means itself) against mode 2
(external per-row constants, no row sums): what shifting all three launches of a 256 < m <= 512 matrix by the SAME constant
and applying P G P afterwards could save.   usage: python tools/cross_mode_probe.py [cells] [f32]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from openmeasure_amd import _lib  # noqa: E402
from openmeasure_amd.engine import HipEngine, _ptr  # noqa: E402
from openmeasure_amd.synth import make_R  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
f32 = len(sys.argv) > 2
eng = HipEngine('cuda:0')
F, m = 16, 512
n = cells * F
R = eng.to_device(make_R(m, 128, seed=1234))
X = eng.synth(n, m, 0, cells, R, 1e-3, 1234, dtype=torch.float32 if f32 else None)
gram = eng.empty((F, m, m))
rowmean = eng.empty((n,))
ws = eng._workspace('cross', eng.lib.spr_gram_cross_workspace(m, F))
fn = eng._x('spr_gram_cross', X)
st = eng._stream()
for mode in (1, 2, 1, 2):
    for _ in range(2):
        _lib.check(fn(_ptr(X), n, m, X.stride(0), 0, cells, F, mode, _ptr(rowmean), _ptr(gram), _ptr(ws), ws.numel(), st), 'cross')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        _lib.check(fn(_ptr(X), n, m, X.stride(0), 0, cells, F, mode, _ptr(rowmean), _ptr(gram), _ptr(ws), ws.numel(), st), 'cross')
    torch.cuda.synchronize()
    print(f'cross block, centre mode {mode}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms', flush=True)
