#!/usr/bin/env python3
"""Is a kernel held back by power?  The same launch on all-zero data draws less (the matrix pipes toggle nothing) -- a kernel whose
time drops on zeros was clock-limited by power on real data, one whose time stays was not (GPU box; MI355X_MICROARCH.md
'DVFS give-back' item 1).  Gram pass and W-stationary projection of a 45M x 256 block."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.synth import make_R

eng = HipEngine()
cells, F, m, r = 5_000_000, 9, 256, 64
n = cells * F
X = eng.synth(n, m, 0, cells, eng.to_device(make_R(m, r)), 1e-3, 1)
Z = torch.zeros_like(X)
W = eng.to_device(np.random.default_rng(0).standard_normal((m, r)))
inv = eng.to_device(np.ones(F))
nrm = eng.empty((n,))
Ur = eng.empty((n, r))


def t(fn, reps=6):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for name, A in (('random data', X), ('all zeros  ', Z), ('random data', X), ('all zeros  ', Z)):
    rm, _, _ = eng.stats_gram(A, 0, cells, F)
    tg = t(lambda: eng.stats_gram(A, 0, cells, F))
    tp = t(lambda: eng.project(A, 0, cells, F, inv, W, out=Ur, rowmean=rm, norms=nrm))
    print(f'{name}: Gram {tg:7.3f} ms ({n * m * m / tg / 1e9:5.1f} TFLOP/s)   projection {tp:7.3f} ms ({2.0 * n * m * r / tp / 1e9:5.1f} TFLOP/s)', flush=True)
