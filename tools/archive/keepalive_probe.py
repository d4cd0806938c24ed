#!/usr/bin/env python3
"""Does a keep-alive kernel in the host gap of fit() pay?  (GPU box; round-4 record in profiles/r04_gap_filler_probe.txt.)
The spinner variants need the lab kernel tools/archive/lab/keepalive.hip built into the library (see its header); without it only the
idle / real-kernel-filler variants run.

One rank's block of BASELINE config 4 at N = 8 (11.25M rows x 256): the step's kernel sequence
Gram -> [3 ms host gap] -> projection -> reconstruct, repeated, each kernel timed with events, for the gap left idle and
for the gap filled by spr_keepalive_start in its modes (1: f64 MFMA on registers, 2: streaming reads, 3: both).
Also the sequence with NO gap at all (what the kernels cost back to back)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.synth import make_R

eng = HipEngine()
HAVE_SPINNER = hasattr(eng, 'keepalive_start')
cells, F, m, r = 1_250_000, 9, 256, 64
n = cells * F
gap_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
R = eng.to_device(make_R(m, r))
X = eng.synth(n, m, 0, cells, R, 1e-3, 1)
W = eng.to_device(np.random.default_rng(0).standard_normal((m, r)))
inv = eng.to_device(np.ones(F))
scale = eng.to_device(np.ones(F))
a = eng.to_device(np.random.default_rng(1).standard_normal((1, r)))
rowmean, _, _ = eng.stats_gram(X, 0, cells, F)
nrm = eng.empty((n,))
Ur = eng.project(X, 0, cells, F, inv, W, rowmean=rowmean, norms=nrm)
fld = eng.reconstruct(Ur, 0, cells, F, rowmean, scale, a)
torch.cuda.synchronize()


def ev():
    return torch.cuda.Event(enable_timing=True)


Xs = X[:n // 5]                                      # filler: the real Gram kernel on a fifth of the rows (~2.5 ms)
Xs2 = X[:int(n * 0.235)]                             # ~2.9 ms
Xp = X[:int(n * 0.4)]                                # projection of 40 % of the rows: ~2.6 ms
Up = eng.empty((Xp.shape[0], r))


def run(mode, gap, pre=False, filler=False):
    rows = []
    for it in range(12):
        e = [ev() for _ in range(6)]
        e[0].record(); rm, _, _ = eng.stats_gram(X, 0, cells, F); e[1].record()
        if gap:
            if pre and mode:                         # queued BEHIND the Gram pass: the queue never runs empty
                eng.keepalive_start(mode=mode, max_ms=8.0, stream_src=X)
            if filler == 1 or filler == 3:
                eng.stats_gram(Xs, 0, cells, F)
            elif filler == 2:
                eng.stats_gram(Xs2, 0, cells, F)
            elif filler == 4:
                eng.project(Xp, 0, cells, F, inv, W, out=Up, rowmean=rm)
            if filler == 3:
                eng.keepalive_start(mode=1, max_ms=8.0)
                mode = 1
            e[1].synchronize()                       # the host has the Gram matrix
            if mode and not pre:
                eng.keepalive_start(mode=mode, max_ms=8.0, stream_src=X)
            t_end = time.perf_counter() + gap * 1e-3
            while time.perf_counter() < t_end:       # the eigen-solve (busy host)
                pass
            if mode:
                eng.keepalive_stop()
        e[2].record(); eng.project(X, 0, cells, F, inv, W, out=Ur, rowmean=rm, norms=nrm); e[3].record()
        e[4].record(); eng.reconstruct(Ur, 0, cells, F, rm, scale, a, out=fld); e[5].record()
        torch.cuda.synchronize()
        rows.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3]), e[4].elapsed_time(e[5]),
                     e[0].elapsed_time(e[5])))
    return np.median(np.array(rows[3:]), axis=0)


for rep in range(2):
    for name, mode, gap, pre, filler in (('back to back', 0, 0.0, False, False), ('idle gap', 0, gap_ms, False, False),
                                         ('keep-alive mfma', 1, gap_ms, False, False),
                                         ('keep-alive both, pre-queued', 3, gap_ms, True, False),
                                         ('keep-alive gram-like (11), pre-queued', 11, gap_ms, True, False),
                                         ('real Gram slice 2.5 ms as filler', 0, gap_ms, False, 1),
                                         ('real Gram slice 2.9 ms as filler', 0, gap_ms, False, 2),
                                         ('Gram slice 2.5 ms + keep-alive mfma', 0, gap_ms, False, 3),
                                         ('real projection slice 2.6 ms as filler', 0, gap_ms, False, 4)):
        if (mode or filler == 3) and not HAVE_SPINNER:
            continue
        g, gp, p, rc, tot = run(mode, gap, pre, filler)
        print(f'{name:38s} gram {g:7.3f}  gap {gp:6.3f}  project {p:6.3f}  reconstruct {rc:6.3f}  total {tot:7.3f} ms', flush=True)
