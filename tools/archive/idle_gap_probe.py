#!/usr/bin/env python3
"""Does a few-ms idle gap in front of a kernel cost time?  Gram pass / projection of a 9M x 256 shard launched back to back
vs after a host sleep of 1 / 3 / 10 / 30 ms.  Measured (MI355X): Gram 9.55 ms back to back, 10.07 / 10.73 / 11.27 / 11.43 ms
after the gaps; projection 5.05 -> 5.26 / 5.69 / 6.14 / 6.31 ms.  A keep-alive kernel in the gap (1 to 1024 single-wave
workgroups polling a pinned flag, sleeping or spinning on f32 FMAs) changed nothing, so it is not kept in the library:
the chip ramps its clock with the LOAD of the previous milliseconds, not with mere occupancy."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.synth import make_R

eng = HipEngine()
cells, F, m, r = 1_000_000, 9, 256, 64
n = cells * F
R = eng.to_device(make_R(m, r))
X = eng.synth(n, m, 0, cells, R, 1e-3, 1)
W = eng.to_device(np.random.default_rng(0).standard_normal((m, r)))
inv = eng.to_device(np.ones(F))
rowmean, _, _ = eng.stats_gram(X, 0, cells, F)
Ur = eng.project(X, 0, cells, F, inv, W, rowmean=rowmean)
torch.cuda.synchronize()


def run(gap_ms, what):
    ts = []
    for _ in range(8):
        torch.cuda.synchronize()
        if gap_ms:
            time.sleep(gap_ms * 1e-3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if what == 'gram':
            eng.stats_gram(X, 0, cells, F)
        else:
            eng.project(X, 0, cells, F, inv, W, out=Ur, rowmean=rowmean)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[2:]))


for what in ('gram', 'project'):
    print(what, ' '.join(f'gap {g} ms: {run(g, what):.3f} ms' for g in (0, 1, 3, 10, 30)))
