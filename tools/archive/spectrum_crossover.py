"""fit() wall time with the device eigen-solver (spectrum.hip, no host sync) vs the host dsyevd path, by m. (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openmeasure_amd.engine import HipEngine
import openmeasure_amd.sparse_sensing as ss
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
from openmeasure_amd.synth import make_R
eng = HipEngine()
cells, F = 1_000_000, 4
for m, r in ((12, 6), (16, 8), (24, 8), (32, 16), (41, 14), (48, 16), (64, 32)):
    R = eng.to_device(make_R(m, r))
    Xd = eng.synth(cells * F, m, 0, cells, R, 1e-3, 1234)
    out = []
    for cap in (0, 64):
        ss._DEVICE_SPECTRUM_MAX_M = cap
        spr = SPR(DeviceMatrix(Xd), F, None, engine=eng)
        for _ in range(3):
            spr.fit(select_modes='number', n_modes=r)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            spr.fit(select_modes='number', n_modes=r)
        torch.cuda.synchronize(); out.append(1e3 * (time.perf_counter() - t0) / 20)
    print(f'm={m:3d} r={r:3d}: host eigh path {out[0]:.3f} ms/fit, device spectrum path {out[1]:.3f} ms/fit')
