"""Where does optimal_placement spend its wall time? (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.sparse_sensing import pivot_loop
eng = HipEngine()
n, r = 9_000_000, 64
U = torch.randn(n, r, dtype=torch.float64, device='cuda')
U, _ = torch.linalg.qr(U[:200000]) if False else (U / (n ** 0.5), None)
def sync(): torch.cuda.synchronize()
for rep in range(3):
    sync(); t0 = time.perf_counter()
    st = eng.qr_begin(U, 0, r)
    sync(); t1 = time.perf_counter()
    j, sweeps, tstep, tsync, tref = 0, 1, 0.0, 0.0, 0.0
    while j < r:
        nb = min(eng.qr_batch, r - j)
        ta = time.perf_counter()
        for t in range(nb):
            eng.qr_step(st, j + t, st['rec'][None], st['tau'][None], first=(t == 0))
        tb = time.perf_counter()
        ok = eng.to_host(st['ok'][j:j + nb])
        tc = time.perf_counter()
        k = nb if ok.all() else int(np.argmin(ok))
        j += k
        if j < r:
            eng.qr_refresh(st, j - k, k); sync()
        td = time.perf_counter()
        tstep += tb - ta; tsync += tc - tb; tref += td - tc
        sweeps += 1
    print(f'rep {rep}: begin {1e3*(t1-t0):.2f} ms, issue steps {1e3*tstep:.2f}, wait flags {1e3*tsync:.2f}, refresh {1e3*tref:.2f}, sweeps {sweeps}')

# the same through the SPR class on the synthetic c3s matrix
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
from openmeasure_amd.synth import make_R
import cProfile, pstats
del U
cells, F, m, s = 1_000_000, 9, 256, 64
R = eng.to_device(make_R(m, s))
Xd = eng.synth(cells * F, m, 0, cells, R, 1e-3, 1234)
spr = SPR(DeviceMatrix(Xd), F, None, engine=eng)
spr.fit(select_modes='number', n_modes=s)
spr.optimal_placement(); sync()
t0 = time.perf_counter(); spr.optimal_placement(); sync(); print(f'rep class: {1e3*(time.perf_counter()-t0):.2f} ms, sweeps {spr.pivot_sweeps_}')
pr = cProfile.Profile(); pr.enable(); spr.optimal_placement(); sync(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
