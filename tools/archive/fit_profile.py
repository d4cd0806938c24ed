"""Host-side profile of fit()+reconstruct() at a small workload (where the Python/ctypes overhead between the kernels is
a visible share of the step): python tools/fit_profile.py [workload]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
from openmeasure_amd.synth import make_R
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'c2']
eng = HipEngine('cuda:0')
R = eng.to_device(make_R(wl['m'], wl['s'], seed=1234))
Xd = eng.synth(wl['cells'] * wl['features'], wl['m'], 0, wl['cells'], R, 1e-3, 1234)
spr = SPR(DeviceMatrix(Xd), wl['features'], None, engine=eng)
spr.fit(select_modes='number', n_modes=wl['s'])
a = eng.to_device(spr.Ar[:1].copy())
for _ in range(5):
    spr.fit(select_modes='number', n_modes=wl['s']); spr.reconstruct(a, to_host=False)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50):
    spr.fit(select_modes='number', n_modes=wl['s']); spr.reconstruct(a, to_host=False)
torch.cuda.synchronize()
print('ms/step', (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    spr.fit(select_modes='number', n_modes=wl['s']); spr.reconstruct(a, to_host=False)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
