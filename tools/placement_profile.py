"""Host-side profile of optimal_placement() at a small workload, where the host's share is visible (GPU box):
python tools/placement_profile.py [workload]   -- wall time per call, then cProfile (tottime) of 50 calls."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
from openmeasure_amd.synth import make_R
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'c1']
eng = HipEngine('cuda:0')
R = eng.to_device(make_R(wl['m'], wl['s'], seed=1234))
Xd = eng.synth(wl['cells'] * wl['features'], wl['m'], 0, wl['cells'], R, 1e-3, 1234)
spr = SPR(DeviceMatrix(Xd), wl['features'], None, engine=eng)
spr.fit(select_modes='number', n_modes=wl['s'])
for _ in range(5):
    spr.optimal_placement()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    spr.optimal_placement()
torch.cuda.synchronize()
print('ms/placement', (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    spr.optimal_placement()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30); print(s.getvalue()[:7000])
