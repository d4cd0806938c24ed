"""optimal_placement with epoch sweeps (SPR.placement_pools): the sequence of batches, pool sweeps and full sweeps at a bench
workload, and the wall time of each phase (device-synchronised around every call: slower than the real run, which only
synchronises once per batch).   usage: python tools/pool_probe.py [c3|c3s|c5|c5s|c2]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

import bench  # noqa: E402
from openmeasure_amd.engine import HipEngine  # noqa: E402
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix, pivot_loop  # noqa: E402
from openmeasure_amd.synth import make_R  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
wl = bench.WORKLOADS[name]
eng = HipEngine('cuda:0')
F, m, s = wl['features'], wl['m'], wl['s']
f32 = wl.get('storage') == 'f32'
n_points = wl['cells']
n = n_points * F
R = eng.to_device(make_R(m, s, seed=1234))
Xd = eng.synth(n, m, 0, n_points, R, 1e-3, 1234, dtype=torch.float32 if f32 else None)
spr = SPR(DeviceMatrix(Xd, basis='f32' if f32 else None), F, None, engine=eng)
spr.fit(select_modes='number', n_modes=s)

# time every engine call of the placement
calls = []
for fn in ('qr_begin', 'qr_steps', 'qr_epoch_sweep', 'qr_pool_build', 'qr_epoch_begin', 'qr_refresh', 'to_host'):
    def wrap(f, nm):
        def g(*a, **k):
            torch.cuda.synchronize(); t = time.perf_counter()
            out = f(*a, **k)
            torch.cuda.synchronize(); calls.append((nm + (' pool' if k.get('pool') else ''), 1e3 * (time.perf_counter() - t)))
            return out
        return g
    setattr(eng, fn, wrap(getattr(eng, fn), fn))
for rep in range(3):
    calls.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    spr.optimal_placement()
    torch.cuda.synchronize(); tot = 1e3 * (time.perf_counter() - t0)
    agg = {}
    for nm, ms in calls:
        a = agg.setdefault(nm, [0, 0.0]); a[0] += 1; a[1] += ms
    print(f'rep {rep}: {tot:.2f} ms, full sweeps {spr.pivot_sweeps_}, pool sweeps {spr.pivot_pool_sweeps_}: ' +
          ', '.join(f'{k} x{v[0]} {v[1]:.2f}' for k, v in agg.items()))
st = eng.qr_begin(spr._d['Ur'], 0, s, norms=spr._d.get('nrm0'))
stats = {}
pivot_loop(eng, st, s, pools=True, stats=stats)
for e in stats['log']:
    print('  ', e)
