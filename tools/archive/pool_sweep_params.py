#!/usr/bin/env python3
"""optimal_placement('qr') at a bench workload under different pool parameters (share of the rows a pool holds, how close
the winners may come to the pool's threshold before a full sweep): time, full and pool sweeps, sensors (must not change).
usage: python tools/pool_sweep_params.py [c3|c3s|c5s|c5|c2] [cells]
Result of round 4 (profiles/r04_placement_pool_probe.txt): no share beats 1/16 across shapes; the defaults stayed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import openmeasure_amd.sparse_sensing as ss
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
from openmeasure_amd.synth import make_R

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
wl = bench.WORKLOADS[name]
eng = HipEngine('cuda:0')
F, m, s = wl['features'], wl['m'], wl['s']
n_points = int(sys.argv[2]) if len(sys.argv) > 2 else wl['cells']
f32 = wl.get('storage') == 'f32'
Xd = eng.synth(n_points * F, m, 0, n_points, eng.to_device(make_R(m, s, seed=1234)), 1e-3, 1234, dtype=torch.float32 if f32 else None)
spr = SPR(DeviceMatrix(Xd, basis='f32' if f32 else None), F, None, engine=eng)
spr.fit(select_modes='number', n_modes=s)
del Xd
ref = None
ROWS = int(os.environ.get('POOL_SWEEP_ROWS', '99'))
for frac, margin in ((1 / 16, 1.15), (1 / 14, 1.15), (1 / 12, 1.15), (1 / 11, 1.15), (1 / 10, 1.15), (1 / 9, 1.15), (1 / 8, 1.15),
                     (1 / 16, 1.25), (1 / 10, 1.25)):
    ss._POOL_FRACTION, ss._POOL_MARGIN = frac, margin
    ROWS -= 1
    if ROWS < 0:
        break
    spr.optimal_placement(); spr.optimal_placement()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        spr.optimal_placement()
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    if ref is None:
        ref = spr.sensors_.copy()
        for e in spr.pivot_log_:      # ('batch', first step, certified, tried, level, tau, theta, pool rows) / ('pool sweep' | 'full sweep', epoch start, step)
            print('   ', e, flush=True)
    print(f'pool fraction 1/{round(1 / frac)}, margin {margin}: {sorted(ts)[1]:7.2f} ms, full sweeps {spr.pivot_sweeps_}, '
          f'pool sweeps {spr.pivot_pool_sweeps_}, same sensors {np.array_equal(ref, spr.sensors_)}', flush=True)
