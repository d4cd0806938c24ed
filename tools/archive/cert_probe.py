"""Where do the batches of optimal_placement end?  Per refresh: tau (the largest residual a NON-candidate can have), the
winner's residual at the first and the last certified step, the number of certified steps -- and the same run with the tau
a global top-K candidate set (K = 16 x blocks) would have had, computed from the norm vector on the host side of the GPU
(torch.topk) at every refresh.

usage: python tools/cert_probe.py [c3|c3s|c2|c5s]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

import bench  # noqa: E402
from openmeasure_amd.engine import HipEngine  # noqa: E402
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix  # noqa: E402
from openmeasure_amd.synth import make_R  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
wl = bench.WORKLOADS[name]
eng = HipEngine('cuda:0')
F, m, s = wl['features'], wl['m'], wl['s']
n_points = wl['cells']
n = n_points * F
R = eng.to_device(make_R(m, s, seed=1234))
Xd = eng.synth(n, m, 0, n_points, R, 1e-3, 1234)
spr = SPR(DeviceMatrix(Xd), F, None, engine=eng)
spr.fit(select_modes='number', n_modes=s)
Ur = spr._d['Ur']
st = eng.qr_begin(Ur, 0, s, norms=spr._d.get('nrm0'))
K = 16 * 1024
j = 0
print(f'{name}: n={n} r={s}; per batch: tau | global top-{K} tau | first winner | last certified winner | certified')
while j < s:
    nb = min(eng.qr_batch, s - j)
    tau = float(st['tau'].item())
    nrm = st['nrm']
    gtau = float(torch.topk(nrm, K + 1).values[-1].item())
    first = float(st['rec'][0].item())
    eng.qr_steps(st, j, nb)
    ok = eng.to_host(st['ok'][j:j + nb])
    k = nb if ok.all() else int(np.argmin(ok))
    gaps = eng.to_host(st['gap'][j:j + nb])
    # residual of step t's winner: re-derive from Q? keep it simple: the candidate record after the steps holds the NEXT best
    nxt = float(st['rec'][0].item())
    print(f'  steps {j:3d}..{j + k - 1:3d}: tau {tau:.4e} | {gtau:.4e} | first {first:.4e} | next-best after batch {nxt:.4e} | {k} of {nb}')
    j += k
    if j < s:
        eng.qr_refresh(st, j - k, k)
print('pivots equal to a plain placement:', bool(np.array_equal(eng.to_host(st['piv']), spr.optimal_placement() is not None and spr.sensors_)))
