#!/usr/bin/env python3
"""Repeat the whole step (fit with the gap filler forced on, placement, train, predict, reconstruct) a few hundred times on
the same resident matrix and count results that differ IN ANY BIT from the first round: spectrum, sensors, coefficients, field.
A race -- the filler's re-queued Gram launch against the real one, the side-stream copies, the fused step kernel's tickets --
would show up as a sporadic difference.  GPU box.   usage: python tools/step_stress.py [rounds] [workload:cells ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
from openmeasure_amd.synth import make_R

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng = HipEngine('cuda:0')
shapes = [(a.split(':')[0], int(a.split(':')[1])) for a in sys.argv[2:]] or [('c3', 250_000), ('c2', 300_000), ('c5s', 60_000),
                                                                             ('c3', 1_250_000)]
for name, cells in shapes:
    wl = bench.WORKLOADS[name]
    F, m, s = wl['features'], wl['m'], wl['s']
    Xd = eng.synth(cells * F, m, 0, cells, eng.to_device(make_R(m, s, seed=1234)), 1e-3, 1234)
    spr = SPR(DeviceMatrix(Xd), F, None, engine=eng)
    spr.gap_filler = True
    spr._GAP_FILL_MIN_MS = 0.0                      # fill whatever gap this host leaves
    ref, bad, fills, pend, deferred = None, 0, 0, None, 0
    reps = rounds if (cells < 1_000_000 or len(sys.argv) > 2) else max(rounds // 4, 10)
    t0 = time.time()
    for it in range(reps):
        spr.fit(select_modes='number', n_modes=s)
        fills += int(getattr(spr, '_gap_fill_rows', 0) > 0)
        if pend is not None:
            # round 6: the previous round's field once more in the ASYNCHRONOUS form -- deferred by default (ROM.defer_reconstruct):
            # its kernel ran in the host gap of the fit() above, on the basis of the round it was called in
            deferred += int(pend.launched)
            if not np.array_equal(eng.to_host(pend.wait())[0], ref[3]):
                bad += 1
        C = spr.optimal_placement()
        spr.train(C)
        piv = spr.sensors_
        y = np.zeros((s, 3))
        y[:, 0] = eng.to_host(Xd[torch.as_tensor(piv, device=Xd.device), 0])
        y[:, 2] = piv // cells
        a, _ = spr.predict(y)
        x = spr.reconstruct(a)
        pend = spr.reconstruct(a, to_host=False, wait=False) if it % 2 else None     # (a gap holds the deferred launch OR the filler)
        got = (spr.Sigma_r.copy(), piv.copy(), a.copy(), x[:, 0].copy())
        if ref is None:
            ref = got
        elif not all(np.array_equal(u, v) for u, v in zip(got, ref)):
            bad += 1
    print(f'{name} shape, {cells} cells (m = {m}, s = {s}): {reps} rounds, {bad} differing from the first in any bit, '
          f'filler ran in {fills}, deferred reconstructs launched inside the next fit {deferred}, {time.time() - t0:.1f} s', flush=True)
    del spr, Xd
    torch.cuda.empty_cache()
