#!/usr/bin/env python3
"""GPU box: where the host's time goes inside ONE fit() + reconstruct() step of a small workload (config 2: the 0.3 ms between
the kernels are a fifth of the step).  Every engine / LAPACK call of the step is wrapped with perf_counter stamps; the table gives,
averaged over the steps, each call's start offset from the beginning of fit() and its duration, next to the three kernels' GPU
times and the gaps between them from HIP events.     python tools/host_timeline.py [workload] [steps]"""
import os
import sys
import time
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import openmeasure_amd.sparse_sensing as ss  # noqa: E402
from openmeasure_amd.engine import HipEngine  # noqa: E402
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix  # noqa: E402
from openmeasure_amd.synth import make_R  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'c2']
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
eng = HipEngine('cuda:0')
R = eng.to_device(make_R(wl['m'], wl['s'], seed=1234))
Xd = eng.synth(wl['cells'] * wl['features'], wl['m'], 0, wl['cells'], R, 1e-3, 1234)
spr = SPR(DeviceMatrix(Xd), wl['features'], None, engine=eng)
log = []
t_fit = [0.0]


def wrap(obj, name, label=None):
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            log.append((label or name, t0 - t_fit[0], time.perf_counter() - t0))
    setattr(obj, name, w)


for n in ('stats_gram', 'gram_combine', 'to_host', 'to_device', 'project', 'reconstruct', 'empty', 'zeros', 'timing_event'):
    if hasattr(eng, n):
        wrap(eng, n, 'eng.' + n)
wrap(ss, '_eigh_small')
wrap(ss, '_eigh_tridiagonal')
wrap(ss, '_eigvecs_top')
wrap(ss, '_sign_fix')
for n in ('_merge_stats', '_basis_from_gram', '_spectrum', '_gram_collective', '_select_rank', '_norms_buffer', '_close_gap',
          '_set_precenter_ratio', '_needs_precenter', '_device_fit', '_stats_pass'):
    wrap(SPR, n, 'rom.' + n)

spr.fit(select_modes='number', n_modes=wl['s'])
a = eng.to_device(spr.Ar[:1].copy())
for _ in range(10):
    spr.fit(select_modes='number', n_modes=wl['s']); spr.reconstruct(a, to_host=False)
torch.cuda.synchronize()
del log[:]
fits, recs, timers = [], [], []
t_all = time.perf_counter()
for _ in range(steps):
    timers.append((eng.time_next('stats_gram'), eng.time_next('project'), eng.time_next('reconstruct')))
    t_fit[0] = time.perf_counter()
    spr.fit(select_modes='number', n_modes=wl['s'])
    t1 = time.perf_counter()
    spr.reconstruct(a, to_host=False)
    t2 = time.perf_counter()
    fits.append(t1 - t_fit[0]); recs.append(t2 - t1)
torch.cuda.synchronize()
ms = (time.perf_counter() - t_all) / steps * 1e3
agg = defaultdict(list)
for name, off, dur in log:
    agg[name].append((off, dur))
print(f'{sys.argv[1] if len(sys.argv) > 1 else "c2"}: {ms:.4f} ms per step over {steps} steps; host: fit() {1e3 * np.mean(fits):.4f} ms, reconstruct() {1e3 * np.mean(recs):.4f} ms')
k = [float(np.mean([tm[i][0].elapsed_time(tm[i][1]) for tm in timers])) for i in range(3)]
g1 = float(np.mean([tm[0][1].elapsed_time(tm[1][0]) for tm in timers]))
g2 = float(np.mean([tm[1][1].elapsed_time(tm[2][0]) for tm in timers]))
g3 = float(np.mean([x[2][1].elapsed_time(y[0][0]) for x, y in zip(timers, timers[1:])]))
print(f'GPU (events): gram {k[0]:.4f} | gap {g1:.4f} | project {k[1]:.4f} | gap {g2:.4f} | reconstruct {k[2]:.4f} | gap to the next gram {g3:.4f}  '
      f'(sum {sum(k) + g1 + g2 + g3:.4f})')
print(f'{"call":28s} {"calls/step":>10s} {"first start us":>15s} {"total us/step":>14s}')
rows = []
for name, v in agg.items():
    per = len(v) / steps
    first = np.mean(sorted(o for o, _ in v)[:max(1, len(v) // max(1, int(round(per))))]) if per >= 1 else np.mean([o for o, _ in v])
    rows.append((np.mean([o for o, _ in v[::max(1, int(round(per)))]]), name, per, sum(d for _, d in v) / steps))
for first, name, per, tot in sorted(rows):
    print(f'{name:28s} {per:10.2f} {1e6 * first:15.1f} {1e6 * tot:14.1f}')
