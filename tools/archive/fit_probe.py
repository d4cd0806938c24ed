"""Steady-state SPR.fit at a bench workload (GPU box): per-call wall time and the stream-sync time of the
small W upload that follows the host eigen-solve.  usage: fit_probe.py [base|busy] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
import openmeasure_amd.sparse_sensing as ss
from openmeasure_amd.synth import make_R
mode = sys.argv[1] if len(sys.argv) > 1 else 'base'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cells, F, m, s = 10_000_000, 9, 256, 64
eng = HipEngine()
def sync(): torch.cuda.synchronize()
R = eng.to_device(make_R(m, s))
Xd = eng.synth(cells * F, m, 0, cells, R, 1e-3, 1234)
spr = SPR(DeviceMatrix(Xd), F, None, engine=eng)
waits = []
orig = eng.to_device
def timed(a, dtype=None):
    t0 = time.perf_counter(); t = orig(a, dtype); waits.append((np.asarray(a).size, 1e3 * (time.perf_counter() - t0)))
    return t
eng.to_device = timed
if mode == 'pin':
    pinned = torch.empty(8 << 20, dtype=torch.float64, pin_memory=True)
    ev_pin = torch.cuda.Event()
    def to_host_pinned(t):
        t = t.contiguous()
        buf = pinned[:t.numel()].view(t.shape)
        buf.copy_(t, non_blocking=True); ev_pin.record(torch.cuda.current_stream()); ev_pin.synchronize()
        return buf.numpy().copy()
    eng.to_host = to_host_pinned
if mode == 'busy':
    orig_eigh = ss._eigh_small
    busy = torch.empty(6_000_000_000 // 8, dtype=torch.float64, device='cuda')
    def eigh_busy(G):
        busy.mul_(1.0); busy.mul_(1.0)
        return orig_eigh(G)
    ss._eigh_small = eigh_busy
marks = []
def gpu_mark(label):
    e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream()); marks.append((label, e))
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        gpu_mark(label + '<')
        t0 = time.perf_counter(); r_ = f(*a, **k); cur[label] = cur.get(label, 0.0) + 1e3 * (time.perf_counter() - t0)
        gpu_mark(label + '>')
        return r_
    setattr(obj, name, g)
cur = {}
wrap(eng, 'to_host', 'to_host'); wrap(ss, '_eigh_small', 'eigh'); wrap(eng, 'stats_gram', 'launch_gram')
wrap(eng, 'project', 'launch_proj'); wrap(eng, 'reconstruct', 'launch_rec')
for _ in range(3):
    spr.fit(select_modes='number', n_modes=s); x = spr.reconstruct(np.ones(s), to_host=False)
sync(); waits.clear()
rows = []
for _ in range(reps):                                   # like bench.py: one sync per step
    cur = {}; marks.clear()
    evs = {k: eng.time_next(k) for k in ('stats_gram', 'project', 'reconstruct')}
    t0 = time.perf_counter()
    spr.fit(select_modes='number', n_modes=s); x = spr.reconstruct(np.ones(s), to_host=False)
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    k = {n_: e0.elapsed_time(e1) for n_, (e0, e1) in evs.items()}
    gaps = (evs['stats_gram'][1].elapsed_time(evs['project'][0]), evs['project'][1].elapsed_time(evs['reconstruct'][0]))
    tl = ' '.join(f'{marks[i + 1][0]}+{marks[i][1].elapsed_time(marks[i + 1][1]):.1f}' for i in range(len(marks) - 1))
    rows.append((1e3 * (t2 - t0), 1e3 * (t2 - t1), dict(cur), k, gaps, tl))
for tot, fin, c, k, gaps, tl in rows:
    print(f'{mode} step {tot:6.1f} final-sync {fin:5.1f} | host: ' + ' '.join(f'{a}={b:.1f}' for a, b in c.items())
          + ' | gpu: ' + ' '.join(f'{a}={b:.1f}' for a, b in k.items()) + f' gap gram->proj {gaps[0]:.1f} proj->rec {gaps[1]:.1f}')
    print(mode, '   gpu timeline:', tl)
