"""Where does optimal_placement spend its wall time? (GPU box)
usage: placement_probe.py [rows of a random basis | c3 | c3s]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.sparse_sensing import SPR, DeviceMatrix
from openmeasure_amd.synth import make_R
eng = HipEngine()
def sync(): torch.cuda.synchronize()
arg = sys.argv[1] if len(sys.argv) > 1 else '9000000'
r = 64
if arg in ('c3', 'c3s'):
    cells, F, m = (10_000_000 if arg == 'c3' else 1_000_000), 9, 256
    R = eng.to_device(make_R(m, r))
    Xd = eng.synth(cells * F, m, 0, cells, R, 1e-3, 1234)
    spr = SPR(DeviceMatrix(Xd), F, None, engine=eng)
    spr.fit(select_modes='number', n_modes=r)
    U = spr._d['Ur']
    n = U.shape[0]
else:
    n = int(arg)
    U = torch.empty(n, r, dtype=torch.float64, device='cuda')
    for i0 in range(0, n, 10_000_000):
        U[i0:i0 + 10_000_000].normal_()
    U.mul_(n ** -0.5)
    spr = None
for rep in range(3):
    sync(); t0 = time.perf_counter()
    st = eng.qr_begin(U, 0, r)
    sync(); t1 = time.perf_counter()
    j, sweeps, tstep, tsync, tref, ks, refs = 0, 1, 0.0, 0.0, 0.0, [], []
    while j < r:
        nb = min(eng.qr_batch, r - j)
        ta = time.perf_counter()
        for t in range(nb):
            eng.qr_step(st, j + t, st['rec'][None], st['tau'][None], first=(t == 0))
        tb = time.perf_counter()
        ok = eng.to_host(st['ok'][j:j + nb])
        tc = time.perf_counter()
        k = nb if ok.all() else int(np.argmin(ok))
        j += k; ks.append(k)
        if j < r:
            eng.qr_refresh(st, j - k, k); sync()
        td = time.perf_counter()
        tstep += tb - ta; tsync += tc - tb; tref += td - tc; refs.append(1e3 * (td - tc))
        sweeps += 1
    print(f'rep {rep}: begin {1e3*(t1-t0):.2f} ms, issue steps {1e3*tstep:.2f}, wait flags {1e3*tsync:.2f}, '
          f'refresh {1e3*tref:.2f}, sweeps {sweeps}, certified per batch {ks}, refresh ms {[round(x, 1) for x in refs]}')
if spr is not None:
    import gc
    gc_log = []
    def gc_cb(phase, info):
        if phase == 'start':
            gc_log.append([info['generation'], time.perf_counter(), None])
        else:
            gc_log[-1][2] = 1e3 * (time.perf_counter() - gc_log[-1][1])
    gc.callbacks.append(gc_cb)
    slow = []
    for name in ('qr_begin', 'qr_step', 'qr_refresh', 'to_host'):
        f = getattr(eng, name)
        def g(*a, _f=f, _n=name, **k):
            t0 = time.perf_counter(); r_ = _f(*a, **k); dt = 1e3 * (time.perf_counter() - t0)
            if dt > 3: slow.append((_n, round(dt, 1)))
            return r_
        setattr(eng, name, g)
    for call in range(8):
        if call == 5:
            gc.disable(); print('gc disabled')
        gc_log.clear(); slow.clear()
        st0 = torch.cuda.memory_stats()['num_device_alloc']
        t0 = time.perf_counter()
        spr.optimal_placement(); sync()
        print(f'class call {call}: {1e3*(time.perf_counter()-t0):.2f} ms, sweeps {spr.pivot_sweeps_}, slow host calls {slow}, '
              f'gc {[(g_, round(d, 1)) for g_, _, d in gc_log]}, new device allocs {torch.cuda.memory_stats()["num_device_alloc"] - st0}')
