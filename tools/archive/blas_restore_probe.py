"""Does restoring the BLAS pool size after the small eigen-solve stall the next H2D copy? (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from threadpoolctl import ThreadpoolController
from openmeasure_amd.engine import HipEngine
eng = HipEngine()
ctl = ThreadpoolController()
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), [ (l.user_api, l.num_threads) for l in ctl.lib_controllers])
rng = np.random.default_rng(0)
m = 256
A = rng.standard_normal((4 * m, m)) * np.logspace(0, -4, m); G = A.T @ A
W = rng.standard_normal((m, 64))
def run(label, fn, reps=5):
    res = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
        eng.to_device(W / 1.0); t2 = time.perf_counter()
        res.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    print(f'{label:34s} eigh ' + ' '.join(f'{a:6.2f}' for a, _ in res) + '   then H2D ' + ' '.join(f'{b:6.2f}' for _, b in res))
def scoped():
    with ctl.limit(limits=1, user_api='blas'):
        np.linalg.eigh(G)
run('scoped limit(1) + restore', scoped)
run('no limit', lambda: np.linalg.eigh(G))
lim = ctl.limit(limits=1, user_api='blas')
run('permanent limit(1)', lambda: np.linalg.eigh(G))
lim.restore_original_limits()
import scipy.linalg as sl
run('scipy evd, no limit', lambda: sl.eigh(G, driver='evd'))
def scoped_sl():
    with ctl.limit(limits=1, user_api='blas'):
        sl.eigh(G, driver='evd')
run('scipy evd, scoped', scoped_sl)
