"""predict() on a wide basis (r > 128): normal-equations workspace kernel vs the SVD (pinv) workspace kernel, per call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from openmeasure_amd.engine import HipEngine
eng = HipEngine('cuda:0')
rng = np.random.default_rng(0)
for r in (200, 300, 600, 1024):
    s = r
    Theta = np.linalg.qr(rng.standard_normal((s, r)))[0] * np.logspace(0, -1, r)
    y = np.zeros((1, s, 3)); y[0, :, 0] = rng.standard_normal(s)
    args = (eng.to_device(Theta), eng.to_device(np.zeros(s)), eng.to_device(np.ones(1)), eng.to_device(y))
    for name, fn in (('ols_wide', eng.solve_ols), ('pinv_wide', eng.solve_pinv)):
        fn(*args); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(*args); torch.cuda.synchronize()
        print(f'r={r} {name}: {1e3 * (time.perf_counter() - t0):.1f} ms')
