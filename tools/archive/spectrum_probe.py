import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openmeasure_amd.engine import HipEngine
eng = HipEngine()
rng = np.random.default_rng(0)
for m, F in [(64, 4), (41, 9), (12, 3)]:
    X = rng.standard_normal((4000 * F, m)) @ np.diag(0.8 ** np.arange(m))
    Xd = eng.to_device(X)
    rm, fs, g = eng.stats_gram(Xd, 0, 4000, F)
    sp = eng.spectrum(g, fs[None], 'std', min(m, 32)); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sp = eng.spectrum(g, fs[None], 'std', min(m, 32)); e1.record(); torch.cuda.synchronize()
    print(m, F, 'ms', e0.elapsed_time(e1), 'info', sp['info'].cpu().numpy())
