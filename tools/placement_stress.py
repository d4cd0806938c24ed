"""Repeat a placement a few hundred times (the three drivers in rotation) and count results that differ from the first: a race in the
last-workgroup logic of the fused step kernel or in the pool build would show up as sporadic pivots.  GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openmeasure_amd.engine import HipEngine
from openmeasure_amd.sparse_sensing import pivot_loop
eng = HipEngine('cuda:0')
rng = np.random.default_rng(5)
for (n, r, f32) in ((2_000_000, 64, False), (500_000, 128, True), (300_000, 32, False), (50_000, 16, False)):
    U = torch.randn(n, r, dtype=torch.float64, device='cuda') * torch.exp(0.7 * torch.randn(n, 1, dtype=torch.float64, device='cuda'))
    U = (U / n ** 0.5)
    if f32:
        U = U.float()
    ref = None
    bad = 0
    t0 = time.time()
    reps = 300 if n <= 500_000 else 150
    for it in range(reps):
        st = eng.qr_begin(U, 0, r)
        # three drivers in rotation: pooled, a refresh per batch, and the per-step path of a sharded placement (one library call per
        # step between the ranks' record exchanges: qr_orth_kernel + the fused down-date / search kernel) with a one-rank gather
        pivot_loop(eng, st, r, pools=(it % 3 == 0), all_gather=(lambda t: t[None]) if it % 3 == 2 else None)
        piv = eng.to_host(st['piv']).copy()
        if ref is None:
            ref = piv
        elif not np.array_equal(piv, ref):
            bad += 1
    print(f'n={n} r={r} f32={f32}: {reps} placements, {bad} differing, {time.time() - t0:.1f}s', flush=True)
